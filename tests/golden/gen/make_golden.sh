#!/bin/bash
# Build-container-only recipe that (re)creates tests/golden/vectors.json.
# 1. builds the UNMODIFIED reference out of tree with its own CMake,
# 2. links the ref_dump driver against the resulting object files,
# 3. runs make_golden.py.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
B=/tmp/refbuild
if [ ! -x $B/Sibelia ]; then
  mkdir -p $B && cd $B
  cmake /root/reference/src -DONLY_SIBELIA=1 -DCMAKE_POLICY_VERSION_MINIMUM=3.5 > cmake.log
  make -j8 > make.log 2>&1
fi
cd $B
OBJS=""
for f in indexedsequence blockfinder bifurcationstorage bulgeremoval dnasequence edge fasta serialization synteny platform stranditerator vertexenumeration blockinstance util; do
  OBJS="$OBJS CMakeFiles/Sibelia.dir/$f.cpp.o"
done
g++ -std=gnu++14 -O2 -DNDEBUG -w -I/root/reference/src -I/root/reference/src/include \
    -I$B/libdivsufsort-2.0.1/include "$HERE/ref_dump.cpp" $OBJS libdivsufsort-2.0.1/lib/libdivsufsort.a -o $B/ref_dump
python3 "$HERE/make_golden.py" "$@"
