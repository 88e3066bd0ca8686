#!/bin/bash
# TEST INFRASTRUCTURE -- the reference's own PROGRAM (src/sibelia.cpp: command line, FASTA reader, post-processor, writers) built twice
# from the sources where they lie under /root/reference, CMake-free like build_ref.sh, outputs into oracle/_ref/ only:
#   oracle/_ref/sibelia_ref      the unmodified reference (all of src/CMakeLists.txt:11's units): writes the fixtures of
#                                tests/golden/dropin_cases.json (tests/golden/gen/make_dropin_golden.py), CPU only
#   oracle/_ref/sibelia_dropin   the same program with the five translation units that define BlockFinder's members
#                                (blockfinder, bulgeremoval, edge, serialization, synteny) replaced by integration/blockfinder_amd.cpp
#                                over libsibelia_amd.so, and Postprocessor::GlueStripes by integration/postprocessor_amd.cpp -- what a
#                                maintainer gets by applying INTEGRATION.md.  Runs on the GPU box.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${SIBELIA_REFERENCE:-/root/reference}"
SRC="$REF/src"
DSS="$SRC/libdivsufsort-2.0.1"
OUT="$HERE/_ref"
[ -d "$SRC" ] || { echo "no reference at $REF" >&2; exit 3; }
[ -f "$OUT/include/divsufsort.h" ] || bash "$HERE/build_ref.sh" > /dev/null
[ -f "$ROOT/sibelia_amd/lib/libsibelia_amd.so" ] || { echo "build the product first (python -c 'import __graft_entry__ as g; g.build()')" >&2; exit 4; }
mkdir -p "$OUT/obj2"
CDEFS='-DHAVE_CONFIG_H=0 -DHAVE_INTTYPES_H=1 -DHAVE_STDDEF_H=1 -DHAVE_STDINT_H=1 -DHAVE_STDLIB_H=1 -DHAVE_STRING_H=1
 -DHAVE_STRINGS_H=1 -DHAVE_MEMORY_H=1 -DINLINE=inline -DPROJECT_VERSION_FULL="2.0.1"
 -D__STDC_CONSTANT_MACROS -D__STDC_FORMAT_MACROS -D__STDC_LIMIT_MACROS'
CFLAGS="-O3 -DNDEBUG -fomit-frame-pointer -w"
CXXFLAGS="-std=gnu++14 -O3 -DNDEBUG -w"
pids=()
for f in divsufsort sssort trsort utils; do
  gcc $CFLAGS $CDEFS -I"$DSS/include" -I"$OUT/include" -c "$DSS/lib/$f.c" -o "$OUT/obj2/dss_$f.o" & pids+=($!)
done
BF="blockfinder bulgeremoval edge serialization synteny"                      # every definition of a BlockFinder member lives in these
REST="sibelia indexedsequence bifurcationstorage dnasequence fasta platform stranditerator vertexenumeration blockinstance util postprocessor outputgenerator resource"
for f in $BF $REST; do
  g++ $CXXFLAGS -I"$SRC/include" -I"$OUT/include" -c "$SRC/$f.cpp" -o "$OUT/obj2/$f.o" & pids+=($!)
done
g++ $CXXFLAGS -I"$SRC" -I"$SRC/include" -I"$OUT/include" -I"$ROOT/include" -c "$ROOT/integration/blockfinder_amd.cpp" -o "$OUT/obj2/blockfinder_amd.o" & pids+=($!)
g++ $CXXFLAGS -I"$SRC" -I"$SRC/include" -I"$OUT/include" -I"$ROOT/include" -c "$ROOT/integration/postprocessor_amd.cpp" -o "$OUT/obj2/postprocessor_amd.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# integration/postprocessor_amd.cpp replaces ONE member of the reference's postprocessor.cpp: that definition becomes weak in a copy of the object
GLUE=$(nm "$OUT/obj2/postprocessor.o" | awk '$2 == "T" && $3 ~ /Postprocessor11GlueStripes/ { print $3 }')
[ -n "$GLUE" ] || { echo "Postprocessor::GlueStripes not found in postprocessor.o" >&2; exit 5; }
objcopy --weaken-symbol="$GLUE" "$OUT/obj2/postprocessor.o" "$OUT/obj2/postprocessor_weak.o"
objs() { for f in "$@"; do echo "$OUT/obj2/$f.o"; done; }
g++ -O3 $(objs $BF $REST) "$OUT"/obj2/dss_*.o -o "$OUT/sibelia_ref"
REST_DROPIN=$(for f in $REST; do [ $f = postprocessor ] && echo postprocessor_weak || echo $f; done)
g++ -O3 "$OUT/obj2/blockfinder_amd.o" "$OUT/obj2/postprocessor_amd.o" $(objs $REST_DROPIN) "$OUT"/obj2/dss_*.o -L"$ROOT/sibelia_amd/lib" -lsibelia_amd -Wl,-rpath,'$ORIGIN/../../sibelia_amd/lib' -o "$OUT/sibelia_dropin"
rm -rf "$OUT/obj2"
echo "built $OUT/sibelia_ref $OUT/sibelia_dropin"
