#!/bin/bash
# TEST INFRASTRUCTURE — CMake-free build of the UNMODIFIED reference's BlockFinder path into oracle/_ref/.
#
# Compiles the reference's own sources WHERE THEY LIE under /root/reference (nothing is copied into the
# repo) with gcc/g++ directly; outputs go to oracle/_ref/ only (git-ignored, travels to the GPU box):
#   oracle/_ref/ref_dump     driver (oracle/ref_dump.cpp) + reference objects: dumps enumeration,
#                            post-stage state, DOT text and synteny blocks of the reference
#   oracle/_ref/build.txt    what was compiled, with which flags
#
# The reference's build (src/CMakeLists.txt:3-11, libdivsufsort-2.0.1/{CMakeLists.txt,include/CMakeLists.txt,
# lib/CMakeLists.txt}) does three things this script restates:
#  1. `configure_file(divsufsort.h.cmake divsufsort.h @ONLY)` with the values its type checks produce on
#     x86-64 Linux/glibc (include/CMakeLists.txt:15-16,58-66,112-128):  W64BIT="", INCFILE="#include <inttypes.h>",
#     DIVSUFSORT_EXPORT="", DIVSUFSORT_IMPORT="", SAUCHAR_TYPE=uint8_t, SAINT32_TYPE=int32_t, SAINDEX_TYPE=int32_t,
#     SAINT_PRId=PRId32, SAINDEX_PRId=PRId32.  Done below with sed on the reference's own template; the result is
#     the only generated file (oracle/_ref/include/divsufsort.h).
#  2. config.h: only feature macros; divsufsort_private.h:34-60 takes them from the command line when
#     HAVE_CONFIG_H is 0, so they are passed as -D flags (same values CMake's checks give here).
#  3. compiles lib/{divsufsort,sssort,trsort,utils}.c with `-O3 -DNDEBUG -fomit-frame-pointer` and the
#     reference's .cpp files with `-O3 -DNDEBUG` (CMAKE_BUILD_TYPE Release).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${SIBELIA_REFERENCE:-/root/reference}"
SRC="$REF/src"
DSS="$SRC/libdivsufsort-2.0.1"
OUT="$HERE/_ref"
[ -d "$SRC" ] || { echo "no reference at $REF" >&2; exit 3; }
mkdir -p "$OUT/include" "$OUT/obj"

sed -e 's/@W64BIT@//g' -e 's/@INCFILE@/#include <inttypes.h>/' \
    -e 's/@DIVSUFSORT_EXPORT@//' -e 's/@DIVSUFSORT_IMPORT@//' \
    -e 's/@SAUCHAR_TYPE@/uint8_t/' -e 's/@SAINT32_TYPE@/int32_t/' -e 's/@SAINDEX_TYPE@/int32_t/' \
    -e 's/@SAINT_PRId@/PRId32/' -e 's/@SAINDEX_PRId@/PRId32/' \
    "$DSS/include/divsufsort.h.cmake" > "$OUT/include/divsufsort.h"

CDEFS='-DHAVE_CONFIG_H=0 -DHAVE_INTTYPES_H=1 -DHAVE_STDDEF_H=1 -DHAVE_STDINT_H=1 -DHAVE_STDLIB_H=1 -DHAVE_STRING_H=1
 -DHAVE_STRINGS_H=1 -DHAVE_MEMORY_H=1 -DINLINE=inline -DPROJECT_VERSION_FULL="2.0.1"
 -D__STDC_CONSTANT_MACROS -D__STDC_FORMAT_MACROS -D__STDC_LIMIT_MACROS'
CFLAGS="-O3 -DNDEBUG -fomit-frame-pointer -w"
CXXFLAGS="-std=gnu++14 -O3 -DNDEBUG -w"
pids=()
for f in divsufsort sssort trsort utils; do
  gcc $CFLAGS $CDEFS -I"$DSS/include" -I"$OUT/include" -c "$DSS/lib/$f.c" -o "$OUT/obj/dss_$f.o" & pids+=($!)
done
# the reference's translation units the BlockFinder class, the post-processor and the writers need (src/CMakeLists.txt:11 minus main and the unit test)
UNITS="indexedsequence blockfinder bifurcationstorage bulgeremoval dnasequence edge fasta serialization synteny
 platform stranditerator vertexenumeration blockinstance util postprocessor outputgenerator resource"
for f in $UNITS; do
  g++ $CXXFLAGS -I"$SRC/include" -I"$OUT/include" -c "$SRC/$f.cpp" -o "$OUT/obj/$f.o" & pids+=($!)
done
g++ $CXXFLAGS -I"$SRC" -I"$SRC/include" -I"$OUT/include" -c "$HERE/ref_dump.cpp" -o "$OUT/obj/ref_dump.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
g++ -O3 "$OUT/obj/ref_dump.o" $(for f in $UNITS; do echo "$OUT/obj/$f.o"; done) \
    "$OUT"/obj/dss_*.o -o "$OUT/ref_dump"
{ echo "reference: $REF (bioinf/Sibelia 3.0.7, unmodified sources compiled in place)";
  echo "gcc: $(gcc --version | head -1)"; echo "CFLAGS: $CFLAGS"; echo "CXXFLAGS: $CXXFLAGS"; echo "units: $UNITS"; } > "$OUT/build.txt"
rm -rf "$OUT/obj"
echo "built $OUT/ref_dump"
