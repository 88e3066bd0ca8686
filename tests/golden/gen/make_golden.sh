#!/bin/bash
# Build-container-only recipe that (re)creates tests/golden/vectors.json.
# 1. oracle/build_ref.sh compiles the UNMODIFIED reference sources in place (gcc/g++ directly, no CMake)
#    into oracle/_ref/ref_dump,
# 2. make_golden.py runs it on every case.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
bash "$HERE/../../../oracle/build_ref.sh"
python3 "$HERE/make_golden.py" "$@"
