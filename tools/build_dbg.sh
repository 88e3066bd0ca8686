#!/bin/bash
# Debug build of the library for tools/diag_case.py: commit.hip with -DSBL_DBG_IDRET (collapses and parks per id, a log of every collapse),
# linked with the objects of the normal build into sibelia_amd/lib/libsibelia_amd_dbg.so.  usage: tools/build_dbg.sh; then on the GPU box
#   SBL_DBG_LIB=$PWD/sibelia_amd/lib/libsibelia_amd_dbg.so DBG_OUT=gpurun_out/a.npy python tools/diag_case.py SEED
# (run it once more with SBL_PARK=0 or another switch and compare the two logs: that is how seed 93194's lost bulge was traced to id 8125)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
mkdir -p /tmp/sbl_dbgobj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSBL_DBG_IDRET -c sibelia_amd/csrc/commit.hip -o /tmp/sbl_dbgobj/commit.o
hipcc --offload-arch=gfx950 -shared -fPIC -o sibelia_amd/lib/libsibelia_amd_dbg.so /tmp/sbl_dbgobj/commit.o $(ls sibelia_amd/lib/obj/*.o | grep -v /commit.o)
echo sibelia_amd/lib/libsibelia_amd_dbg.so
