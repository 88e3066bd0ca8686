#!/bin/bash
# One GPU-box session: dense-vector timings, GPU parity suite, default bench.  usage: tools/gpu_round.sh <tag> [skip-tests]
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python tools/dense_vectors.py 800 8000 > $out/dense_huge.log 2>&1
if [ -z "$2" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "pytest rc $?" >> $out/gpu_tests.log; fi
timeout 900 python bench.py --no-cpu-full > $out/bench.log 2>&1
tail -5 $out/gpu_tests.log; grep '^{' $out/bench.log | cut -c1-2500; cat $out/dense_huge.log
