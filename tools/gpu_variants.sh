#!/bin/bash
# default bench line with each variant library (tools/build_variants.sh) copied over the product library on the GPU box's scratch copy
# usage: tools/gpu_variants.sh <tag> name1 name2 ...     ("base" = the library as built)
tag=${1:-var}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag; mkdir -p $out; cd $R
cp sibelia_amd/lib/libsibelia_amd.so /tmp/base.so
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so sibelia_amd/lib/libsibelia_amd.so; else cp sibelia_amd/lib/var_$v.so sibelia_amd/lib/libsibelia_amd.so; fi
  touch sibelia_amd/lib/libsibelia_amd.so
  timeout 600 python bench.py --no-cpu-baseline --steps 5 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2), d.get('matches_reference_fixture'), {k: round(x,2) for k,x in d.get('phase_ms',{}).items() if k in ('commit_ms','simplify_ms')})" | tee -a $out/variants.log
done; done
cp /tmp/base.so sibelia_amd/lib/libsibelia_amd.so
