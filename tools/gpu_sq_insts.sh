#!/bin/bash
# instruction mix of the round kernels (through gpurun): usage tools/gpu_sq_insts.sh <tag> [env settings]   -> gpurun_out/<tag>/insts.txt
tag=${1:-sqi}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM_RD --output-format csv -d $out/sqi -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sqi.log 2>&1
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sqj -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sqj.log 2>&1
cd $R
python - $out <<'PY' > $out/insts.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
for sub in ("sqi", "sqj"):
    fs = sorted(glob.glob(os.path.join(out, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not fs: print(sub, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith(("k_commit", "k_reserve", "k_probe_idx")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, c in agg.items():
        print(k, "launches", len(n[k]), " ".join("%s=%.0f" % (a, b / len(n[k])) for a, b in sorted(c.items())))
PY
cat $out/insts.txt
find $out -name '*_kernel_trace.csv' -size +20M -delete; find $out -name '*_counter_collection.csv' -size +20M -delete
