cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
for cfg in "0.15 1.0 4" "0.15 0.5 4" "0.15 0.25 4" "0.1 0.5 4" "0.15 0.5 8" "0.08 1.0 8"; do
set -- $cfg
SBL_KEEP_BLOCKED=$1 SBL_WMIN=$2 SBL_WMAX=$3 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2o/b.log 2>&1; echo -n "keep $1 wmin $2 wmax $3: "; grep '^{' gpurun_out/r2o/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['config']['rounds'], {k:round(v,1) for k,v in d['phase_ms'].items() if k in ('commit_ms','probe_ms','reserve_ms')})"
done
