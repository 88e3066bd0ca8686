#!/bin/bash
# round 6, verdict-table bound of k_probe_idx: config 4 (62 strains) with the exact bound, with a 1024-slot table; the headline; parity
mkdir -p gpurun_out/l
export TMPDIR=/tmp
for v in "" 10; do
  echo "== config 4 vbits=${v:-9}"
  SBL_PIDX_VBITS=$v SBL_TEST_FLAGS=32 timeout 600 python bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "block index|reservations|^\{" | cut -c1-600
  SBL_PIDX_VBITS=$v timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | cut -c1-300
done
echo "== headline"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "^\{" | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parking.py -x -q 2>&1 | tail -5
