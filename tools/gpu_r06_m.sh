#!/bin/bash
# round 6: stress of the exact verdict-table bound (k_probe_idx): many strains with the default tables, with a 256-slot table (the tight path
# on nearly every live entry), plain cases with the 256-slot table; then config 4 and the headline on the final build
export TMPDIR=/tmp
echo "== MANY, default tables"; MANY=1 timeout 500 python tools/stress.py 240 91000 2>&1 | tail -3
echo "== MANY, 256 slots"; SBL_PIDX_VBITS=8 MANY=1 timeout 500 python tools/stress.py 240 92000 2>&1 | tail -3
echo "== plain, 256 slots"; SBL_PIDX_VBITS=8 timeout 400 python tools/stress.py 180 93000 2>&1 | tail -3
echo "== config 4"; timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | cut -c1-330
