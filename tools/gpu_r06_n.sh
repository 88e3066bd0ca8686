#!/bin/bash
# round 6, after the parking fix: stress where parked transactions meet neighbours (PARKY=1), default cap and cap 1; three virtual ranks; three-stage cascades
export TMPDIR=/tmp
echo "== PARKY default cap"; PARKY=1 timeout 400 python tools/stress.py 240 96000 2>&1 | tail -2
echo "== PARKY cap 1"; SBL_PARK=1 PARKY=1 timeout 400 python tools/stress.py 240 97000 2>&1 | tail -2
echo "== PARKY cap 3, 3 virtual ranks"; SBL_PARK=3 SHARD=3 PARKY=1 timeout 300 python tools/stress.py 150 98000 2>&1 | tail -2
echo "== three-stage cascades"; STAGES=3 timeout 300 python tools/stress.py 150 99000 2>&1 | tail -2
