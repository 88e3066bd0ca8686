#!/bin/bash
# round 6, last stress: many strains with parking after every collapse and with the default cap of that regime (4); blocks + reports after random stages
export TMPDIR=/tmp
echo "== MANY cap 1"; SBL_PARK=1 MANY=1 timeout 300 python tools/stress.py 150 101000 2>&1 | tail -2
echo "== MANY default"; MANY=1 timeout 300 python tools/stress.py 150 102000 2>&1 | tail -2
echo "== blocks + reports"; N2=1 timeout 250 python tools/stress.py 120 103000 2>&1 | tail -2
