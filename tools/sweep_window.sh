#!/bin/bash
# sweep of the BASE window of the ordered rounds on the default bench workload (performance knob only; results are window independent;
# the driver still widens the window while a round is capacity-bound).  Usage: tools/sweep_window.sh 12288 16384 24576
for w in "$@"; do
  SBL_BASE_WINDOW=$w timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | W=$w python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ['W'], round(d['ms_per_step'],2), d['config']['rounds'], d['config']['replays'], {k:round(v,1) for k,v in d['phase_ms'].items()})"
done
