#!/bin/bash
# window-size sweep of the default bench workload (performance knob only; results are window independent)
for w in "$@"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --window $w 2>/dev/null | W=$w python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ['W'], round(d['ms_per_step'],1), d['config']['rounds'], d['config']['replays'], {k:round(v,1) for k,v in d['phase_ms'].items()})"
done
