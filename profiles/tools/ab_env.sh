#!/bin/bash
# Same-box A/B of one environment switch on the default bench step:  gpurun -- 'bash profiles/tools/ab_env.sh VAR valA valB [rounds]'
VAR=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v python3 bench.py --no-cpu-baseline --no-f32-companion --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', '%.3f ms/step' % d['ms_per_step'], {k: round(v,3) for k,v in d['roofline']['entry_points_ms'].items()})"
  done
done
