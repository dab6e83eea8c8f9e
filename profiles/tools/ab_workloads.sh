#!/bin/bash
for wl in S28 S28F S128G; do
  for v in 0 1; do
    TVAE_FUSE_ENC_TAIL=$v python3 bench.py --workload $wl --no-cpu-baseline --no-f32-companion --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl FUSE=$v', '%.3f ms/step' % d['ms_per_step'], '%.0f img/s' % d['value'])"
  done
done
