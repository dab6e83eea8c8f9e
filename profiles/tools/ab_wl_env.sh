#!/bin/bash
# Same-box A/B of one environment switch on one bench workload:
#   gpurun -- 'bash profiles/tools/ab_wl_env.sh S128G TVAE_CONV_DFT "0 1" [batch]'
WL=$1; VAR=$2; VALS=$3; BATCH=${4:-}
for rep in 1 2; do
  for v in $VALS; do
    export $VAR=$v
    python3 bench.py --workload $WL ${BATCH:+--batch $BATCH} --no-cpu-baseline --no-f32-companion --no-workloads --no-small-batch --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL ${BATCH:+B=$BATCH }$VAR=$v', '%.3f ms/step' % d['ms_per_step'], '%.0f img/s' % d['value'], 'elbo', d.get('elbo'))"
  done
done
