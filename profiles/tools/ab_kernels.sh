#!/bin/bash
# Per-kernel averages of the default bench step under several values of one environment switch (rocprofv3 --stats):
#   gpurun -- 'bash profiles/tools/ab_kernels.sh VAR "v1 v2 ..." "<grep pattern of kernel names>" [workload]'
set -u
export TMPDIR=/tmp
VAR=$1; VALS=$2; PAT=${3:-dft_}; WL=${4:-S64}
for v in $VALS; do
  export $VAR=$v
  OUT=$PWD/gpurun_out/abk_${VAR}_$v
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-f32-companion > "$OUT/stats.log" 2>&1
  S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
  echo "== $VAR=$v  $(python3 -c "import json,sys; d=json.loads([l for l in open('$OUT/stats.log') if l.startswith('{')][-1]); print('%.3f ms/step' % d['ms_per_step'])")"
  python3 - "$S" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        print('   %-70s n %4s avg %9.1f us  min %9.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
  rm -rf "$OUT/stats"
done
