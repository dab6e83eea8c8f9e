#!/bin/bash
# Per-kernel averages of the bench step at another batch size (the strong-scaling proxy: 32 images = 256 / 8 GPUs):
#   gpurun -- 'bash profiles/tools/quick_stats_batch.sh 32'
set -u
export TMPDIR=/tmp
B=${1:-32}
OUT=$PWD/gpurun_out/qsb_$B
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-f32-companion --no-workloads > "$OUT/stats.log" 2>&1
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$S" "$OUT/kernel_stats.csv"; rm -rf "$OUT/stats"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step %.3f ms' % (tot / 7 / 1e6))
for r in rows[:26]:
    print('%-72s n %4s avg %8.1f us %5.1f%%' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
