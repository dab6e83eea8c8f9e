#!/bin/bash
# Per-kernel averages of one bench workload: gpurun -- 'bash profiles/tools/quick_stats_wl.sh S128G [steps]'
set -u
export TMPDIR=/tmp
WL=${1:-S128G}; ST=${2:-3}
OUT=$PWD/gpurun_out/qs_$WL
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --no-f32-companion --no-workloads > "$OUT/stats.log" 2>&1
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$S" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
tail -1 "$OUT/stats.log" | cut -c1-400
python3 - "$OUT/kernel_stats.csv" $ST <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) + 1
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step %.3f ms' % (tot / n / 1e6))
for r in rows[:30]:
    print('%-88s n %4s avg %9.1f us %5.1f%%' % (r['Name'][:88], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
