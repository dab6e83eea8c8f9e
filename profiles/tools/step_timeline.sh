#!/bin/bash
# Timeline of ONE training step of the default bench (every kernel dispatch in start order with its duration and the idle gap
# before it):  gpurun -- 'bash profiles/tools/step_timeline.sh [workload]'  ->  gpurun_out/timeline_<wl>.txt
set -u
export TMPDIR=/tmp
WL=${1:-S64}
OUT=$PWD/gpurun_out/tl_$WL
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -o k -- python3 bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-f32-companion --no-workloads --no-small-batch > "$OUT/log.txt" 2>&1
F=$(find "$OUT/t" -name '*kernel_trace.csv' | head -1)
python3 - "$F" > "$PWD/gpurun_out/timeline_$WL.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# one step = from one adam_flat_kernel to the next; take the last complete one
idx = [i for i, r in enumerate(rows) if 'adam_flat_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
prev_end = int(rows[a - 1]['End_Timestamp'])
busy = gap = 0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = max(0, s - prev_end)
    busy += e - s
    gap += g
    print('%9.1f us  dur %8.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, g / 1e3, r['Kernel_Name'][:110]))
    prev_end = max(prev_end, e)
print('kernels %d  busy %.3f ms  gaps %.3f ms  span %.3f ms' % (len(step), busy / 1e6, gap / 1e6, (prev_end - t0) / 1e6))
PY
rm -rf "$OUT/t"
tail -1 "$PWD/gpurun_out/timeline_$WL.txt"
