#!/bin/bash
# Per-kernel time of one of the extra workloads (bench.py --workload): gpurun -- 'bash profiles/profile_workload.sh S28F'
set -u
export TMPDIR=/tmp
WL=${1:-S28}
OUT=$PWD/gpurun_out/wl_$WL
mkdir -p "$OUT"
python3 bench.py --workload "$WL" --no-cpu-baseline > "$OUT/bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --workload "$WL" --steps 5 --warmup 2 --no-cpu-baseline --no-f32-companion > "$OUT/stats.log" 2>&1
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$S" "$OUT/${WL}_kernel_stats.csv"
rm -rf "$OUT/stats"
tail -1 "$OUT/bench.log" | cut -c1-400
head -25 "$OUT/${WL}_kernel_stats.csv" | cut -d, -f1-4 | cut -c1-160
