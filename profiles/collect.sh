#!/bin/bash
# Collect the judged artifacts of one round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r01_d'
# 1. plain bench line; 2. rocprofv3 --kernel-trace --stats of the same command; 3./4. FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (never combined with sys/runtime tracing).  Outputs land in gpurun_out/<tag>/; copy the
# summaries into profiles/ afterwards (see the end of this script).
set -u
TAG=${1:-r01_x}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" > "$OUT/${TAG}_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-companion --no-workloads > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-companion --no-workloads > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-companion --no-workloads > "$OUT/write.log" 2>&1
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
cp "$S" "$OUT/${TAG}_bench_kernel_stats.csv"
python3 profiles/summarize_pmc.py "$F" "$W" "$OUT/pmc_traffic.json" | tee "$OUT/pmc_summary.txt"
# keep the per-kernel counter rows small enough to commit: only tvae kernels, one line per launch
head -1 "$F" > "$OUT/${TAG}_pmc_fetch_size.csv"; grep 'tvae::' "$F" >> "$OUT/${TAG}_pmc_fetch_size.csv"
head -1 "$W" > "$OUT/${TAG}_pmc_write_size.csv"; grep 'tvae::' "$W" >> "$OUT/${TAG}_pmc_write_size.csv"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
tail -1 "$OUT/bench.log"
