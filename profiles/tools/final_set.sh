#!/bin/bash
# The whole judged profile set of a round in one call (about 12 minutes of GPU time):
#   gpurun --timeout 1200 -- 'bash profiles/tools/final_set.sh r06_d'
# -> gpurun_out/<tag>/: bench line, kernel stats, FETCH / WRITE passes (collect.sh), SQ counters and instruction mix of the step
#    (sq_kernels.sh, sq_issue.sh), one-step timeline, per-kernel stats of the other workloads, SQ counters of the galaxy step.
# Copy the files into profiles/ afterwards (names already carry the tag).
set -u
TAG=${1:-r00_x}
D=$PWD/gpurun_out/$TAG
mkdir -p "$D"
bash profiles/collect.sh "$TAG" > "$D/collect.log" 2>&1 && echo "collect done"
bash profiles/tools/sq_kernels.sh "$TAG" > /dev/null 2>&1 && cp "gpurun_out/sq_$TAG/sq_counters.txt" "$D/${TAG}_sq_counters.txt" && echo "sq done"
bash profiles/tools/sq_issue.sh "$TAG" > /dev/null 2>&1 && cp "gpurun_out/sqi_$TAG/sq_issue.txt" "$D/${TAG}_sq_issue.txt" && echo "sq_issue done"
bash profiles/tools/step_timeline.sh S64 > /dev/null 2>&1 && cp gpurun_out/timeline_S64.txt "$D/${TAG}_step_timeline.txt" && echo "timeline done"
for WL in S28 S28F S128G; do
  bash profiles/tools/quick_stats_wl.sh $WL 3 > "$D/qs_$WL.txt" 2>&1 && cp "gpurun_out/qs_$WL/kernel_stats.csv" "$D/${TAG}_${WL}_kernel_stats.csv" && echo "stats $WL done"
done
bash profiles/tools/sq_kernels.sh "${TAG}g" 'tvae::' S128G > /dev/null 2>&1 && cp "gpurun_out/sq_${TAG}g/sq_counters.txt" "$D/${TAG}_S128G_sq_counters.txt" && echo "sq S128G done"
ls "$D"
