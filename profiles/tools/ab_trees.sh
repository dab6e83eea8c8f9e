#!/bin/bash
# Same-box A/B of two TREES (different ABI versions cannot share the Python side: profiles/tools/ab_kernels.sh swaps only the
# library): per-kernel rocprofv3 averages of the default bench step in each tree, then the plain bench line of each.
#   gpurun -- 'bash profiles/tools/ab_trees.sh ab_r03 . [rounds]'      (ab_r03: `git worktree add ab_r03 <commit>` + make, git-ignored)
set -u
A=$1; B=$2; R=${3:-1}
ROOT=$PWD
for t in $A $B; do
  (cd $ROOT/$t && bash profiles/quick_stats.sh ab > /dev/null 2>&1; echo "== $t"; python3 - gpurun_out/qs_ab/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print('   %-72s n %3s avg %8.1f us' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  )
done
for r in $(seq $R); do
  for t in $A $B; do
    (cd $ROOT/$t && python3 bench.py --no-cpu-baseline --no-f32-companion --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', '%.3f ms/step' % d['ms_per_step'], {k: round(v,3) for k,v in d['roofline']['entry_points_ms'].items()})")
  done
done
