#!/bin/bash
# LDS-side diagnosis (one --pmc pass, kernel-trace only):  gpurun -- 'bash profiles/pmc_lds.sh <ONLY-filter> <tag>'
set -u
export TMPDIR=/tmp ONLY=${1:-x6_conv1} REPS=2
OUT=$PWD/gpurun_out/lds_${2:-x}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/p" -o q -- python3 profiles/kernel_bench.py > "$OUT/log.txt" 2>&1
F=$(find "$OUT/p" -name '*counter_collection.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'tvae::' not in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'][:50] + ' g' + r['Grid_Size']
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    gui = m.get('GRBM_GUI_ACTIVE', 1) / 8          # cycles
    print(k)
    print('   ' + '  '.join(f'{n}={v:.3e}' for n, v in sorted(m.items())))
    print(f"   per CU-cycle: lds_idx_active {m.get('SQ_LDS_IDX_ACTIVE',0)/(gui*256):.3f}  bank_conflict {m.get('SQ_LDS_BANK_CONFLICT',0)/(gui*256):.3f}  unaligned {m.get('SQ_LDS_UNALIGNED_STALL',0)/(gui*256):.3f}")
PY
rm -rf "$OUT/p"
