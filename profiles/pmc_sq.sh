#!/bin/bash
# SQ-level diagnosis of the GEMM-shaped kernels (one --pmc pass, kernel-trace only):
#   gpurun -- 'bash profiles/pmc_sq.sh <ONLY-filter> <tag>'
set -u
export TMPDIR=/tmp ONLY=${1:-dec} REPS=2 MODE=x6
OUT=$PWD/gpurun_out/sq_${2:-x}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/p" -o q -- python3 profiles/kernel_bench.py > "$OUT/log.txt" 2>&1
F=$(find "$OUT/p" -name '*counter_collection.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'tvae::' not in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'][:56] + ' g' + r['Grid_Size']
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    ms = sum(dur[k]) / len(dur[k])
    clk = m.get('GRBM_GUI_ACTIVE', 0) / 8 / (ms * 1e-3) / 1e9
    mf = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    print(f"{k}\n   {ms:7.3f} ms  clk {clk:5.2f} GHz  wait_any {m.get('SQ_WAIT_ANY',0)/wc:5.2f}  wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:5.2f}"
          f"  active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f}  wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:5.2f}"
          f"  mfma_busy/(gui*4simd*32cu) {mf/max(m.get('GRBM_GUI_ACTIVE',1)*4*32/1,1):6.3f}  mfma_busy {mf:.3e} gui {m.get('GRBM_GUI_ACTIVE',0):.3e} sqbusy {m.get('SQ_BUSY_CYCLES',0):.3e}")
PY
rm -rf "$OUT/p"
