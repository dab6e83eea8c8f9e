#!/bin/bash
# SQ counters of the kernels of the default bench step (ONE --pmc pass, kernel-trace only; never with sys / runtime tracing):
#   gpurun -- 'bash profiles/tools/sq_kernels.sh <tag> [name regex] [workload]'
# Prints, per kernel (mean over its launches): duration, effective clock (GRBM_GUI_ACTIVE / 8 / wall), the shares of
# wave-cycles spent parked (SQ_WAIT_ANY: s_waitcnt / barrier), issue-stalled (SQ_WAIT_INST_ANY), issuing
# (SQ_ACTIVE_INST_ANY), and the MFMA-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (clock cycles x 1024 SIMDs)).
set -u
export TMPDIR=/tmp
TAG=${1:-x}; PAT=${2:-tvae::}; WL=${3:-S64}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/p" -o q -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-f32-companion --no-workloads > "$OUT/log.txt" 2>&1
F=$(find "$OUT/p" -name '*counter_collection.csv' | head -1)
python3 - "$F" "$PAT" <<'PY' | tee "$OUT/sq_counters.txt"
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if not re.search(sys.argv[2], r['Kernel_Name']):
        continue
    k = re.sub(r'\(.*', '', r['Kernel_Name'])[:64] + ' g' + r['Grid_Size']
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    ms = sum(dur[k]) / len(dur[k])
    rows.append((ms * len(dur[k]), k, m, ms, len(dur[k])))
for _, k, m, ms, n in sorted(rows, reverse=True)[:24]:
    wc = m.get('SQ_WAVE_CYCLES', 1)
    gui = m.get('GRBM_GUI_ACTIVE', 0)
    clk = gui / 8 / (ms * 1e-3) / 1e9
    mf = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    print(f"{k}  x{n}\n   {ms:7.3f} ms  clk {clk:5.2f} GHz  wait_any {m.get('SQ_WAIT_ANY',0)/wc:5.2f}  wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:5.2f}"
          f"  active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f}  wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:5.2f}"
          f"  mfma_busy {mf / max(gui / 8 * 1024, 1):5.3f}  waves/SIMD {wc * 4 / max(gui / 8 * 1024, 1):5.2f}")
PY
rm -rf "$OUT/p"
