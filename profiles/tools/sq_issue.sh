#!/bin/bash
# Instruction mix / issue-slot counters of the big kernels of the default bench step ("report what the
# instruction stalls are before changing tiles").  Two --pmc passes (kernel-trace only; never with sys / runtime tracing):
#   gpurun -- 'bash profiles/tools/sq_issue.sh <tag> [name regex] [workload]'
# Per kernel (mean over its launches): instructions per wave by class, VALU instructions per MFMA, the shares of wave-cycles in
# which a wave had a VALU / LDS / VMEM / scalar instruction in flight, MFMA-busy and MFMA+VALU co-execution cycles.
set -u
export TMPDIR=/tmp
TAG=${1:-x}; PAT=${2:-dense_x6_kernel|dense_wgrad_x6_dma|xres|dft_out_ring|dft_dy_ring|enc_tail}; WL=${3:-S64}
OUT=$PWD/gpurun_out/sqi_$TAG
mkdir -p "$OUT"
CMD="python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-f32-companion --no-workloads --no-small-batch"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM -d "$OUT/a" -o q -- $CMD > "$OUT/log_a.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT -d "$OUT/b" -o q -- $CMD > "$OUT/log_b.txt" 2>&1
python3 - "$OUT" "$PAT" <<'PY' | tee "$OUT/sq_issue.txt"
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if not re.search(sys.argv[2], r['Kernel_Name']):
            continue
        k = re.sub(r'\(.*', '', r['Kernel_Name'])[:72] + ' g' + r['Grid_Size']
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w, wc = max(m.get('SQ_WAVES', 1), 1), max(m.get('SQ_WAVE_CYCLES', 1), 1)
    mf = max(m.get('SQ_INSTS_MFMA', 0), 1e-9)
    print(k)
    print('   per wave: VALU %8.0f  MFMA %7.0f  LDS %7.0f  SALU %7.0f  VMEM %6.0f  SMEM %5.0f   | VALU (non-MFMA) per MFMA %5.1f   LDS per MFMA %4.2f'
          % (m.get('SQ_INSTS_VALU', 0) / w, mf / w, m.get('SQ_INSTS_LDS', 0) / w, m.get('SQ_INSTS_SALU', 0) / w,
             m.get('SQ_INSTS_VMEM', 0) / w, m.get('SQ_INSTS_SMEM', 0) / w, (m.get('SQ_INSTS_VALU', 0) - mf) / mf, m.get('SQ_INSTS_LDS', 0) / mf))
    print('   wave-cycle shares: VALU in flight %.2f  LDS %.2f  VMEM %.2f  scalar %.2f   | MFMA busy / coexec with VALU (cycles, x1e6): %.1f / %.1f   LDS bank-conflict cycles x1e6: %.2f'
          % (m.get('SQ_ACTIVE_INST_VALU', 0) / wc, m.get('SQ_ACTIVE_INST_LDS', 0) / wc, m.get('SQ_ACTIVE_INST_VMEM', 0) / wc,
             m.get('SQ_ACTIVE_INST_SCA', 0) / wc, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1e6, m.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / 1e6,
             m.get('SQ_LDS_BANK_CONFLICT', 0) / 1e6))
PY
rm -rf "$OUT/a" "$OUT/b"
