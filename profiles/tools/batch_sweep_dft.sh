# Is T (resp. S') served from the Infinity Cache when it fits?  Per-kernel time AND HBM fetch of the frequency-domain
# convolution against the batch size, h3 arithmetic: B = 4 / 8 (T = 53 / 106 MB: fits the 256 MB cache), 16, 64, 256.
#   gpurun -- 'bash profiles/tools/batch_sweep_dft.sh [path/to/libtvae_hip.so]'       (second library: a -DTVAE_T_CACHED build,
#   default cache policy on the T store and load instead of nontemporal)
export TMPDIR=/tmp
for LIB in "" "$1"; do
  [ -z "$LIB" ] && [ -n "$1" ] && true
  echo "=== library: ${LIB:-shipped (nontemporal T store / load)}"
  for B in 4 8 16 64 256; do
    rm -rf /tmp/mt_$B /tmp/mf_$B
    TVAE_LIB=$LIB B=$B ONLY=dft_conv1 REPS=5 MODE=h3 PARTS=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt_$B -o s -- python3 profiles/kernel_bench.py > /tmp/mt_$B.log 2>&1
    TVAE_LIB=$LIB B=$B ONLY=dft_conv1 REPS=2 MODE=h3 PARTS=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/mf_$B -o f -- python3 profiles/kernel_bench.py > /tmp/mf_$B.log 2>&1
    S=$(find /tmp/mt_$B -name '*kernel_stats.csv' | head -1)
    F=$(find /tmp/mf_$B -name '*counter_collection.csv' | head -1)
    echo "B=$B"; python3 - "$S" "$F" $B <<'PY'
import csv, sys
from collections import defaultdict
B = int(sys.argv[3])
fetch = defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if r['Counter_Name'] == 'FETCH_SIZE':
        fetch[r['Kernel_Name'][:40]].append(float(r['Counter_Value']) * 1024 * 2)      # KiB -> bytes, x2 (gfx950 note)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('dft_out', 'dft_dy', 'plain4', 'dense_x6_kernel<0', 'dense_wgrad_x6_dma')):
        f = fetch.get(n[:40], [0])
        print('   %-44s avg %8.1f us  %6.2f us/img   HBM fetch %8.1f MB  %6.2f MB/img' %
              (n[:44], float(r['AverageNs']) / 1e3, float(r['AverageNs']) / 1e3 / B, sum(f) / len(f) / 1e6, sum(f) / len(f) / 1e6 / B))
PY
  done
  [ -z "$1" ] && break
done
