# Per-kernel time of the frequency-domain convolution against the batch size (is T served from the Infinity Cache when it fits?)
export TMPDIR=/tmp
for B in 16 32 64 256; do
  rm -rf /tmp/mt_$B
  B=$B ONLY=dft_conv1 REPS=3 MODE=x6 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt_$B -o s -- python3 profiles/kernel_bench.py > /tmp/mt_$B.log 2>&1
  S=$(find /tmp/mt_$B -name '*kernel_stats.csv' | head -1)
  echo "B=$B"; python3 - "$S" $B <<'PY'
import csv, sys
B = int(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('dft_out_mf', 'dft_dy_mf', 'dense_x6_kernel<0>', 'dense_wgrad_x6_dma', 'dft_image', 'dft_dbank', 'dft_bank', 'splitk')):
        print('   %-60s avg %9.1f us   %7.2f us/image' % (n[:60], float(r['AverageNs']) / 1e3, float(r['AverageNs']) / 1e3 / B))
PY
done
