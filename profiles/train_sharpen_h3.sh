#!/bin/bash
# Long training in the regime that stresses h3 (VERDICT r03 missing #6): STRUCTURED images -- an asymmetric two-stroke glyph,
# randomly rotated and translated -- so that the reference's own training sharpens the attention posterior (the noise images
# of train_equivalence_h3.sh leave it flat: KL 0.18).  Full widths (128 kernels, hidden 512), the 44-wide ring frame, the real
# `--dataset mnist-U` code path of the CLI (files written below), 2 048 + 256 images, 40 epochs x 32 minibatches of 64 =
# 1 280 optimizer steps, in the three fp32-equivalent arithmetics.  Prints the last test lines (ELBO, -log p, KL).
set -e
cd "$(dirname "$0")/../target-vae_amd"
mkdir -p ../gpurun_out data/mnist_U
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(0)
def glyphs(n, side=28):
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float32)
    out = np.zeros((n, side, side), np.float32)
    for i in range(n):
        th = rng.uniform(0, 2 * np.pi)
        cx, cy = side / 2 + rng.uniform(-4, 4), side / 2 + rng.uniform(-4, 4)
        c, s = np.cos(th), np.sin(th)
        u = (xx - cx) * c + (yy - cy) * s            # glyph frame
        v = -(xx - cx) * s + (yy - cy) * c
        long_bar = np.exp(-(v / 1.3) ** 2) * (np.abs(u) < 7)                     # a bar ...
        hook = np.exp(-((u - 6) / 1.3) ** 2) * ((v > 0) & (v < 5))               # ... with a hook at one end (no symmetry)
        dot = np.exp(-(((u + 5) / 1.5) ** 2 + ((v + 3.5) / 1.5) ** 2))           # and a dot on the other side
        out[i] = np.clip(long_bar + hook + dot, 0, 1)
    return (out * 255).astype(np.uint8)
np.save('data/mnist_U/images_train.npy', glyphs(2048))
np.save('data/mnist_U/images_test.npy', glyphs(256))
PY
run() {
  TVAE_GEMM=$1 python train_mnist.py --dataset mnist-U --image-dim 28 -z 2 --minibatch-size 64 --num-epochs 40 \
    --save-interval 100 --encoder-kernel-number 128 --generator-hidden-dim 512 --encoder-kernel-size 28 --encoder-padding 8 \
    --seed 0 --log-root ../gpurun_out/logs_sharp_$1 > ../gpurun_out/sharp_$1.log 2>&1
  echo "== $1"; grep -aoP "^\d+\t(train|test)\t\S+\t\S+\t\S+" ../gpurun_out/sharp_$1.log | awk 'NR==2 || NR==20 || NR==40' ; grep -aoP "^\d+\t(train|test)\t\S+\t\S+\t\S+" ../gpurun_out/sharp_$1.log | tail -3
}
for m in ${MODES:-h3 x6 f32}; do run $m; done        # MODES=h3 bash profiles/train_sharpen_h3.sh: one arithmetic only
rm -rf data/mnist_U
