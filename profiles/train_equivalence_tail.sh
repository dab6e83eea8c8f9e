#!/bin/bash
# Many-step equivalence with the fused encoder tail in the loop: the same CLI run (synthetic 32x32 particles, 128 encoder
# channels, 30 epochs x 2 minibatches) (a) default arithmetic, fused tail, (b) default arithmetic, unfused tail
# (TVAE_FUSE_ENC_TAIL=0), (c) exact fp32 products (TVAE_GEMM=f32).  Prints the last train lines of the three logs.
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np
rng = np.random.RandomState(0)
np.save('gpurun_out/stack2.npy', rng.randn(64, 32, 32).astype(np.float32))
PY
cd target-vae_amd
run() {
  python train_particles.py --train-path ../gpurun_out/stack2.npy --normalize \
    --encoder-kernel-size 32 --encoder-padding 8 --encoder-kernel-number 128 --generator-hidden-dim 512 \
    --num-epochs 30 --minibatch-size 32 --seed 0 --log-root ../gpurun_out/logs_eqt_$1 > ../gpurun_out/eqt_$1.log 2>&1
  echo "== $1"; grep -aoP "\d+\ttrain\t\S+\t\S+\t\S+" ../gpurun_out/eqt_$1.log | tail -3
}
TVAE_GEMM=x6 run x6_fused
TVAE_GEMM=x6 TVAE_FUSE_ENC_TAIL=0 run x6_unfused
TVAE_GEMM=f32 run f32
