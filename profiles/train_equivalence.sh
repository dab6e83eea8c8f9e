#!/bin/bash
# Many-step equivalence of the arithmetic modes: the same CLI run (synthetic particles, CTF + mask, 30 epochs) in the
# default mode and with TVAE_GEMM=f32; prints the last log lines of both (ELBO trajectories must agree to ~1e-5).
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np
rng = np.random.RandomState(0)
np.save('gpurun_out/stack2.npy', rng.randn(64, 32, 32).astype(np.float32))
PY
cd target-vae_amd
for mode in x6 f32; do
  TVAE_GEMM=$mode python train_particles.py --train-path ../gpurun_out/stack2.npy --normalize \
    --encoder-kernel-size 32 --encoder-padding 8 --encoder-kernel-number 32 --generator-hidden-dim 512 \
    --num-epochs 30 --minibatch-size 32 --seed 0 --log-root ../gpurun_out/logs_eq_$mode > ../gpurun_out/eq_$mode.log 2>&1
  echo "== $mode"; grep -P "^\d+\ttrain" ../gpurun_out/eq_$mode.log | tail -3
done
