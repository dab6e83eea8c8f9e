#!/bin/bash
# CLI smoke of train_particles.py with CTF filters and circular mask on synthetic files (GPU box).
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np
rng = np.random.RandomState(0)
np.save('gpurun_out/stack.npy', rng.randn(40, 32, 32).astype(np.float32))
rows = np.stack([rng.uniform(1, 3, 40), np.full(40, 2.7), np.full(40, 300.), np.full(40, 1.2), rng.uniform(50, 150, 40),
                 np.full(40, 7.), np.zeros(40), rng.uniform(0, 180, 40)], 1)
np.savetxt('gpurun_out/ctf.txt', rows)
PY
cd target-vae_amd
python train_particles.py --train-path ../gpurun_out/stack.npy --ctf-train ../gpurun_out/ctf.txt --normalize \
  --mask-radius 12 --encoder-kernel-size 32 --encoder-padding 8 --encoder-kernel-number 16 --generator-hidden-dim 64 \
  --num-epochs 2 --minibatch-size 16 --seed 0 --log-root ../gpurun_out/logs_particles
