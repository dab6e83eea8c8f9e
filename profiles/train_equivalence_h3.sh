#!/bin/bash
# Many-step equivalence of the three fp32-equivalent arithmetics with EVERY h3 instance in the loop: the same CLI run
# (train_mnist.py on 256 synthetic 28x28 images: the 44-wide ring frame, 128 encoder channels, hidden 512; 40 epochs x 4
# minibatches of 64 = 160 optimizer steps) with TVAE_GEMM=h3 (default), x6 and f32.  Prints the last train / test lines.
set -e
cd "$(dirname "$0")/../target-vae_amd"
mkdir -p ../gpurun_out
run() {
  TVAE_GEMM=$1 python train_mnist.py --dataset mnist-U --synthetic 256 --image-dim 28 -z 2 --minibatch-size 64 --num-epochs 40 \
    --save-interval 100 --encoder-kernel-number 128 --generator-hidden-dim 512 --encoder-kernel-size 28 --encoder-padding 8 \
    --seed 0 --log-root ../gpurun_out/logs_eqh_$1 > ../gpurun_out/eqh_$1.log 2>&1
  echo "== $1"; grep -aoP "^\d+\t(train|test)\t\S+\t\S+\t\S+" ../gpurun_out/eqh_$1.log | tail -4
}
run h3
run x6
run f32
