#!/bin/bash
# Round 6: many-step equivalence of the arithmetics on the kernels that round 6 added -- the wide encoder tail (z = 50: 103 head
# rows), h3 for every hidden layer of a deep (4-layer) Fourier decoder with 3 outputs, the lean store epilogues.  The same CLI
# run (train_galaxy.py on 96 synthetic 64x64x3 images, P16, k = 32, p = 16, 128 encoder channels, hidden 512; 24 epochs x 6
# minibatches of 16 = 144 optimizer steps) with TVAE_GEMM=h3 (default), x6 (exact split; the tail then on the fp32-MFMA GEMMs)
# and f32.  Prints the last train / test lines:  gpurun -- 'bash profiles/train_equivalence_galaxy.sh'
set -e
cd "$(dirname "$0")/../target-vae_amd"
mkdir -p ../gpurun_out
run() {
  TVAE_GEMM=$1 python train_galaxy.py --synthetic 96 --image-dim 64 -z 50 --minibatch-size 16 --num-epochs 24 \
    --save-interval 100 --encoder-kernel-number 128 --generator-hidden-dim 512 --generator-num-layers 4 --encoder-kernel-size 32 \
    --encoder-padding 16 --groupconv 16 --fourier-expansion --seed 0 --log-root ../gpurun_out/logs_eqg_$1 > ../gpurun_out/eqg_$1.log 2>&1
  echo "== $1"; grep -aoP "^\d+\t(train|test)\t\S+\t\S+\t\S+" ../gpurun_out/eqg_$1.log | tail -4
}
run h3
run x6
run f32
