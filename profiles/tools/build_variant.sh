#!/bin/bash
# Builds a variant of libtvae_hip.so with ONE unit recompiled under extra flags (ablation / experiment macros), next to the
# shipped library:  bash profiles/tools/build_variant.sh abi_dense_wgrad_x6_h "-DTVAE_WW_ABL=1" ab_var/ww1.so
# Compare on one box:  gpurun -- 'bash profiles/tools/ab_kernels.sh TVAE_LIB "$PWD/ab_var/ww1.so ..." wgrad'
set -eu
UNIT=$1; XF=$2; OUT=$3
cd "$(dirname "$0")/../../target-vae_amd/csrc"
mkdir -p "$(dirname "../../$OUT")" build/var
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
TAG=$(echo "$UNIT$XF" | md5sum | cut -c1-8)
/opt/rocm/bin/hipcc $FLAGS $XF -c $UNIT.hip -o build/var/$UNIT.$TAG.o
OBJS=$(ls build/*.o | grep -v "build/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -fPIC $OBJS build/var/$UNIT.$TAG.o -o "../../$OUT"
echo "built $OUT"
