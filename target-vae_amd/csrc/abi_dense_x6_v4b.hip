// libtvae_hip.so: dense_x6_kernel<4, 1> -- two-valued implicit LeakyReLU gradient operand + row sums of H against gy (VirtGrad.rpart); one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(4, 1)
