// libtvae_hip.so: dense_x6_kernel<3, 1> -- two-valued implicit LeakyReLU gradient operand (VirtGrad.csum); one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(3, 1)
