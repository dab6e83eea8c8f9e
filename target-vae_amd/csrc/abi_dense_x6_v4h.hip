// libtvae_hip.so: dense_x6_kernel<4, 2> -- two-valued implicit LeakyReLU gradient operand + row sums of H against gy (VirtGrad.rpart); h3 arithmetic (two fp16 parts, three products).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(4, 2)
