// libtvae_hip.so: dense_x6_kernel<4, 3> -- two-valued implicit LeakyReLU gradient operand + row sums of H against gy (VirtGrad.rpart); exact three-part split.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(4, 3)
