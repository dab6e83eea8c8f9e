// libtvae_hip.so: dense_x6_kernel<3, 3> -- two-valued implicit LeakyReLU gradient operand (VirtGrad.csum); exact three-part split.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(3, 3)
