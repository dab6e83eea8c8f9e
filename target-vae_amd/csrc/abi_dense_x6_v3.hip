// libtvae_hip.so: dense_x6_kernel<3> -- two-valued implicit LeakyReLU gradient operand (VirtGrad.csum), 3 MFMAs per block.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(3)
