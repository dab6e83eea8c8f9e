// libtvae_hip.so: dense_x6_kernel<1> -- implicit gradient operand (VirtGrad).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(1)
