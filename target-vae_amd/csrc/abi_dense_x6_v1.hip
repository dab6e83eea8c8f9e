// libtvae_hip.so: dense_x6_kernel<1, 3> -- implicit gradient operand (VirtGrad); exact three-part split.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(1, 3)
