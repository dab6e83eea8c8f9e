// libtvae_hip.so: dense_x6_kernel<0, 3> -- X read from memory; exact three-part split.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(0, 3)
