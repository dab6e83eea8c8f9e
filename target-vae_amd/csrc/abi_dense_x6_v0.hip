// libtvae_hip.so: dense_x6_kernel<0> -- X read from memory.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(0)
