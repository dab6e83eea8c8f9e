// libtvae_hip.so: dense_x6_kernel<0, 1> -- X read from memory; one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(0, 1)
