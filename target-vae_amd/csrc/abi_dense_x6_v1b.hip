// libtvae_hip.so: dense_x6_kernel<1, 1> -- implicit gradient operand (VirtGrad); one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(1, 1)
