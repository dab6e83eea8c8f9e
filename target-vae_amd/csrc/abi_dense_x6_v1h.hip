// libtvae_hip.so: dense_x6_kernel<1, 2> -- implicit gradient operand; h3 arithmetic (two fp16 parts, three products).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(1, 2)
