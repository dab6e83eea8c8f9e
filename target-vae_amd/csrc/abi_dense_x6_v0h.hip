// libtvae_hip.so: dense_x6_kernel<0, 2> -- X read from memory; h3 arithmetic (two fp16 parts, three products).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(0, 2)
