// libtvae_hip.so: dense_x6_kernel<2, 2> -- recomputed first-layer activation operand; h3 arithmetic (two fp16 parts, three products).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(2, 2)
