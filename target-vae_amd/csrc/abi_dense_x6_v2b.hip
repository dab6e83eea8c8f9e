// libtvae_hip.so: dense_x6_kernel<2, 1> -- recomputed first-layer activation operand (VirtAct); one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(2, 1)
