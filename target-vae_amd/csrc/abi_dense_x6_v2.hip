// libtvae_hip.so: dense_x6_kernel<2, 3> -- recomputed first-layer activation operand (VirtAct); exact three-part split.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(2, 3)
