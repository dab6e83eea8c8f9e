// libtvae_hip.so: dense_wgrad_x6_dma_kernel<.., 2> -- weight gradient in the h3 arithmetic (two fp16 parts, three products).
#include "abi_dense_x6.hpp"
TVAE_WG_LAUNCH_DEF(2)
TVAE_WGW_LAUNCH_DEF(2)
