// libtvae_hip.so: dense_wgrad_x6_dma_kernel<.., 1> -- weight gradient in the one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_WG_LAUNCH_DEF(1)
TVAE_WG_LAUNCH_DEF_ABF
