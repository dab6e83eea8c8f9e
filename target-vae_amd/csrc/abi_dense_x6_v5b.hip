// libtvae_hip.so: dense_x6_kernel<5, 1> -- two-valued data gradient with the 0 / 1 operand from stored sign bits; one-part bf16 throughput mode.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(5, 1)
