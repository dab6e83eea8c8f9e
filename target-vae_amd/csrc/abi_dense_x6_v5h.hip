// libtvae_hip.so: dense_x6_kernel<5, 2> -- two-valued data gradient with the 0 / 1 operand from stored sign bits; h3 arithmetic (two fp16 parts; two products against the 0 / 1 operand).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF(5, 2)
