// libtvae_hip.so: dense_x6_kernel<5, 3, 3> -- two-valued data gradient from sign bits with a STORED result: lean store epilogue.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF_E(5, 3, 3)
