// libtvae_hip.so: dense_x6_kernel<0, 3, 1> -- forward with the operand read from memory, lean epilogue (activation not stored: column dot + sign bits).
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF_E(0, 3, 1)
