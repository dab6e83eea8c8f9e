// libtvae_hip.so: dense_x6_kernel<0, 1, 4> -- batched spectral contraction on the eight-wave tile: lean store epilogue.
#include "abi_dense_x6.hpp"
TVAE_DX6_LAUNCH_DEF_E(0, 1, 4)
