// Host-side declarations shared by the split-pipe dense units (abi_dense_x6.hip, abi_dense_wgrad_x6.hip) and the
// frequency-domain convolution (abi_conv_dft.hip), which runs its spectral contraction as batched launches of them.
#pragma once
#include "abi_common.hpp"
#include "dense_x6_kernels.hpp"

namespace tvae {

// k-octets of the weight cells: whole 16-k steps
static inline int dense_k8pad(int K) { return x6_round_up((K + 7) / 8, 2); }
static inline long dense_x6_bytes(int rows, int K) {
    const long Rpad = x6_round_up(rows, DX6_ROWS), K8pad = dense_k8pad(K);
    return 3 * K8pad * Rpad * 16;
}

// raw launchers of dense_x6_kernel<XV>, one translation unit each (abi_dense_x6_v0/1/2/3.hip): the kernel is the slowest
// to compile in the library, so its instances build in parallel
#define TVAE_DX6_LAUNCH_ARGS                                                                                          \
    const uint4 *a3, const float *X, long ldx, const Epilogue &ep, int M, int Mpad, int N, int K, int K8pad,          \
        const TileMap &tm, const DenseBatch &bt, const ColDot &cd, const InTail &it, const VirtGrad &vg,              \
        const VirtAct &va, hipStream_t st
TVAE_INTERNAL int dense_x6_launch_v0(TVAE_DX6_LAUNCH_ARGS);
TVAE_INTERNAL int dense_x6_launch_v1(TVAE_DX6_LAUNCH_ARGS);
TVAE_INTERNAL int dense_x6_launch_v2(TVAE_DX6_LAUNCH_ARGS);
TVAE_INTERNAL int dense_x6_launch_v3(TVAE_DX6_LAUNCH_ARGS);
#define TVAE_DX6_LAUNCH_DEF(XV_)                                                                                      \
    namespace tvae {                                                                                                  \
    int dense_x6_launch_v##XV_(TVAE_DX6_LAUNCH_ARGS) {                                                                \
        hipLaunchKernelGGL(dense_x6_kernel<XV_>, dim3(tm.grid()), dim3(DX6_THREADS), 0, st, a3, X, ldx, ep, M, Mpad,  \
                           N, K, K8pad, tm, bt, cd, it, vg, va);                                                      \
        return (int)hipGetLastError();                                                                                \
    }                                                                                                                 \
    }

// batched forward GEMM of the spectral contraction: rows of all problems stacked in w3 (abi_dense_x6.hip)
TVAE_INTERNAL int dense_x6_batched(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem,
                                   int rows_total, int N, int K, const TileMap& tm, const DenseBatch& bt,
                                   hipStream_t st);
// batched weight-gradient GEMM into split-K slabs (abi_dense_wgrad_x6.hip)
TVAE_INTERNAL int dense_wgrad_x6_batched(const float* dY, long ldd, const float* X, long ldx, float* slabs, int M,
                                         int Kf, int N, int nchunk, const TileMap& tm, const DenseBatch& bt,
                                         long dy_stride, const ATile& atile, hipStream_t st);

}  // namespace tvae
