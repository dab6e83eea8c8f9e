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

// h3 cells (tvae_dense_split2h): one maximum per padded row (and, behind them, scratch words of the GEMM entry points) behind
// the two part arrays
static inline float* h3_trailer(const void* a3, int rows, int K) {
    const long total = (long)dense_k8pad(K) * x6_round_up(rows, DX6_ROWS);
    return reinterpret_cast<float*>(const_cast<void*>(a3)) + 2 * total * 4;
}

// raw launchers of dense_x6_kernel<XV, NP>, one translation unit each (abi_dense_x6_v<XV>.hip: exact three-part split,
// abi_dense_x6_v<XV>b.hip: one-part bf16 throughput mode): the kernel is the slowest to compile in the library, so its
// instances build in parallel
#define TVAE_DX6_LAUNCH_ARGS                                                                                          \
    const uint4 *a3, const float *X, long ldx, const Epilogue &ep, int M, int Mpad, int N, int K, int K8pad,          \
        const TileMap &tm, const DenseBatch &bt, const ColDot &cd, const InTail &it, const VirtGrad &vg,              \
        const VirtAct &va, hipStream_t st, const H3Scale &hs
#define TVAE_DX6_DECL(XV_)                                              \
    TVAE_INTERNAL int dense_x6_launch_v##XV_##_p3(TVAE_DX6_LAUNCH_ARGS); \
    TVAE_INTERNAL int dense_x6_launch_v##XV_##_p2(TVAE_DX6_LAUNCH_ARGS); \
    TVAE_INTERNAL int dense_x6_launch_v##XV_##_p1(TVAE_DX6_LAUNCH_ARGS);
TVAE_DX6_DECL(0) TVAE_DX6_DECL(1) TVAE_DX6_DECL(2) TVAE_DX6_DECL(3) TVAE_DX6_DECL(4) TVAE_DX6_DECL(5)
#define TVAE_DX6_LAUNCH_DEF(XV_, NP_)                                                                                 \
    namespace tvae {                                                                                                  \
    int dense_x6_launch_v##XV_##_p##NP_(TVAE_DX6_LAUNCH_ARGS) {                                                       \
        hipLaunchKernelGGL((dense_x6_kernel<XV_, NP_>), dim3(tm.grid()), dim3(DX6_THREADS), 0, st, a3, X, ldx, ep, M, \
                           Mpad, N, K, K8pad, tm, bt, cd, it, vg, va, hs);                                            \
        return (int)hipGetLastError();                                                                                \
    }                                                                                                                 \
    }
// lean-epilogue instances (dense_x6_kernel<XV, NP, EPI>): abi_dense_x6_v2e1{,b,h}.hip and abi_dense_x6_v0e1{,b,h}.hip (forward
// without the stored activation: operand recomputed / read from memory) and abi_dense_x6_v5e2{,b,h}.hip (two-valued data gradient from bits with the fused first-layer backward)
#define TVAE_DX6_DECL_E(XV_, E_)                                              \
    TVAE_INTERNAL int dense_x6_launch_v##XV_##e##E_##_p3(TVAE_DX6_LAUNCH_ARGS); \
    TVAE_INTERNAL int dense_x6_launch_v##XV_##e##E_##_p2(TVAE_DX6_LAUNCH_ARGS); \
    TVAE_INTERNAL int dense_x6_launch_v##XV_##e##E_##_p1(TVAE_DX6_LAUNCH_ARGS);
TVAE_DX6_DECL_E(2, 1) TVAE_DX6_DECL_E(5, 2) TVAE_DX6_DECL_E(0, 1) TVAE_DX6_DECL_E(0, 3) TVAE_DX6_DECL_E(0, 4) TVAE_DX6_DECL_E(5, 3)
#define TVAE_DX6_LAUNCH_DEF_E(XV_, NP_, E_)                                                                           \
    namespace tvae {                                                                                                  \
    int dense_x6_launch_v##XV_##e##E_##_p##NP_(TVAE_DX6_LAUNCH_ARGS) {                                                \
        hipLaunchKernelGGL((dense_x6_kernel<XV_, NP_, E_>), dim3(tm.grid()), dim3(DX6_THREADS), 0, st, a3, X, ldx, ep, M, \
                           Mpad, N, K, K8pad, tm, bt, cd, it, vg, va, hs);                                            \
        return (int)hipGetLastError();                                                                                \
    }                                                                                                                 \
    }
#define TVAE_DX6_DISPATCH_E(XV_, E_, parts_, ...)                            \
    ((parts_) == 1 ? dense_x6_launch_v##XV_##e##E_##_p1(__VA_ARGS__)           \
                   : ((parts_) == 2 ? dense_x6_launch_v##XV_##e##E_##_p2(__VA_ARGS__) : dense_x6_launch_v##XV_##e##E_##_p3(__VA_ARGS__)))
// parts = 3 (exact split), 2 (h3: two fp16 parts) or 1 (bf16 throughput mode); anything else is rejected by the entry points
#define TVAE_DX6_DISPATCH(XV_, parts_, ...)                            \
    ((parts_) == 1 ? dense_x6_launch_v##XV_##_p1(__VA_ARGS__)           \
                   : ((parts_) == 2 ? dense_x6_launch_v##XV_##_p2(__VA_ARGS__) : dense_x6_launch_v##XV_##_p3(__VA_ARGS__)))

// weight-gradient launchers (abi_dense_wgrad_x6.hip: three parts, abi_dense_wgrad_x6_b.hip: one part)
#define TVAE_WG_LAUNCH_ARGS                                                                                           \
    int variant, const float *dY, long ldd, const float *X, long ldx, float *ws, int M, int Kf, int N, int nchunk,     \
        const TileMap &tm, const DenseBatch &bt, long dy_stride, const VirtGrad &vg, const VirtAct &va,               \
        const ATile &atile, hipStream_t st, const H3Scale &hs
TVAE_INTERNAL int dense_wgrad_x6_launch_p3(TVAE_WG_LAUNCH_ARGS);
TVAE_INTERNAL int dense_wgrad_x6_launch_p2(TVAE_WG_LAUNCH_ARGS);
TVAE_INTERNAL int dense_wgrad_x6_launch_p1(TVAE_WG_LAUNCH_ARGS);
// variant = VIRT | XVA << 1 | LRF << 2   (LRF: 0 off, 1 two-valued from H, 2 two-valued from sign bits)
#define TVAE_WG_ONE(V_, X_, L_, NP_)                                                                                  \
    do {                                                                                                              \
        hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<V_, X_, L_, NP_>, WG_RING_BYTES);                     \
        if (e_ != hipSuccess) return (int)e_;                                                                         \
        hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<V_, X_, L_, NP_>), dim3(tm.grid()), dim3(DX6_THREADS),           \
                           WG_RING_BYTES, st, dY, ldd, X, ldx, ws, M, Kf, N, nchunk, tm, bt, dy_stride, vg, va, atile, hs); \
        return (int)hipGetLastError();                                                                                \
    } while (0)
#define TVAE_WG_LAUNCH_DEF(NP_)                                                                                       \
    namespace tvae {                                                                                                  \
    int dense_wgrad_x6_launch_p##NP_(TVAE_WG_LAUNCH_ARGS) {                                                           \
        switch (variant) {                                                                                            \
            case 0: TVAE_WG_ONE(false, false, 0, NP_);                                                            \
            case 1: TVAE_WG_ONE(true, false, 0, NP_);                                                             \
            case 2: TVAE_WG_ONE(false, true, 0, NP_);                                                             \
            case 3: TVAE_WG_ONE(true, true, 0, NP_);                                                              \
            case 5: TVAE_WG_ONE(true, false, 1, NP_);                                                                 \
            case 7: TVAE_WG_ONE(true, true, 1, NP_);                                                                  \
            case 9: TVAE_WG_ONE(true, false, 2, NP_);                                                                 \
            case 11: TVAE_WG_ONE(true, true, 2, NP_);                                                                 \
            default: return (int)hipErrorInvalidValue;                                                                \
        }                                                                                                             \
    }                                                                                                                 \
    }

// one-part mode with the A operand STORED as bf16 (dense_wgrad_x6_dma_kernel<.., ABF>; abi_dense_wgrad_x6_b.hip)
TVAE_INTERNAL int dense_wgrad_x6_launch_p1_abf(TVAE_WG_LAUNCH_ARGS);
#define TVAE_WG_LAUNCH_DEF_ABF                                                                                        \
    namespace tvae {                                                                                                  \
    int dense_wgrad_x6_launch_p1_abf(TVAE_WG_LAUNCH_ARGS) {                                                           \
        if (variant != 0) return (int)hipErrorInvalidValue;                                                           \
        hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<false, false, 0, 1, true>, WG_RING_BYTES);            \
        if (e_ != hipSuccess) return (int)e_;                                                                         \
        hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<false, false, 0, 1, true>), dim3(tm.grid()), dim3(DX6_THREADS),  \
                           WG_RING_BYTES, st, dY, ldd, X, ldx, ws, M, Kf, N, nchunk, tm, bt, dy_stride, vg, va, atile, hs); \
        return (int)hipGetLastError();                                                                                \
    }                                                                                                                 \
    }

// the exact-fit 256 x 192 tile of the spectral weight gradient (dense_wgrad_x6_wide_kernel; tm / bt count 256-row tiles)
#define TVAE_WGW_LAUNCH_ARGS                                                                                          \
    const float *dY, long ldd, const float *X, long ldx, float *ws, int M, int Kf, int N, int nchunk, const TileMap &tm, \
        const DenseBatch &bt, long dy_stride, const ATile &atile, hipStream_t st, const H3Scale &hs
TVAE_INTERNAL int dense_wgrad_x6_wide_p3(TVAE_WGW_LAUNCH_ARGS);
TVAE_INTERNAL int dense_wgrad_x6_wide_p2(TVAE_WGW_LAUNCH_ARGS);
#define TVAE_WGW_LAUNCH_DEF(NP_)                                                                                      \
    namespace tvae {                                                                                                  \
    int dense_wgrad_x6_wide_p##NP_(TVAE_WGW_LAUNCH_ARGS) {                                                            \
        if (M % WW_ROWS != 0 || Kf <= 128 || bt.tiles_per_batch <= 0 ||                                             \
            tm.tilesN != (Kf <= 160 ? 1 : cdiv(Kf, 192)))                                                             \
            return (int)hipErrorInvalidValue;                                                                         \
        const unsigned grid_ =                                                                                        \
            8u * cdiv(tm.splits * (tm.tilesM / bt.tiles_per_batch), 8) * bt.tiles_per_batch * tm.tilesN;              \
        if (Kf <= 160) {             /* five column groups: the 66-wide frame of the 50 x 50 geometry (132 columns) */ \
            hipError_t e_ = allow_big_lds(dense_wgrad_x6_wide_kernel<NP_, 5>, WW_RING_BYTES);                         \
            if (e_ != hipSuccess) return (int)e_;                                                                     \
            hipLaunchKernelGGL((dense_wgrad_x6_wide_kernel<NP_, 5>), dim3(grid_), dim3(DX6_THREADS), WW_RING_BYTES, st, \
                               dY, ldd, X, ldx, ws, M, Kf, N, nchunk, tm, bt, dy_stride, atile, hs);                  \
            return (int)hipGetLastError();                                                                            \
        }                                                                                                             \
        hipError_t e_ = allow_big_lds(dense_wgrad_x6_wide_kernel<NP_, 6>, WW_RING_BYTES);                             \
        if (e_ != hipSuccess) return (int)e_;                                                                         \
        hipLaunchKernelGGL((dense_wgrad_x6_wide_kernel<NP_, 6>), dim3(grid_), dim3(DX6_THREADS), WW_RING_BYTES, st,   \
                           dY, ldd, X, ldx, ws, M, Kf, N, nchunk, tm, bt, dy_stride, atile, hs);                      \
        return (int)hipGetLastError();                                                                                \
    }                                                                                                                 \
    }

// the same with the 256-row / four-wave tile (dense_x6_plain4_kernel: short reductions); tm / bt count 256-row tiles
TVAE_INTERNAL int dense_x6_batched4(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem,
                                    int rows_total, int N, int K, const TileMap& tm, const DenseBatch& bt, int parts,
                                    hipStream_t st, H3Scale hs = H3_NONE, bool out_bf16 = false);
// the spectral contraction with the streamed panel resident in LDS (dense_x6_xres_kernel); false: shape not handled, nothing launched
TVAE_INTERNAL bool dense_x6_batched_xres(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem,
                                         int Mb, int nprob, int N, int K, long x_stride, long c_stride, int parts,
                                         hipStream_t st, H3Scale hs, int* rc);
// batched forward GEMM of the spectral contraction: rows of all problems stacked in w3 (abi_dense_x6.hip)
TVAE_INTERNAL int dense_x6_batched(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem,
                                   int rows_total, int N, int K, const TileMap& tm, const DenseBatch& bt, int parts,
                                   hipStream_t st, H3Scale hs = H3_NONE);
// batched weight-gradient GEMM into split-K slabs (abi_dense_wgrad_x6.hip)
TVAE_INTERNAL int dense_wgrad_x6_batched(const float* dY, long ldd, const float* X, long ldx, float* slabs, int M,
                                         int Kf, int N, int nchunk, const TileMap& tm, const DenseBatch& bt,
                                         long dy_stride, const ATile& atile, int parts, hipStream_t st,
                                         H3Scale hs = H3_NONE, bool a_bf16 = false, bool wide = false);

}  // namespace tvae
