// libtvae_hip.so, weight-gradient GEMM of the wide dense layers on the bf16 matrix pipe with exactly split
// operands, every streamed operand through an LDS-DMA ring (dense_x6_kernels.hpp: dense_wgrad_x6_dma_kernel).
#include "abi_dense_x6.hpp"

using namespace tvae;

namespace tvae {
int dense_wgrad_x6_batched(const float* dY, long ldd, const float* X, long ldx, float* slabs, int M, int Kf, int N,
                           int nchunk, const TileMap& tm, const DenseBatch& bt, long dy_stride, const ATile& atile,
                           hipStream_t st) {
    hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<false, false, false>, WG_RING_BYTES);
    if (e_ != hipSuccess) return (int)e_;
    hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<false, false, false>), dim3(tm.grid()), dim3(DX6_THREADS), WG_RING_BYTES,
                       st, dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, VirtGrad{nullptr, nullptr, 0, 0.f},
                       VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, atile);
    return (int)hipGetLastError();
}
}  // namespace tvae

extern "C" {

int tvae_linear_wgrad_x6(const float* dpre, const float* X, float* dW, float* ws, long ws_floats, int M, int N, int K,
                         long ldd, long ldx, int accumulate, const float* vg_wo, const float* vg_gy, int vg_act,
                         float vg_slope, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                         int va_np, tvae_stream_t stream) {
    // dW[m][k] = sum_n dpre[m][n] X[k][n]  (output M x K, reduction N), exact-split bf16 arithmetic
    if (M <= 0 || K <= 0) return 0;
    if (N <= 0 || N % 16 != 0 || ldd % 4 != 0 || !aligned16(dpre) || !ws) return (int)hipErrorInvalidValue;
    if (va_xr ? (va_np % 16 != 0 || !aligned16(va_xr))      // the DMA ring moves 16 columns of one image per step
            : (ldx % 4 != 0 || !X || !aligned16(X)))
        return (int)hipErrorInvalidValue;
    const VirtGrad vgs{vg_wo, vg_gy, vg_act, vg_slope};
    const VirtAct vas{va_xr, va_wc, va_bc, va_lb, va_np > 0 ? va_np : 1, vg_act, vg_slope};
    const int tilesM = cdiv(M, DX6_ROWS), tilesK = cdiv(K, 128);
    const long per = (long)M * K;
    const long cap = ws_floats / per;
    if (cap < 2 || N < 32) return (int)hipErrorInvalidValue;
    int splits = (2 * 256 + tilesM * tilesK - 1) / (tilesM * tilesK);     // ~2 workgroups per CU
    if (splits > cap) splits = (int)cap;
    if (splits > N / 16) splits = N / 16;
    if (splits < 2) splits = 2;                        // TileMap groups by reduction slice only when there are >= 2
    const int nchunk = cdiv(cdiv(N, splits), 16) * 16;
    splits = cdiv(N, nchunk);
    if (splits < 2) return (int)hipErrorInvalidValue;
    const TileMap tmk{tilesM, tilesK, splits};
#define TVAE_WG_LAUNCH(V_, X_, L_)                                                                                  \
    do {                                                                                                            \
        hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<V_, X_, L_>, WG_RING_BYTES);                        \
        if (e_ != hipSuccess) return (int)e_;                                                                       \
        hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<V_, X_, L_>), dim3(tmk.grid()), dim3(DX6_THREADS),             \
                           WG_RING_BYTES, S(stream), dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk,                   \
                           DenseBatch{0, 0, 0}, 0L, vgs, vas, ATILE_PLAIN);                                         \
    } while (0)
    // the implicit LeakyReLU gradient is factored (row factor x column factor x two-valued matrix, see the kernel)
    const bool lrf = vg_wo && vg_act == ACT_LRELU;
    if (vg_wo) {
        if (va_xr) { if (lrf) TVAE_WG_LAUNCH(true, true, true); else TVAE_WG_LAUNCH(true, true, false); }
        else { if (lrf) TVAE_WG_LAUNCH(true, false, true); else TVAE_WG_LAUNCH(true, false, false); }
    } else {
        if (va_xr) TVAE_WG_LAUNCH(false, true, false); else TVAE_WG_LAUNCH(false, false, false);
    }
#undef TVAE_WG_LAUNCH
    TVAE_CHECK_LAUNCH();
    Epilogue ep;
    ep.C = dW; ep.ldc = K;
    ep.accumulate = accumulate;
    int blocks = cdiv(per, 64);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, S(stream), (const float*)ws, splits, M, K, ep);
    TVAE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
