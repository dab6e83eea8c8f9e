// libtvae_hip.so, weight-gradient GEMM of the wide dense layers on the bf16 matrix pipe with exactly split
// operands, every streamed operand through an LDS-DMA ring (dense_x6_kernels.hpp: dense_wgrad_x6_dma_kernel).
#include "abi_dense_x6.hpp"

using namespace tvae;

TVAE_WG_LAUNCH_DEF(3)
TVAE_WGW_LAUNCH_DEF(3)

namespace tvae {
int dense_wgrad_x6_batched(const float* dY, long ldd, const float* X, long ldx, float* slabs, int M, int Kf, int N,
                           int nchunk, const TileMap& tm, const DenseBatch& bt, long dy_stride, const ATile& atile,
                           int parts, hipStream_t st, H3Scale hs, bool a_bf16, bool wide) {
    const VirtGrad vg{nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0};
    const VirtAct va{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f};
    if (a_bf16) {        // dY holds 2-byte bf16 elements (ldd, dy_stride, atile in elements): one-part mode only
        if (parts != 1 || dy_stride % 2 != 0) return (int)hipErrorInvalidValue;
        return dense_wgrad_x6_launch_p1_abf(0, dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, vg, va, atile, st, hs);
    }
    if (wide) {          // exact-fit tile (tm / bt in 256-row tiles, one column tile); two- and three-part arithmetic
        if (parts == 2 && hs.amax_a && hs.amax_x)
            return dense_wgrad_x6_wide_p2(dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, atile, st, hs);
        if (parts == 3) return dense_wgrad_x6_wide_p3(dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, atile, st, hs);
        return (int)hipErrorInvalidValue;
    }
    if (parts == 1) return dense_wgrad_x6_launch_p1(0, dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, vg, va, atile, st, hs);
    if (parts == 2 && hs.amax_a && hs.amax_x)
        return dense_wgrad_x6_launch_p2(0, dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, vg, va, atile, st, hs);
    if (parts == 3) return dense_wgrad_x6_launch_p3(0, dY, ldd, X, ldx, slabs, M, Kf, N, nchunk, tm, bt, dy_stride, vg, va, atile, st, hs);
    return (int)hipErrorInvalidValue;
}
}  // namespace tvae

namespace {
// Second half of the finalize of the two-valued weight gradient with RAW slabs (VirtGrad.raw: the row factor wo[m] left off;
// the generic split-K finalize has just left G = sum of the slices in dW): one workgroup per row m writes
// dW[m][k] = wo[m] G[m][k] in place and the row's dot product with the layer's own weight, rowdot[m] = sum_k W[m][k] G[m][k]
// -- from which the weight gradient of the single-output Linear BEHIND this layer follows without the layer's activation
// ever having been stored (VirtGrad, dense_x6_kernels.hpp).
__global__ void wgrad_lrf_rowdot_kernel(int K, const float* __restrict__ wo, const float* __restrict__ W, long ldw,
                                        float* __restrict__ dW, float* __restrict__ rowdot) {
    __shared__ float sm[16];
    const int m = blockIdx.x;
    float acc[1] = {0.f};
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float g = dW[(long)m * K + k];
        dW[(long)m * K + k] = wo[m] * g;
        acc[0] = __fmaf_rn(W[(long)m * ldw + k], g, acc[0]);
    }
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) rowdot[m] = acc[0];
}
}  // namespace

extern "C" {

// reduction slices of tvae_linear_wgrad_x6: a function of the SHAPE only -- never of the workspace the caller happens to
// pass (a cap by ws_floats made the first step of a process, whose shared workspace was still small, sum in a different
// order than every later step: 2.4e-7 relative on one tensor; round 3, profiles/tools/graph_replay_probe.py)
static int wgrad_x6_splits(int M, int N, int K) {
    const int tilesM = cdiv(M, DX6_ROWS), tilesK = cdiv(K, 128);
    // two rounds of one workgroup per CU, and NEVER one workgroup more (round 6: the Fourier first layer's K = 1 026 is 9 column
    // tiles; rounding UP gave 57 slices = 513 workgroups -- a third round of the 256 CUs for a single workgroup, 480 us where two
    // full rounds take 330)
    int splits = (2 * 256) / (tilesM * tilesK);
    if (splits > N / 16) splits = N / 16;
    if (splits < 2) splits = 2;                        // TileMap groups by reduction slice only when there are >= 2
    const int nchunk = cdiv(cdiv(N, splits), 16) * 16;
    return cdiv(N, nchunk);
}
constexpr int WG_H3_SLOTS = 8;         // behind the slabs: the operand bounds of the h3 arithmetic (4 words + one per X row)
long tvae_linear_wgrad_x6_ws_floats(int M, int N, int K) {
    if (M <= 0 || K <= 0 || N < 32) return 0;
    return (long)wgrad_x6_splits(M, N, K) * M * K + WG_H3_SLOTS + K;
}

int tvae_linear_wgrad_x6(const float* dpre, const float* X, float* dW, float* ws, long ws_floats, int M, int N, int K,
                         long ldd, long ldx, int accumulate, const float* vg_wo, const float* vg_gy, int vg_act,
                         float vg_slope, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                         int va_np, const void* vg_bits, int parts, const float* rd_w, long rd_ldw, float* rd_rowdot,
                         const float* a_amax, const float* x_amax, int x_amax_rows, tvae_stream_t stream) {
    // dW[m][k] = sum_n dpre[m][n] X[k][n]  (output M x K, reduction N), exact-split bf16 arithmetic
    if (M <= 0 || K <= 0) return 0;
    if (parts != 1 && parts != 2 && parts != 3) return (int)hipErrorInvalidValue;
    const bool from_bits = vg_bits && vg_wo && vg_act == ACT_LRELU;        // two-valued form from stored sign bits: dpre unused
    // h3 instances: (a) sign bits (exact 0 / 1 operand) against gy x the recomputed first-layer activation, whose bound is formed
    // here; (b) sign bits against gy x an operand from memory whose bound the caller supplies (x_amax: max |gy[n] X[k][n]| or
    // more); (c) two plain operands from memory with both bounds supplied (a_amax, x_amax)
    const bool h3_recomp = from_bits && va_xr && vg_gy;
    const bool h3_bits_mem = from_bits && !va_xr && vg_gy && x_amax;
    const bool h3_plain = !vg_wo && !va_xr && !vg_bits && a_amax && x_amax;
    if (parts == 2 && !(h3_recomp || h3_bits_mem || h3_plain)) return (int)hipErrorInvalidValue;
    if (vg_bits && !from_bits) return (int)hipErrorInvalidValue;
    if (N <= 0 || N % 16 != 0 || !ws || (from_bits ? N % 32 != 0 : (ldd % 4 != 0 || !dpre || !aligned16(dpre))))
        return (int)hipErrorInvalidValue;
    if (va_xr ? (va_np % 16 != 0 || !aligned16(va_xr))      // the DMA ring moves 16 columns of one image per step
            : (ldx % 4 != 0 || !X || !aligned16(X)))
        return (int)hipErrorInvalidValue;
    // rowdot (optional; two-valued form only, no accumulation): rd_rowdot[m] = sum_k rd_w[m][k] G[m][k], G = dW before wo[m]
    if (rd_rowdot && (!rd_w || !(vg_wo && vg_act == ACT_LRELU) || accumulate)) return (int)hipErrorInvalidValue;
    const VirtGrad vgs{vg_wo, vg_gy, vg_act, vg_slope, nullptr, (const unsigned*)vg_bits, nullptr, rd_rowdot ? 1 : 0};
    const VirtAct vas{va_xr, va_wc, va_bc, va_lb, va_np > 0 ? va_np : 1, vg_act, vg_slope};
    const int tilesM = cdiv(M, DX6_ROWS), tilesK = cdiv(K, 128);
    const long per = (long)M * K;
    if (N < 32 || ws_floats < tvae_linear_wgrad_x6_ws_floats(M, N, K)) return (int)hipErrorInvalidValue;
    const int splits = wgrad_x6_splits(M, N, K);
    const int nchunk = cdiv(cdiv(N, splits), 16) * 16;
    if (splits < 2) return (int)hipErrorInvalidValue;
    const TileMap tmk{tilesM, tilesK, splits};
    // the implicit LeakyReLU gradient runs in its two-valued form (0 / 1 streamed operand, see the kernel)
    const bool lrf = vg_wo && vg_act == ACT_LRELU;
    const int variant = (vg_wo ? 1 : 0) | (va_xr ? 2 : 0) | (lrf ? (from_bits ? 8 : 4) : 0);
    int rc;
    if (parts == 2 && !h3_recomp) {
        rc = dense_wgrad_x6_launch_p2(variant, dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk, DenseBatch{0, 0, 0}, 0L, vgs, vas,
                                      ATILE_PLAIN, S(stream),
                                      // x_amax_rows: one bound per ROW of X (= per column of dW) instead of one for the tensor
                                      H3Scale{a_amax, x_amax, 0, 0, 0, x_amax_rows ? 1 : 0, x_amax_rows ? K : 0});
    } else if (parts == 2) {
        float* slots = ws + (long)splits * per;
        hipLaunchKernelGGL(h3_zero_slots_kernel, dim3(1), dim3(256), 0, S(stream), slots, 4 + K);
        TVAE_CHECK_LAUNCH();
        const long nlb = va_lb ? (long)(N / vas.Np) * K : 0;
        hipLaunchKernelGGL(dec_l0_bound_kernel, dim3(grid1d(N / 2, 256, 512)), dim3(256), 0, S(stream), va_xr, 2L * N, va_wc,
                           va_bc, va_lb, nlb, K, slots);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(dense_absmax_kernel, dim3(grid1d((long)N, 256 * 8, 512)), dim3(256), 0, S(stream), vg_gy, (long)N, 1, N,
                           0, (const float*)nullptr, slots + 3);
        TVAE_CHECK_LAUNCH();
        rc = dense_wgrad_x6_launch_p2(variant, dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk, DenseBatch{0, 0, 0}, 0L, vgs, vas,
                                      ATILE_PLAIN, S(stream), H3Scale{nullptr, slots, 0, 0, 0, 0, 0});
    } else {
        rc = parts == 1
            ? dense_wgrad_x6_launch_p1(variant, dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk, DenseBatch{0, 0, 0}, 0L, vgs, vas, ATILE_PLAIN, S(stream), H3_NONE)
            : dense_wgrad_x6_launch_p3(variant, dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk, DenseBatch{0, 0, 0}, 0L, vgs, vas, ATILE_PLAIN, S(stream), H3_NONE);
    }
    if (rc) return rc;
    TVAE_CHECK_LAUNCH();
    Epilogue ep;
    ep.C = dW; ep.ldc = K;
    ep.accumulate = accumulate;
    int blocks = cdiv(per, 64);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, S(stream), (const float*)ws, splits, M, K, ep);
    TVAE_CHECK_LAUNCH();
    if (rd_rowdot) {                                    // raw slabs: dW holds G; scale the rows by wo, take the row dot products
        hipLaunchKernelGGL(wgrad_lrf_rowdot_kernel, dim3(M), dim3(128), 0, S(stream), K, vg_wo, rd_w, rd_ldw, dW, rd_rowdot);
        TVAE_CHECK_LAUNCH();
    }
    return 0;
}

}  // extern "C"
