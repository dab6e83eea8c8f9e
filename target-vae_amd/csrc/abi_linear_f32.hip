// libtvae_hip.so, dense layers on the fp32 matrix pipe (gemm_f32_mfma.hpp): the 128-wide encoder 1x1x1
// layers in every arithmetic, and every dense layer when the caller asks for exact fp32 products.
#include "abi_common.hpp"

using namespace tvae;

// the float4 epilogue needs plain row-major C and 16-B aligned C / residual / aux rows
static inline int vec_epilogue_ok(const Epilogue& ep) {
    const bool plain = ep.convP == 0 && ep.gbias == nullptr && ep.accumulate == 0;
    const bool c_ok = aligned16(ep.C) && ep.ldc % 4 == 0;
    const bool r_ok = !ep.res || (aligned16(ep.res) && ep.ldres % 4 == 0);
    const bool a_ok = ep.mask == ACT_NONE || (aligned16(ep.aux) && ep.ldaux % 4 == 0);
    return (plain && c_ok && r_ok && a_ok) ? 1 : 0;
}

extern "C" {

int tvae_linear_fwd(const float* W, const float* X, const float* bias, const float* gbias, int group,
                    const float* res, float* Y, int M, int N, int K, long ldx, long ldy, int act, float slope,
                    tvae_stream_t stream) {
    LoadKContig al{W, (long)K, M};
    LoadXContig bl{X, ldx, N};
    Epilogue ep;
    ep.C = Y; ep.ldc = ldy;
    ep.bias = bias;
    ep.gbias = gbias; ep.ldg = M; ep.group = group > 0 ? group : 1;
    ep.res = res; ep.ldres = ldy;
    ep.act = act; ep.slope = slope;
    if (M % BM == 0 && N % BN == 0 && K % BK == 0 && ldx % 4 == 0 && aligned16(W) && aligned16(X)) {
        // aligned shapes: the activation tile is staged by LDS-DMA (global_load_lds), the weight through registers
        LoadKContigV4 af{W, (long)K, M};
        const TileMap tm{M / BM, N / BN, 1};
        hipLaunchKernelGGL((gemm_f32_glds_kernel<LoadKContigV4>), dim3(tm.grid()), dim3(GEMM_THREADS), 0, S(stream), af,
                           X, ldx, ep, M, N, K, tm, vec_epilogue_ok(ep));
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    return (int)launch_gemm(al, bl, ep, M, N, K, 1, nullptr, 0, S(stream));
}

int tvae_linear_dgrad(const float* W, const float* dpre, const float* add, const float* aux, float* dX, int M, int N,
                      int K, long ldd, long ldx, int mask, float slope, tvae_stream_t stream) {
    // dX[k][n] = sum_m W[m][k] dpre[m][n]: output rows = K, reduction = M
    LoadXContig al{W, (long)K, K};
    LoadXContig bl{dpre, ldd, N};
    Epilogue ep;
    ep.C = dX; ep.ldc = ldx;
    ep.res = add; ep.ldres = ldx;
    ep.aux = aux; ep.ldaux = ldx;
    ep.mask = aux ? mask : ACT_NONE; ep.slope = slope;
    if (K % BM == 0 && N % BN == 0 && M % BK == 0 && ldd % 4 == 0 && aligned16(W) && aligned16(dpre)) {
        // both operands are row-contiguous along their tile dimension: A(kout, m) = W[m][kout], B = dpre[m][n]
        const TileMap tm{K / BM, N / BN, 1};
        hipLaunchKernelGGL(gemm_f32_glds2_kernel, dim3(tm.grid()), dim3(GEMM_THREADS), 0, S(stream), W, (long)K, dpre,
                           ldd, ep, K, N, M, tm, vec_epilogue_ok(ep));
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    return (int)launch_gemm(al, bl, ep, K, N, M, 1, nullptr, 0, S(stream));
}

int tvae_linear_wgrad(const float* dpre, const float* X, float* dW, float* ws, long ws_floats, int M, int N, int K,
                      long ldd, long ldx, int accumulate, tvae_stream_t stream) {
    // dW[m][k] = sum_n dpre[m][n] X[k][n]: output M x K, reduction = N
    LoadKContig al{dpre, ldd, M};
    LoadKContig bl{X, ldx, K};
    Epilogue ep;
    ep.C = dW; ep.ldc = K;
    ep.accumulate = accumulate;
    const int tiles = cdiv(M, BM) * cdiv(K, BN);
    if (M % BM == 0 && K % BN == 0 && N % (BK * 1) == 0 && ldd % 4 == 0 && ldx % 4 == 0 && aligned16(dpre) &&
        aligned16(X)) {
        // split-K chunks are multiples of BK, and N % BK == 0, so every k-step of every slice is full
        LoadKContigV4 af{dpre, ldd, M};
        LoadKContigV4 bf{X, ldx, K};
        return (int)launch_gemm(af, bf, ep, M, K, N, pick_splits(tiles, N), ws, ws_floats, S(stream));
    }
    return (int)launch_gemm(al, bl, ep, M, K, N, pick_splits(tiles, N), ws, ws_floats, S(stream));
}

}  // extern "C"
