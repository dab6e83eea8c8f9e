// libtvae_hip.so, lifting convolution with exact fp32 products (conv_img_kernels.hpp: image-resident
// implicit GEMM on v_mfma_f32_32x32x2_f32; generic implicit-im2col loaders of the GEMM core when the padded image does
// not fit LDS).
#include "abi_common.hpp"
#include "conv_img_kernels.hpp"

using namespace tvae;

// LDS budget for the image-resident conv kernels (160 KiB per CU on gfx950; keep room for 1 workgroup).
static const size_t CONV_IMG_LDS_MAX = 150 * 1024;

extern "C" {

int tvae_conv1_fwd(const float* y, const float* bank, const float* bias, float* out, int B, int Cin, int n, int ksz,
                   int pad, int C, int R, int act, float slope, tvae_stream_t stream) {
    const ConvGeom g = make_geom(B, Cin, n, ksz, pad, R);
    if (g.Ho <= 0) return (int)hipErrorInvalidValue;
    const int M = C * R, N = B * g.P, K = Cin * g.K2;
    Epilogue ep;
    ep.C = out; ep.ldc = (long)B * R * g.P;
    int sh = 0; while ((1 << sh) < R) ++sh;
    if ((1 << sh) != R) return (int)hipErrorInvalidValue;   // reference allows R in {4, 8, 16}
    ep.bias = bias; ep.bias_shift = sh;
    ep.act = act; ep.slope = slope;
    ep.convR = R; ep.conv_shift = sh; ep.convP = g.P;
    const int rows = conv_fwd_img_rows(n, ksz, pad);
    const size_t lds = conv_img_lds_bytes(Cin, rows, n, pad);
    if (lds <= CONV_IMG_LDS_MAX) {
        // image-resident path: the padded-image rows of the tile in LDS, B fragments read straight from them
        const int tilesPerImg = cdiv(g.P, BN);
        const long nblk = (long)cdiv(M, BM) * B * tilesPerImg;
        if (nblk > 2147483647L) return (int)hipErrorInvalidValue;
        const bool vec = (K % BK == 0) && (M % BM == 0);
        const size_t lds2 = conv_img_lds_bytes(Cin, rows, n, pad, 2);
        hipError_t e;
        if (vec && M % (2 * BM) == 0 && lds2 <= CONV_IMG_LDS_MAX) {
            // 256 x 128 tile: each wave 128 x 64 (8 MFMAs per operand wait)
            const long nblk2 = (long)(M / (2 * BM)) * B * tilesPerImg;
            e = allow_big_lds(conv1_fwd_img_kernel<true, 2>, lds2);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_fwd_img_kernel<true, 2>), dim3((unsigned)nblk2), dim3(GEMM_THREADS), lds2,
                               S(stream), bank, y, g, ep, M, K, tilesPerImg, rows);
        } else if (vec) {
            e = allow_big_lds(conv1_fwd_img_kernel<true, 1>, lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_fwd_img_kernel<true, 1>), dim3((unsigned)nblk), dim3(GEMM_THREADS), lds, S(stream),
                               bank, y, g, ep, M, K, tilesPerImg, rows);
        } else {
            e = allow_big_lds(conv1_fwd_img_kernel<false, 1>, lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_fwd_img_kernel<false, 1>), dim3((unsigned)nblk), dim3(GEMM_THREADS), lds, S(stream),
                               bank, y, g, ep, M, K, tilesPerImg, rows);
        }
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    // generic path (padded image does not fit in LDS): implicit im2col staged through LDS
    LoadKContig al{bank, (long)K, M};
    LoadConvPatchFwd bl{y, g, N};
    return (int)launch_gemm(al, bl, ep, M, N, K, 1, nullptr, 0, S(stream));
}

int tvae_conv1_wgrad(const float* y, const float* dpre, float* dbank, float* ws, long ws_floats, int B, int Cin,
                     int n, int ksz, int pad, int C, int R, tvae_stream_t stream) {
    const ConvGeom g = make_geom(B, Cin, n, ksz, pad, R);
    if (g.Ho <= 0) return (int)hipErrorInvalidValue;
    const int M = C * R, N = Cin * g.K2;
    const long Kl = (long)B * g.P;
    if (Kl > 2147483647L) return (int)hipErrorInvalidValue;
    const int K = (int)Kl;
    Epilogue ep;
    ep.C = dbank; ep.ldc = N;
    const int tilesM = cdiv(M, BM), tilesN = cdiv(N, BN);
    const int tiles = tilesM * tilesN;
    const int rows = conv_wgrad_img_rows(Cin, n, ksz, pad);
    const size_t lds = conv_img_lds_bytes(Cin, rows, n, pad);
    if (lds <= CONV_IMG_LDS_MAX) {
        // image-resident path.  The image reduction is split into ~4 waves of resident workgroups: zero skipping makes
        // edge-tap tiles up to 2x lighter, and many smaller slices let the dispatcher balance that (sweep at cfg4:
        // 3 / 6 / 12 / 32 slices -> 20.4 / 20.1 / 19.65 / 19.7 ms)
        const long per = (long)M * N;
        const long cap = ws ? ws_floats / per : 0;
        auto slices = [&](int out_tiles, int per_cu, int& ips) {
            int sp = (4 * 256 * per_cu + out_tiles / 2) / out_tiles;
            if (sp > B) sp = B;
            if (cap < 2) sp = 1; else if (sp > cap) sp = (int)cap;
            if (sp < 1) sp = 1;
            ips = cdiv(B, sp);
            return cdiv(B, ips);
        };
        int ips = B, sp = 1;
        // 128 x 256 tile (wave 64 x 128, 32 positions per k-step): every dY panel is re-read by half as many tap tiles
        const int rows2 = conv_wgrad_img_rows(Cin, n, ksz, pad, 2);
        const size_t ldsn = conv_img_lds_bytes(Cin, rows2, n, pad, 1, 32);
        const size_t lds32 = conv_img_lds_bytes(Cin, rows, n, pad, 1, 32);
        hipError_t e;
        if (N % (2 * BN) == 0 && ldsn * 2 <= 160 * 1024) {
            const int tilesN2 = N / (2 * BN);
            sp = slices(tilesM * tilesN2, 2, ips);
            e = allow_big_lds(conv1_wgrad_img_kernel<1, 32, 2>, ldsn);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_wgrad_img_kernel<1, 32, 2>), dim3((unsigned)(tilesM * tilesN2 * sp)),
                               dim3(GEMM_THREADS), ldsn, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips,
                               sp > 1 ? ws : nullptr, tilesN2, rows2, sp, tilesM * sp);
        } else if (lds32 * 3 <= 160 * 1024) {
            // 128 x 128 tile, 32 positions per k-step: half the barriers, still 3 workgroups per CU
            sp = slices(tiles, 3, ips);
            e = allow_big_lds(conv1_wgrad_img_kernel<1, 32, 1>, lds32);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_wgrad_img_kernel<1, 32, 1>), dim3((unsigned)(tiles * sp)), dim3(GEMM_THREADS),
                               lds32, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips, sp > 1 ? ws : nullptr,
                               tilesN, rows, sp, tilesM * sp);
        } else {
            sp = slices(tiles, 3, ips);
            e = allow_big_lds(conv1_wgrad_img_kernel<1, 16, 1>, lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_wgrad_img_kernel<1, 16, 1>), dim3((unsigned)(tiles * sp)), dim3(GEMM_THREADS),
                               lds, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips, sp > 1 ? ws : nullptr,
                               tilesN, rows, sp, tilesM * sp);
        }
        TVAE_CHECK_LAUNCH();
        if (sp > 1) {
            int blocks = cdiv(per, 64);
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, S(stream), (const float*)ws, sp, M, N,
                               ep);
            TVAE_CHECK_LAUNCH();
        }
        return 0;
    }
    LoadConvDY al{dpre, (long)B * R * g.P, M, R, g.P};
    LoadConvPatchWgrad bl{y, g, N};
    return (int)launch_gemm(al, bl, ep, M, N, K, pick_splits(tiles, K), ws, ws_floats, S(stream));
}

}  // extern "C"
