// libtvae_hip.so, direct lifting convolution on the bf16 matrix pipe with exactly split operands
// (conv_x6_kernels.hpp) -- the path of geometries the frequency-domain convolution does not cover.
#include "abi_common.hpp"
#include "conv_x6_kernels.hpp"

using namespace tvae;

// Geometry of the 3xbf16-split ("x6") lifting-convolution path (conv_x6_kernels.hpp).
struct X6Plan {
    int M, Mpad, opr, K8pad, Wp;
    int rows_f, arr_f, rows_w, arr_w, PT, arr_t;
    X6WgK kk;
    size_t lds_f, lds_w;
    long bank_cells, dy_cells;
};
static X6Plan x6_plan(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    X6Plan q;
    const int Hp = n + 2 * pad, Ho = Hp - ksz + 1;
    q.M = C * R;
    q.Mpad = x6_round_up(q.M, 256);
    q.opr = (ksz + 7) / 8;
    q.K8pad = x6_round_up(Cin * ksz * q.opr, 2);
    q.kk = x6_wg_k(Ho > 0 ? Ho : 1);
    q.Wp = x6_round_up(Hp, 2);
    q.rows_f = conv_fwd_img_rows(n, ksz, pad);
    q.arr_f = x6_arr_elems(Cin * q.rows_f * q.Wp);
    q.lds_f = (size_t)X6_FWD_RING_BYTES + X6_FWD_BIAS_BYTES + (size_t)6 * q.arr_f * 2;
    q.rows_w = conv_wgrad_img_rows(Cin, n, ksz, pad, 2);
    q.arr_w = x6_arr_elems(Cin * q.rows_w * q.Wp);
    // transposed copies of the leftover columns: pitch covers 8*opc rows of cells plus the tap-row spread of a tile
    q.PT = x6_round_up(8 * q.kk.opc + (q.rows_w - Ho) + 2, 2);
    q.arr_t = q.kk.rem > 0 ? x6_arr_elems(Cin * (ksz + 7) * q.PT) : 0;
    q.lds_w = (size_t)2 * X6_STAGE_CELLS_WG * 16 + X6_WG_TAB_INTS * 4 + (size_t)6 * (q.arr_w + q.arr_t) * 2;
    if (q.lds_w < 64 * 128 * 4) q.lds_w = 64 * 128 * 4;       // epilogue staging tile
    q.bank_cells = (long)3 * q.K8pad * q.Mpad;
    q.dy_cells = (long)3 * B * q.kk.QP * q.Mpad;
    return q;
}

extern "C" {

// ---- lifting convolution on the bf16 matrix pipe with fp32-equivalent results (3xbf16 split, 6 products) ----------
int tvae_conv1_x6_supported(int Cin, int n, int ksz, int pad) {
    const X6Plan q = x6_plan(1, Cin, n, ksz, pad, 1, 4);
    return (n + 2 * pad - ksz + 1 > 0 && q.lds_f <= X6_LDS_MAX && q.lds_w <= X6_LDS_MAX &&
            q.kk.cells + 2 <= X6_WG_TAB_INTS / 2) ? 1 : 0;
}
long tvae_conv1_x6_bank_bytes(int C, int R, int Cin, int ksz) {
    return x6_plan(1, Cin, ksz, ksz, 0, C, R).bank_cells * 16;
}
long tvae_conv1_x6_dy_bytes(int B, int C, int R, int n, int ksz, int pad) {
    return x6_plan(B, 1, n, ksz, pad, C, R).dy_cells * 16;
}

int tvae_bank_split3(const float* bank, void* a3, long a3_bytes, int C, int R, int Cin, int ksz,
                     tvae_stream_t stream) {
    const X6Plan q = x6_plan(1, Cin, ksz, ksz, 0, C, R);
    if (a3_bytes < q.bank_cells * 16 || !aligned16(a3)) return (int)hipErrorInvalidValue;
    const long total = (long)q.K8pad * q.Mpad;
    hipLaunchKernelGGL(bank_split3_kernel, dim3(grid1d(total, 256)), dim3(256), 0, S(stream), bank, (uint4*)a3, q.M,
                       q.Mpad, Cin, ksz, q.opr, q.K8pad);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_conv1_fwd_x6(const float* y, const void* a3, const float* bias, float* out, int B, int Cin, int n, int ksz,
                      int pad, int C, int R, int act, float slope, tvae_stream_t stream) {
    const ConvGeom g = make_geom(B, Cin, n, ksz, pad, R);
    if (g.Ho <= 0) return (int)hipErrorInvalidValue;
    const X6Plan q = x6_plan(B, Cin, n, ksz, pad, C, R);
    if (q.lds_f > X6_LDS_MAX || !aligned16(a3)) return (int)hipErrorInvalidValue;
    Epilogue ep;
    ep.C = out; ep.ldc = (long)B * R * g.P;
    int sh = 0; while ((1 << sh) < R) ++sh;
    if ((1 << sh) != R) return (int)hipErrorInvalidValue;
    ep.bias = bias; ep.bias_shift = sh;
    ep.act = act; ep.slope = slope;
    ep.convR = R; ep.conv_shift = sh; ep.convP = g.P;
    const int tilesPerImg = cdiv(g.P, BN);
    const long nblk = (long)(q.Mpad / 256) * B * tilesPerImg;
    if (nblk > 2147483647L) return (int)hipErrorInvalidValue;
    hipError_t e = allow_big_lds(conv1_fwd_x6_kernel, q.lds_f);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv1_fwd_x6_kernel, dim3((unsigned)nblk), dim3(GEMM_THREADS), q.lds_f, S(stream),
                       (const uint4*)a3, y, g, ep, q.M, q.Mpad, q.K8pad, q.opr, tilesPerImg, q.rows_f, q.Wp, q.arr_f);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_dy_split3(const float* dpre, void* d3, long d3_bytes, int B, int Cin, int n, int ksz, int pad, int C, int R,
                   tvae_stream_t stream) {
    const ConvGeom g = make_geom(B, Cin, n, ksz, pad, R);
    if (g.Ho <= 0) return (int)hipErrorInvalidValue;
    const X6Plan q = x6_plan(B, Cin, n, ksz, pad, C, R);
    if (d3_bytes < q.dy_cells * 16 || !aligned16(d3)) return (int)hipErrorInvalidValue;
    const size_t tile_bytes = (size_t)R * g.P * sizeof(float);
    if (tile_bytes > 150 * 1024) return (int)hipErrorInvalidValue;
    hipError_t e0 = allow_big_lds(dy_split3_kernel, tile_bytes);
    if (e0 != hipSuccess) return (int)e0;
    hipLaunchKernelGGL(dy_split3_kernel, dim3(B, q.Mpad / R), dim3(256), tile_bytes, S(stream), dpre, (long)B * R * g.P,
                       (uint4*)d3, B, C, R, g.Ho, q.kk.opwf, q.kk.opc, q.kk.row_cells, q.kk.cells, q.kk.QP, q.Mpad);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_conv1_wgrad_x6(const float* y, const void* d3, float* dbank, float* ws, long ws_floats, int B, int Cin, int n,
                        int ksz, int pad, int C, int R, tvae_stream_t stream) {
    const ConvGeom g = make_geom(B, Cin, n, ksz, pad, R);
    if (g.Ho <= 0) return (int)hipErrorInvalidValue;
    const X6Plan q = x6_plan(B, Cin, n, ksz, pad, C, R);
    const int M = q.M, N = Cin * g.K2;
    if (q.lds_w > X6_LDS_MAX || !aligned16(d3) || q.kk.cells + 2 > X6_WG_TAB_INTS / 2) return (int)hipErrorInvalidValue;
    const long per = (long)M * N;
    if (!ws || ws_floats < per) return (int)hipErrorInvalidValue;
    const int tilesM = cdiv(M, 128), tilesN = cdiv(N, 256);
    const int otiles = tilesM * tilesN;
    int sp = (4 * 256 * 2 + otiles / 2) / otiles;
    if (sp > B) sp = B;
    const long cap = ws_floats / per;
    if (sp > cap) sp = (int)cap;
    if (sp < 1) sp = 1;
    const int ips = cdiv(B, sp);
    sp = cdiv(B, ips);
    hipError_t e = allow_big_lds(conv1_wgrad_x6_kernel, q.lds_w);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv1_wgrad_x6_kernel, dim3((unsigned)(otiles * sp)), dim3(GEMM_THREADS), q.lds_w, S(stream),
                       (const uint4*)d3, y, g, M, q.Mpad, N, q.kk, ips, ws, tilesN, q.rows_w, q.Wp, q.arr_w, q.PT,
                       q.arr_t, sp, tilesM * sp);
    TVAE_CHECK_LAUNCH();
    Epilogue ep;
    ep.C = dbank; ep.ldc = N;
    int blocks = cdiv(per, 64);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, S(stream), (const float*)ws, sp, M, N, ep);
    TVAE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
