// libtvae_hip.so -- C ABI (include/tvae_hip.h) over the gfx950 kernels of the TARGET-VAE hot path.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/tvae_hip.h"
#include "gemm_f32_mfma.hpp"
#include "small_kernels.hpp"
#include "fused_tail_kernels.hpp"
#include "conv_x6_kernels.hpp"
#include "dense_x6_kernels.hpp"
#include "conv_dft_kernels.hpp"
#include "conv_img_kernels.hpp"
#include "gemm_bf16x3.hpp"

using namespace tvae;

#define TVAE_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

static inline hipStream_t S(tvae_stream_t s) { return (hipStream_t)s; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
// LDS-DMA (global_load_lds) staging of the aligned dense GEMMs; TVAE_GLDS=0 falls back to register staging
static inline bool use_glds() {
    static const bool on = [] { const char* e = getenv("TVAE_GLDS"); return !(e && e[0] == '0'); }();
    return on;
}
// the float4 epilogue needs plain row-major C and 16-B aligned C / residual / aux rows
static inline int vec_epilogue_ok(const Epilogue& ep) {
    const bool plain = ep.convP == 0 && ep.gbias == nullptr && ep.accumulate == 0;
    const bool c_ok = aligned16(ep.C) && ep.ldc % 4 == 0;
    const bool r_ok = !ep.res || (aligned16(ep.res) && ep.ldres % 4 == 0);
    const bool a_ok = ep.mask == ACT_NONE || (aligned16(ep.aux) && ep.ldaux % 4 == 0);
    return (plain && c_ok && r_ok && a_ok) ? 1 : 0;
}
static inline int grid1d(long total, int block, int cap = 8192) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
// number of split-K slices that brings a GEMM with `tiles` output tiles to ~4 workgroups per CU
static inline int pick_splits(int tiles, long K) {
    long want = (1024 + tiles - 1) / tiles;
    long maxs = K / (4 * BK);
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    return (int)want;
}

// LDS budget for the image-resident conv kernels (160 KiB per CU on gfx950; keep room for 1 workgroup).
static const size_t CONV_IMG_LDS_MAX = 150 * 1024;
static const size_t X6_LDS_MAX = 160 * 1024;       // whole LDS of a CU (one workgroup per CU by design)

template <class KernelT>
static hipError_t allow_big_lds(KernelT kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// GEMM arithmetic mode: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-bf16 x3 (fp32-level accuracy).
static int g_gemm_mode = 0;

static inline int panels_of(long N, int width) { return (int)((N + width - 1) / width); }
template <int NO>
static void launch_heads_fwd(const float* W, const float* X, long ldx, const float* bias, float* Y, long ldy, int C,
                             long N, int vec, hipStream_t st) {
    hipLaunchKernelGGL(heads_fwd_kernel<NO>, dim3(panels_of(N, 1024)), dim3(256), 0, st, W, X, ldx, bias, Y, ldy, C, N,
                       vec);
}
template <int NO>
static void launch_heads_bwd(const float* W, const float* dY, long ldy, const float* X, long ldx, float* dX, long lddx,
                             int C, long N, int act, float slope, float* part, int vec, hipStream_t st) {
    hipLaunchKernelGGL(heads_bwd_kernel<NO>, dim3(panels_of(N, PANEL8)), dim3(256), 0, st, W, dY, ldy, X, ldx, dX, lddx,
                       C, N, act, slope, part, vec);
}


static ConvGeom make_geom(int B, int Cin, int n, int ksz, int pad, int R) {
    ConvGeom g;
    g.B = B; g.Cin = Cin; g.n = n; g.ksz = ksz; g.pad = pad; g.R = R;
    g.Ho = n + 2 * pad - ksz + 1;
    g.P = g.Ho * g.Ho;
    g.K2 = ksz * ksz;
    return g;
}

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

static long tvae_dense_x6_bytes_impl(int rows, int K) {
    const long Rpad = x6_round_up(rows, DX6_ROWS), K8pad = x6_round_up((K + 7) / 8, 2);
    return 3 * K8pad * Rpad * 16;
}

// TVAE_DENSE_DMA=1: forward / data-gradient GEMM with LDS-DMA rings for its streamed operands (dense_x6_dma_kernel)
static bool dense_dma() {
    static const bool on = [] { const char* e_ = getenv("TVAE_DENSE_DMA"); return e_ && e_[0] == '1'; }();
    return on;
}
// TVAE_WGRAD_LRF=0: implicit LeakyReLU gradient formed and split per element instead of the factored two-valued form
static bool wgrad_lrf() {
    static const bool on = [] { const char* e_ = getenv("TVAE_WGRAD_LRF"); return !(e_ && e_[0] == '0'); }();
    return on;
}
// TVAE_WGRAD_DMA=0: weight-gradient GEMM with per-lane A loads instead of the LDS-DMA ring (dense_x6_kernels.hpp)
static bool wgrad_dma() {
    static const bool on = [] { const char* e_ = getenv("TVAE_WGRAD_DMA"); return !(e_ && e_[0] == '0'); }();
    return on;
}

// Geometry / workspace of the frequency-domain lifting convolution (conv_dft_kernels.hpp).
constexpr int DFT_WG_SPLITS = 8;
struct DftPlan {
    int L, Lh, Ho, M, K2;      // frame, half spectrum, output size, rows C*R, reduction 2L
    int LHP, NT, REM1;         // forward w-transform instance: frequencies processed, 32-row output tiles, extra row
    int NS, NRT;               // backward w-transform instance: k2-steps (pairs of w), 32-row tiles of (fx, ri)
    int Mb;                    // rows per fx in the stacked spectral weight (2M rounded up to the 512-row tile)
    long NB, NBpad;            // (image, output row) columns
    long at_floats;            // A^T [Lh][2L][NBpad]
    long w_floats;             // W   [Lh][2M][2L]
    long w3_floats;            // split cells of W
    long t_floats;             // T / S' [Lh][2M][NBpad]
    long tab_floats;
    long g_floats;             // G [Lh][2M][2L] (finalised spectral weight gradient)
    bool ok;
};
static DftPlan dft_plan(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    DftPlan q;
    q.L = n + 2 * pad;
    q.Lh = q.L / 2 + 1;
    q.Ho = q.L - ksz + 1;
    q.M = C * R;
    q.K2 = 2 * q.L;
    q.NB = (long)B * q.Ho;
    q.NBpad = (q.NB + 127) / 128 * 128;
    q.Mb = x6_round_up(2 * q.M, DX6_ROWS);
    q.at_floats = (long)q.Lh * q.K2 * q.NBpad;
    q.w_floats = (long)q.Lh * q.Mb * q.K2;
    q.w3_floats = tvae_dense_x6_bytes_impl(q.Lh * q.Mb, q.K2) / 4;
    q.t_floats = (long)q.Lh * 2 * q.M * q.NBpad;
    q.LHP = q.Lh == 23 ? 23 : (q.Lh == 49 ? 49 : 64);          // exact instances of the two reference frames, else generic
    q.NT = q.Ho <= 33 ? 1 : 2;
    q.REM1 = q.Ho == 33 ? 1 : 0;
    if (q.Ho <= 18 && q.Lh <= 32) { q.NS = 9; q.NRT = 2; }
    else if (q.Ho <= 34) { q.NS = 17; q.NRT = 4; }
    else { q.NS = 32; q.NRT = 4; }
    q.tab_floats = 64L * 2 * 64 + 32L * 4 * 64 + 4L * q.Lh * DFT_WMAX;       // EO + ED (largest instances) + ALU tables
    q.g_floats = (long)q.Lh * 2 * q.M * q.K2;
    const size_t lds_img = (size_t)n * n * 4 + (size_t)n * q.Lh * 8 + (size_t)q.L * q.Lh * 8 + (size_t)q.L * 8;
    const size_t lds_bank = (size_t)ksz * ksz * 4 + (size_t)ksz * q.Lh * 8 + (size_t)q.L * q.Lh * 8 + (size_t)q.L * 8;
    q.ok = Cin == 1 && q.Ho >= 1 && q.Ho <= DFT_WROWS && q.Lh <= 64 && lds_img <= 150 * 1024 &&
           lds_bank <= 150 * 1024 && (long)q.Lh * 2 * q.M < 2000000000L / 1;
    return q;
}

extern "C" {

int tvae_abi_version(void) { return 1; }
int tvae_set_gemm_mode(int mode) {
    // 0 = exact fp32 MFMA everywhere, 1 = split-bf16 x3 everywhere (lower accuracy), 2 = "x6": the caller routes the
    // lifting convolution through tvae_conv1_*_x6 (3 x bf16 exact split, fp32-equivalent); the dense entry points here
    // keep the exact fp32 MFMA
    if (mode != 0 && mode != 1 && mode != 2) return (int)hipErrorInvalidValue;
    g_gemm_mode = mode;
    return 0;
}
int tvae_get_gemm_mode(void) { return g_gemm_mode; }

// 1 if tvae_conv1_fwd wants the k-major bank bankT[Cin*k*k][C*R] (barrier-free kernel), 0 for bank[C*R][Cin*k*k]
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
    static const int splits_env = [] { const char* e = getenv("TVAE_CONV1_WGRAD_SPLITS"); return e ? atoi(e) : 0; }();
    if (splits_env > 0) sp = splits_env;
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

// ---- dense layers on the bf16 matrix pipe with exactly split operands (dense_x6_kernels.hpp) ---------------------
long tvae_dense_x6_bytes(int rows, int K) { return tvae_dense_x6_bytes_impl(rows, K); }
int tvae_dense_split3(const float* W, long ldw, void* a3, long a3_bytes, int rows, int K, int transpose,
                      tvae_stream_t stream) {
    if (rows <= 0 || K <= 0) return 0;
    if (a3_bytes < tvae_dense_x6_bytes(rows, K) || !aligned16(a3)) return (int)hipErrorInvalidValue;
    const int Rpad = x6_round_up(rows, DX6_ROWS), K8pad = x6_round_up((K + 7) / 8, 2);
    const long total = (long)K8pad * Rpad;
    hipLaunchKernelGGL(dense_split3_kernel, dim3(grid1d(total, 256)), dim3(256), 0, S(stream), W, ldw, (uint4*)a3, rows,
                       Rpad, K, K8pad, transpose);
    TVAE_CHECK_LAUNCH();
    return 0;
}
static int launch_dense_x6(const void* a3, const float* X, long ldx, const Epilogue& ep, int rows, int N, int K,
                           hipStream_t st, ColDot cd = ColDot{nullptr, nullptr, nullptr},
                           InTail it = InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1},
                           VirtGrad vg = VirtGrad{nullptr, nullptr, 0, 0.f},
                           VirtAct va = VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}) {
    if ((cd.w || it.xr) && rows > DX6_ROWS) return (int)hipErrorInvalidValue;   // the fused tails need ONE row tile
    if (rows <= 0 || N <= 0) return 0;
    if (N % 128 != 0 || !aligned16(a3)) return (int)hipErrorInvalidValue;
    const int Rpad = x6_round_up(rows, DX6_ROWS), K8pad = x6_round_up((K + 7) / 8, 2);
    const TileMap tm{Rpad / DX6_ROWS, N / 128, 1};
    // the recomputed operands need tiles inside one image and tables of <= 512 entries
    if ((va.xr && (K > 512 || va.Np % 128 != 0 || vg.wo)) || (it.bc && it.Np % 128 != 0) || (!va.xr && !X))
        return (int)hipErrorInvalidValue;
    if (dense_dma() && (!vg.wo || K <= 512)) {
#define TVAE_DX_DMA(V_)                                                                                               \
    do {                                                                                                              \
        const size_t rb_ = (V_) == 2 ? (size_t)8 * 2 * DX_A_SLOT : (size_t)DX_RING_BYTES;   /* no X ring when recomputed */ \
        hipError_t e_ = allow_big_lds(dense_x6_dma_kernel<V_>, rb_);                                                  \
        if (e_ != hipSuccess) return (int)e_;                                                                         \
        hipLaunchKernelGGL(dense_x6_dma_kernel<V_>, dim3(tm.grid()), dim3(DX6_THREADS), rb_, st,                       \
                           (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, DenseBatch{0, 0, 0}, cd, it, vg, va); \
    } while (0)
        if (va.xr) TVAE_DX_DMA(2); else if (vg.wo) TVAE_DX_DMA(1); else TVAE_DX_DMA(0);
#undef TVAE_DX_DMA
    } else if (va.xr)
        hipLaunchKernelGGL(dense_x6_kernel<2>, dim3(tm.grid()), dim3(DX6_THREADS), 0, st, (const uint4*)a3, X, ldx, ep,
                           rows, Rpad, N, K, K8pad, tm, DenseBatch{0, 0, 0}, cd, it, vg, va);
    else if (vg.wo)
        hipLaunchKernelGGL(dense_x6_kernel<1>, dim3(tm.grid()), dim3(DX6_THREADS), 0, st, (const uint4*)a3, X, ldx, ep,
                           rows, Rpad, N, K, K8pad, tm, DenseBatch{0, 0, 0}, cd, it, vg, va);
    else
        hipLaunchKernelGGL(dense_x6_kernel<0>, dim3(tm.grid()), dim3(DX6_THREADS), 0, st, (const uint4*)a3, X, ldx, ep,
                           rows, Rpad, N, K, K8pad, tm, DenseBatch{0, 0, 0}, cd, it, vg, va);
    hipError_t e = hipGetLastError();
    return (int)e;
}
int tvae_linear_fwd_x6(const void* w3, const float* X, const float* bias, const float* res, float* Y, int M, int N,
                       int K, long ldx, long ldy, int act, float slope, const float* col_w, const float* col_b,
                       float* col_y, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                       int va_np, tvae_stream_t stream) {
    Epilogue ep;
    ep.C = Y; ep.ldc = ldy;
    ep.bias = bias;
    ep.res = res; ep.ldres = ldy;
    ep.act = act; ep.slope = slope;
    return launch_dense_x6(w3, X, ldx, ep, M, N, K, S(stream), ColDot{col_w, col_b, col_y},
                           InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1}, VirtGrad{nullptr, nullptr, 0, 0.f},
                           VirtAct{va_xr, va_wc, va_bc, va_lb, va_np > 0 ? va_np : 1, act, slope});
}
int tvae_linear_dgrad_x6(const void* w3t, const float* dpre, const float* add, const float* aux, float* dX, int M,
                         int N, int K, long ldd, long ldx, int mask, float slope, const float* in_xr,
                         const float* in_wc, float* in_gxr, float* in_part, long in_part_floats, const float* vg_wo,
                         const float* vg_gy, const float* in_bc, const float* in_lb, int in_np, tvae_stream_t stream) {
    // dX[k][n] = act'(aux[k][n]) * (add[k][n] + sum_m W[m][k] dpre[m][n]): rows = K, reduction = M; w3t = split of W^T
    Epilogue ep;
    ep.C = dX; ep.ldc = ldx;                             // dX may be NULL when the fused first-layer backward consumes it
    ep.res = add; ep.ldres = ldx;
    ep.aux = aux; ep.ldaux = ldx;
    ep.mask = (aux || in_bc) ? mask : ACT_NONE; ep.slope = slope;
    if (in_bc && !in_xr) return (int)hipErrorInvalidValue;               // the recomputed mask needs the coordinates
    if (in_xr) {
        if (!in_wc || !in_gxr || !in_part || in_part_floats < (long)(N / 128) * K * 3) return (int)hipErrorInvalidValue;
    } else if (!dX) {
        return (int)hipErrorInvalidValue;
    }
    return launch_dense_x6(w3t, dpre, ldd, ep, K, N, M, S(stream), ColDot{nullptr, nullptr, nullptr},
                           InTail{in_xr, in_wc, in_gxr, in_part, in_bc, in_lb, in_np > 0 ? in_np : 1},
                           VirtGrad{vg_wo, vg_gy, mask, slope});
}

int tvae_dec_in_total(const float* part, int B, int cpi, int F, float* Simg, float* dbc, float* dWc,
                      tvae_stream_t stream) {
    // second stage of the fused first-layer backward: part[B*cpi panels][F][3] -> per-image sums, bias and weight grads
    if (B <= 0 || F <= 0) return 0;
    hipLaunchKernelGGL(dec_in_total_kernel, dim3(F), dim3(256), 0, S(stream), part, B, cpi, F, Simg, dbc, dWc);
    TVAE_CHECK_LAUNCH();
    return 0;
}
int tvae_linear_wgrad_x6(const float* dpre, const float* X, float* dW, float* ws, long ws_floats, int M, int N, int K,
                         long ldd, long ldx, int accumulate, const float* vg_wo, const float* vg_gy, int vg_act,
                         float vg_slope, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                         int va_np, tvae_stream_t stream) {
    // dW[m][k] = sum_n dpre[m][n] X[k][n]  (output M x K, reduction N), exact-split bf16 arithmetic
    if (M <= 0 || K <= 0) return 0;
    if (N <= 0 || N % 16 != 0 || ldd % 4 != 0 || !aligned16(dpre) || !ws) return (int)hipErrorInvalidValue;
    if (va_xr ? (va_np % 4 != 0 || !aligned16(va_xr)) : (ldx % 4 != 0 || !X || !aligned16(X)))
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
#define TVAE_WG_LAUNCH(V_, X_)                                                                                        \
    do {                                                                                                              \
        if (wgrad_dma() && (!va_xr || va_np % 16 == 0)) {                                                             \
            if ((V_) && vg_act == ACT_LRELU && wgrad_lrf()) {                                                         \
                hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<V_, X_, V_>, WG_RING_BYTES);                  \
                if (e_ != hipSuccess) return (int)e_;                                                                 \
                hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<V_, X_, V_>), dim3(tmk.grid()), dim3(DX6_THREADS),       \
                                   WG_RING_BYTES, S(stream), dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk,             \
                                   DenseBatch{0, 0, 0}, 0L, vgs, vas, ATILE_PLAIN);                                   \
            } else {                                                                                                  \
                hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<V_, X_, false>, WG_RING_BYTES);               \
                if (e_ != hipSuccess) return (int)e_;                                                                 \
                hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<V_, X_, false>), dim3(tmk.grid()), dim3(DX6_THREADS),    \
                                   WG_RING_BYTES, S(stream), dpre, ldd, X, ldx, ws, M, K, N, nchunk, tmk,             \
                                   DenseBatch{0, 0, 0}, 0L, vgs, vas, ATILE_PLAIN);                                   \
            }                                                                                                         \
        } else {                                                                                                      \
            hipLaunchKernelGGL((dense_wgrad_x6_kernel<V_, X_>), dim3(tmk.grid()), dim3(DX6_THREADS), 0, S(stream), dpre, \
                               ldd, X, ldx, ws, M, K, N, nchunk, tmk, DenseBatch{0, 0, 0}, 0L, vgs, vas, ATILE_PLAIN); \
        }                                                                                                             \
    } while (0)
    if (vg_wo) { if (va_xr) TVAE_WG_LAUNCH(true, true); else TVAE_WG_LAUNCH(true, false); }
    else { if (va_xr) TVAE_WG_LAUNCH(false, true); else TVAE_WG_LAUNCH(false, false); }
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

// ---- lifting convolution through the frequency domain (conv_dft_kernels.hpp) -------------------------------------
int tvae_conv1_dft_supported(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    return dft_plan(B, Cin, n, ksz, pad, C, R).ok ? 1 : 0;
}
long tvae_conv1_dft_at_floats(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    return dft_plan(B, Cin, n, ksz, pad, C, R).at_floats;
}
long tvae_conv1_dft_ws_floats(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    // forward: W + W3 + T + tables; backward: S' (= T) + slabs (2 x G) + G + tables
    const long fwd = q.w_floats + q.w3_floats + q.t_floats + q.tab_floats + 64;
    const long bwd = q.t_floats + (DFT_WG_SPLITS + 1) * q.g_floats + q.tab_floats + 64;
    return fwd > bwd ? fwd : bwd;
}

int tvae_conv1_fwd_dft(const float* y, const float* bank, const float* bias, float* out, float* at, float* ws,
                       long ws_floats, int B, int Cin, int n, int ksz, int pad, int C, int R, int act, float slope,
                       tvae_stream_t stream) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    if (!q.ok || ws_floats < tvae_conv1_dft_ws_floats(B, Cin, n, ksz, pad, C, R) || !aligned16(ws) || !aligned16(at))
        return (int)hipErrorInvalidValue;
    hipStream_t st = S(stream);
    float* W = ws;
    float* W3 = W + ((q.w_floats + 3) & ~3L);
    float* T = W3 + ((q.w3_floats + 3) & ~3L);
    float* tab = T + ((q.t_floats + 3) & ~3L);
    if (q.NBpad != q.NB) {
        hipError_t e = hipMemsetAsync(at, 0, (size_t)q.at_floats * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    float* EO = tab;
    float* ED = EO + 64L * 2 * 64;
    float* vtab = ED + 32L * 4 * 64;
    const size_t lds_img = (size_t)n * n * 4 + (size_t)n * q.Lh * 8 + (size_t)q.L * q.Lh * 8 + (size_t)q.L * 8;
    hipError_t e = allow_big_lds(dft_image_kernel, lds_img);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dft_image_kernel, dim3(B), dim3(256), lds_img, st, y, at, n, pad, q.L, q.Lh, q.Ho, q.NBpad);
    TVAE_CHECK_LAUNCH();
    const size_t lds_bank = (size_t)ksz * ksz * 4 + (size_t)ksz * q.Lh * 8 + (size_t)q.L * q.Lh * 8 + (size_t)q.L * 8;
    e = allow_big_lds(dft_bank_kernel, lds_bank);
    if (e != hipSuccess) return (int)e;
    if (q.Mb != 2 * q.M) {                             // rows that pad 2M to the 512-row tile must be zero
        e = hipMemsetAsync(W, 0, (size_t)q.w_floats * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(dft_bank_kernel, dim3(q.M), dim3(256), lds_bank, st, bank, W, ksz, q.L, q.Lh, q.M, q.Mb);
    TVAE_CHECK_LAUNCH();
    // split the stacked spectral weights [Lh*2M rows][2L] into cells, then ONE batched launch of the split dense GEMM
    const int rows = q.Lh * q.Mb;
    int rc = tvae_dense_split3(W, q.K2, W3, q.w3_floats * 4, rows, q.K2, 0, stream);
    if (rc) return rc;
    {
        Epilogue ep;
        ep.C = T; ep.ldc = (long)q.Lh * 128;              // T is [n >> 7][m'][fx][n & 127] (dft_t_off)
        ep.ctile = (long)2 * q.M * q.Lh * 128;
        const int Rpad = x6_round_up(rows, DX6_ROWS), K8pad = x6_round_up((q.K2 + 7) / 8, 2);
        TileMap tm{Rpad / DX6_ROWS, (int)(q.NBpad / 128), 1};
        tm.bt = q.Mb / DX6_ROWS;                       // group = (fx, quarter of the column tiles): 4*Lh groups over 8 XCDs
        tm.nch = 4;
        const DenseBatch bt{q.Mb / DX6_ROWS, (long)q.K2 * q.NBpad, 128};
        if (dense_dma()) {
            hipError_t e_ = allow_big_lds(dense_x6_dma_kernel<0>, DX_RING_BYTES);
            if (e_ != hipSuccess) return (int)e_;
            hipLaunchKernelGGL(dense_x6_dma_kernel<0>, dim3(tm.grid()), dim3(DX6_THREADS), DX_RING_BYTES, st, (const uint4*)W3,
                               (const float*)at, q.NBpad, ep, 2 * q.M, Rpad, (int)q.NBpad, q.K2, K8pad, tm, bt,
                               ColDot{nullptr, nullptr, nullptr}, InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1},
                               VirtGrad{nullptr, nullptr, 0, 0.f}, VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f});
        } else
        hipLaunchKernelGGL(dense_x6_kernel<0>, dim3(tm.grid()), dim3(DX6_THREADS), 0, st, (const uint4*)W3, (const float*)at,
                           q.NBpad, ep, 2 * q.M, Rpad, (int)q.NBpad, q.K2, K8pad, tm, bt, ColDot{nullptr, nullptr, nullptr},
                           InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1}, VirtGrad{nullptr, nullptr, 0, 0.f},
                           VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f});
        TVAE_CHECK_LAUNCH();
    }
    // TVAE_DFT_W_VALU = 1 | out | dy selects the vector-ALU transform(s) along w instead of the matrix-pipe ones
    static const bool w_valu = [] { const char* e_ = getenv("TVAE_DFT_W_VALU"); return e_ && (e_[0] == '1' || e_[0] == 'o'); }();
    if (!(w_valu && q.Ho <= DFT_WMAX)) {
        // contraction over fx on the fp32 matrix pipe (dft_out_mf_kernel)
        const int NTT = q.NT + q.REM1;
        hipLaunchKernelGGL(dft_wtab_kernel, dim3(32), dim3(256), 0, st, EO, ED, q.L, q.Lh, q.Ho, q.LHP, q.NT, NTT, q.NS,
                           q.NRT);
        TVAE_CHECK_LAUNCH();
        const size_t lds_o = ((size_t)q.LHP * NTT * 64 + (size_t)4 * 32 * (q.Ho | 1)) * 4;
        const long ntiles = (long)q.M * (q.NBpad / 32);
        const int grid = (int)((ntiles + 3) / 4 < 768 ? (ntiles + 3) / 4 : 768);
        const int iters = (int)((ntiles + 4L * grid - 1) / (4L * grid));
#define TVAE_OUT_MF(L_, N_, R_)                                                                                     \
    do {                                                                                                            \
        e = allow_big_lds(dft_out_mf_kernel<L_, N_, R_>, lds_o);                                                    \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((dft_out_mf_kernel<L_, N_, R_>), dim3(grid), dim3(256), lds_o, st, (const float*)T,      \
                           (const float*)EO, bias, out, q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope, iters);        \
    } while (0)
#define TVAE_OUT_MF_L(L_)                                                                                           \
    do {                                                                                                            \
        if (q.REM1) TVAE_OUT_MF(L_, 1, true); else if (q.NT == 1) TVAE_OUT_MF(L_, 1, false); else TVAE_OUT_MF(L_, 2, false); \
    } while (0)
        if (q.LHP == 23) TVAE_OUT_MF_L(23); else if (q.LHP == 49) TVAE_OUT_MF_L(49); else TVAE_OUT_MF_L(64);
#undef TVAE_OUT_MF_L
#undef TVAE_OUT_MF
        TVAE_CHECK_LAUNCH();
    } else {
        // vector-ALU contraction over fx (opt-in: TVAE_DFT_W_VALU=1)
        hipLaunchKernelGGL(dft_tables_kernel, dim3(8), dim3(256), 0, st, vtab, q.L, q.Lh);
        TVAE_CHECK_LAUNCH();
        const dim3 og((unsigned)((q.NB + 255) / 256), q.M);
        if (q.Ho == 33)
            hipLaunchKernelGGL(dft_out_kernel<33>, og, dim3(256), 0, st, (const float*)T, (const float*)vtab, bias, out,
                               q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope);
        else if (q.Ho == 17)
            hipLaunchKernelGGL(dft_out_kernel<17>, og, dim3(256), 0, st, (const float*)T, (const float*)vtab, bias, out,
                               q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope);
        else
            hipLaunchKernelGGL(dft_out_kernel<DFT_WMAX>, og, dim3(256), 0, st, (const float*)T, (const float*)vtab, bias,
                               out, q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope);
        TVAE_CHECK_LAUNCH();
    }
    return 0;
}

int tvae_conv1_wgrad_dft(const float* dpre, const float* at, float* dbank, float* dbias, float* ws, long ws_floats, int B,
                         int Cin, int n, int ksz, int pad, int C, int R, tvae_stream_t stream) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    if (!q.ok || ws_floats < tvae_conv1_dft_ws_floats(B, Cin, n, ksz, pad, C, R) || !aligned16(ws) || !aligned16(at))
        return (int)hipErrorInvalidValue;
    hipStream_t st = S(stream);
    float* Sp = ws;
    float* slabs = Sp + ((q.t_floats + 3) & ~3L);
    float* G = slabs + ((DFT_WG_SPLITS * q.g_floats + 3) & ~3L);
    float* tab = G + ((q.g_floats + 3) & ~3L);
    float* EO = tab;
    float* ED = EO + 64L * 2 * 64;
    float* vtab = ED + 32L * 4 * 64;
    static const bool w_valu = [] { const char* e_ = getenv("TVAE_DFT_W_VALU"); return e_ && (e_[0] == '1' || e_[0] == 'd'); }();
    if (!(w_valu && q.Ho <= DFT_WMAX)) {
        hipLaunchKernelGGL(dft_wtab_kernel, dim3(32), dim3(256), 0, st, EO, ED, q.L, q.Lh, q.Ho, q.LHP, q.NT,
                           q.NT + q.REM1, q.NS, q.NRT);
        TVAE_CHECK_LAUNCH();
        const size_t lds_d = ((size_t)q.NS * q.NRT * 64 + (size_t)4 * (32 * ((2 * q.NS) | 1) + 64)) * 4;
        const long ntiles = (long)q.M * (q.NBpad / 32);
        const int grid = (int)((ntiles + 3) / 4 < 768 ? (ntiles + 3) / 4 : 768);
        const int iters = (int)((ntiles + 4L * grid - 1) / (4L * grid));
        hipError_t e0 = hipSuccess;
#define TVAE_DY_MF(S_, T_, L2_, A_)                                                                                 \
    do {                                                                                                            \
        e0 = allow_big_lds(dft_dy_mf_kernel<S_, T_, L2_, A_>, lds_d);                                               \
        if (e0 != hipSuccess) return (int)e0;                                                                       \
        hipLaunchKernelGGL((dft_dy_mf_kernel<S_, T_, L2_, A_>), dim3(grid), dim3(256), lds_d, st, dpre,             \
                           (const float*)ED, Sp, q.M, R, B, q.Ho, q.Lh, q.NBpad, iters);                            \
    } while (0)
        if (q.NS == 9) { if (q.Lh == 23) TVAE_DY_MF(9, 2, 46, true); else TVAE_DY_MF(9, 2, 0, true); }
        else if (q.NS == 17) { if (q.Lh == 49) TVAE_DY_MF(17, 4, 98, true); else TVAE_DY_MF(17, 4, 0, true); }
        else TVAE_DY_MF(32, 4, 0, false);
#undef TVAE_DY_MF
        TVAE_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(dft_tables_kernel, dim3(8), dim3(256), 0, st, vtab, q.L, q.Lh);
        TVAE_CHECK_LAUNCH();
        const dim3 dg((unsigned)((q.NBpad + 255) / 256), q.M);
        if (q.Ho == 33)
            hipLaunchKernelGGL(dft_dy_kernel<33>, dg, dim3(256), 0, st, dpre, (const float*)vtab, Sp, q.M, R, B, q.Ho, q.Lh,
                               q.NBpad);
        else if (q.Ho == 17)
            hipLaunchKernelGGL(dft_dy_kernel<17>, dg, dim3(256), 0, st, dpre, (const float*)vtab, Sp, q.M, R, B, q.Ho, q.Lh,
                               q.NBpad);
        else
            hipLaunchKernelGGL(dft_dy_kernel<DFT_WMAX>, dg, dim3(256), 0, st, dpre, (const float*)vtab, Sp, q.M, R, B, q.Ho,
                               q.Lh, q.NBpad);
        TVAE_CHECK_LAUNCH();
    }
    if (dbias) {
        hipLaunchKernelGGL(dft_dbias_kernel, dim3(C), dim3(256), 0, st, (const float*)Sp, dbias, R, q.Lh, q.NB, q.M);
        TVAE_CHECK_LAUNCH();
    }
    // G[fx][m'][k] = sum_n S'[fx][m'][n] A^T[fx][k][n]: batched split-pipe weight-gradient GEMM, two reduction slices
    {
        const int M2 = 2 * q.M, tiles_b = q.Mb / DX6_ROWS, tilesM = q.Lh * tiles_b, tilesK = cdiv(q.K2, 128);
        // 8 reduction slices: TileMap deals the slices round-robin to the 8 XCDs, fewer would leave XCDs idle
        const int splits = DFT_WG_SPLITS;
        const int nchunk = cdiv(cdiv((int)q.NBpad, splits), 16) * 16;
        const TileMap tmk{tilesM, tilesK, splits};
        const DenseBatch bt{tiles_b, (long)q.K2 * q.NBpad, 0};
        if (wgrad_dma()) {
            hipError_t e_ = allow_big_lds(dense_wgrad_x6_dma_kernel<false, false, false>, WG_RING_BYTES);
            if (e_ != hipSuccess) return (int)e_;
            hipLaunchKernelGGL((dense_wgrad_x6_dma_kernel<false, false, false>), dim3(tmk.grid()), dim3(DX6_THREADS), WG_RING_BYTES,
                               st, (const float*)Sp, (long)q.Lh * 128, at, q.NBpad, slabs, M2, q.K2, (int)q.NBpad, nchunk,
                               tmk, bt, 128L, VirtGrad{nullptr, nullptr, 0, 0.f},
                               VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, ATile{7, 127, (long)M2 * q.Lh * 128});
        } else {
            hipLaunchKernelGGL((dense_wgrad_x6_kernel<false, false>), dim3(tmk.grid()), dim3(DX6_THREADS), 0, st,
                               (const float*)Sp, (long)q.Lh * 128, at, q.NBpad, slabs, M2, q.K2, (int)q.NBpad, nchunk, tmk,
                               bt, 128L, VirtGrad{nullptr, nullptr, 0, 0.f},
                               VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, ATile{7, 127, (long)M2 * q.Lh * 128});
        }
        TVAE_CHECK_LAUNCH();
        Epilogue ep;
        ep.C = G; ep.ldc = q.K2;
        const long per = (long)q.Lh * M2 * q.K2;
        int blocks = cdiv(per, 64);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, st, (const float*)slabs, splits,
                           q.Lh * M2, q.K2, ep);
        TVAE_CHECK_LAUNCH();
    }
    const size_t lds_db = (size_t)q.L * q.Lh * 8 + (size_t)ksz * q.Lh * 8 + (size_t)q.L * 8;
    hipError_t e = allow_big_lds(dft_dbank_kernel, lds_db);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dft_dbank_kernel, dim3(q.M), dim3(256), lds_db, st, (const float*)G, dbank, ksz, q.L, q.Lh, q.M);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_rotate_bank_fwd(const float* weight, const int* tap_idx, const float* tap_w, float* bank, int C, int Cin,
                         int ksz, int R, tvae_stream_t stream) {
    const int k2 = ksz * ksz;
    const long total = (long)C * R * Cin * k2;
    hipLaunchKernelGGL(rotate_bank_fwd_kernel, dim3(grid1d(total, 256)), dim3(256), 0, S(stream), weight, tap_idx,
                       tap_w, bank, C, Cin, k2, R);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_rotate_bank_bwd(const float* dbank, const int* csr_ptr, const int* csr_r, const int* csr_dst,
                         const float* csr_w, float* dweight, int C, int Cin, int ksz, int R, int accumulate,
                         tvae_stream_t stream) {
    const int k2 = ksz * ksz;
    const long total = (long)C * Cin * k2;
    hipLaunchKernelGGL(rotate_bank_bwd_kernel, dim3(grid1d(total, 256)), dim3(256), 0, S(stream), dbank, csr_ptr,
                       csr_r, csr_dst, csr_w, dweight, C, Cin, k2, R, accumulate);
    TVAE_CHECK_LAUNCH();
    return 0;
}


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
    if (g_gemm_mode == 1) {
        LoadKContig8 al8{bank, (long)K, M};
        LoadConvPatchFwd bl8{y, g, N};
        return (int)launch_gemm_bf16x3(al8, bl8, ep, M, N, K, 1, nullptr, 0, S(stream));
    }
    const int rows = conv_fwd_img_rows(n, ksz, pad);
    const size_t lds = conv_img_lds_bytes(Cin, rows, n, pad);
    if (lds <= CONV_IMG_LDS_MAX) {
        // image-resident path: the padded-image rows of the tile in LDS, B fragments read straight from them
        const int tilesPerImg = cdiv(g.P, BN);
        const long nblk = (long)cdiv(M, BM) * B * tilesPerImg;
        if (nblk > 2147483647L) return (int)hipErrorInvalidValue;
        const bool vec = (K % BK == 0) && (M % BM == 0);
        static const bool wide = [] { const char* e = getenv("TVAE_CONV1_WIDE"); return !(e && e[0] == '0'); }();
        const size_t lds2 = conv_img_lds_bytes(Cin, rows, n, pad, 2);
        hipError_t e;
        if (vec && wide && M % (2 * BM) == 0 && lds2 <= CONV_IMG_LDS_MAX) {
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
    if (g_gemm_mode == 1) {
        LoadConvDY8 al8{dpre, (long)B * R * g.P, M, R, g.P};
        LoadConvPatchWgrad bl8{y, g, N};
        return (int)launch_gemm_bf16x3(al8, bl8, ep, M, N, K, pick_splits(tiles, K), ws, ws_floats, S(stream));
    }
    const int rows = conv_wgrad_img_rows(Cin, n, ksz, pad);
    const size_t lds = conv_img_lds_bytes(Cin, rows, n, pad);
    if (lds <= CONV_IMG_LDS_MAX) {
        static const bool wide_env = [] { const char* e = getenv("TVAE_CONV1_WIDE"); return e && e[0] == '1'; }();
        const size_t lds2w = conv_img_lds_bytes(Cin, rows, n, pad, 2);
        const bool use_wide = wide_env && M % (2 * BM) == 0 && lds2w <= CONV_IMG_LDS_MAX;
        // split the image reduction into ~4 waves of resident workgroups (256 CUs x 3 or 2 per CU): zero skipping makes
        // edge-tap tiles up to 2x lighter, and many smaller slices let the dispatcher balance that (sweep at cfg4:
        // 3 / 6 / 12 / 32 slices -> 20.4 / 20.1 / 19.65 / 19.7 ms)
        const int out_tiles = use_wide ? (M / (2 * BM)) * tilesN : tiles;
        const int capacity = 4 * 256 * (use_wide ? 2 : 3);
        int splits = (capacity + out_tiles / 2) / out_tiles;
        static const int splits_env = [] { const char* e = getenv("TVAE_CONV1_WGRAD_SPLITS"); return e ? atoi(e) : 0; }();
        if (splits_env > 0) splits = splits_env;
        if (splits < 1) splits = 1;
        if (splits > B) splits = B;
        const long per = (long)M * N;
        const long cap = ws ? ws_floats / per : 0;
        if (cap < 2) splits = 1; else if (splits > cap) splits = (int)cap;
        if (splits < 1) splits = 1;
        const int ips = cdiv(B, splits);
        splits = cdiv(B, ips);
        const size_t lds2 = lds2w;
        static const int nh_env = [] { const char* e = getenv("TVAE_CONV1_WGRAD_NH"); return e ? atoi(e) : 2; }();
        {
            // 128 x 256 tile (wave 64 x 128): every dY panel is re-read by half as many tap-tiles
            const int rows2 = conv_wgrad_img_rows(Cin, n, ksz, pad, 2);
            const size_t ldsn = conv_img_lds_bytes(Cin, rows2, n, pad, 1, 32);
            if (nh_env == 2 && !use_wide && N % (2 * BN) == 0 && ldsn * 2 <= 160 * 1024) {
                const int tilesN2 = N / (2 * BN);
                const int otiles = tilesM * tilesN2;
                int sp = (4 * 256 * 2 + otiles / 2) / otiles;
                if (splits_env > 0) sp = splits_env;
                if (sp > B) sp = B;
                if (cap < 2) sp = 1; else if (sp > cap) sp = (int)cap;
                if (sp < 1) sp = 1;
                const int ips2 = cdiv(B, sp);
                sp = cdiv(B, ips2);
                hipError_t e = allow_big_lds(conv1_wgrad_img_kernel<1, 32, 2>, ldsn);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL((conv1_wgrad_img_kernel<1, 32, 2>), dim3((unsigned)(otiles * sp)), dim3(GEMM_THREADS),
                                   ldsn, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips2, sp > 1 ? ws : nullptr,
                                   tilesN2, rows2, sp, tilesM * sp);
                TVAE_CHECK_LAUNCH();
                if (sp > 1) {
                    int blocks = cdiv(per, 64);
                    if (blocks > 16384) blocks = 16384;
                    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, S(stream), (const float*)ws, sp,
                                       M, N, ep);
                    TVAE_CHECK_LAUNCH();
                }
                return 0;
            }
        }
        static const int kb_env = [] { const char* e = getenv("TVAE_CONV1_WGRAD_KB"); return e ? atoi(e) : 32; }();
        const size_t lds32 = conv_img_lds_bytes(Cin, rows, n, pad, 1, 32);
        if (use_wide) {
            const int tiles2 = out_tiles;
            hipError_t e = allow_big_lds(conv1_wgrad_img_kernel<2, 16, 1>, lds2);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_wgrad_img_kernel<2, 16, 1>), dim3((unsigned)(tiles2 * splits)),
                               dim3(GEMM_THREADS), lds2, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips,
                               splits > 1 ? ws : nullptr, tilesN, rows, splits, (tiles2 / tilesN) * splits);
        } else if (kb_env == 32 && lds32 * 3 <= 160 * 1024) {
            // 32 positions per k-step: half the barriers, still 3 workgroups per CU
            hipError_t e = allow_big_lds(conv1_wgrad_img_kernel<1, 32, 1>, lds32);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_wgrad_img_kernel<1, 32, 1>), dim3((unsigned)(tiles * splits)),
                               dim3(GEMM_THREADS), lds32, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips,
                               splits > 1 ? ws : nullptr, tilesN, rows, splits, tilesM * splits);
        } else {
            hipError_t e = allow_big_lds(conv1_wgrad_img_kernel<1, 16, 1>, lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((conv1_wgrad_img_kernel<1, 16, 1>), dim3((unsigned)(tiles * splits)),
                               dim3(GEMM_THREADS), lds, S(stream), dpre, (long)B * R * g.P, y, g, ep, M, N, ips,
                               splits > 1 ? ws : nullptr, tilesN, rows, splits, tilesM * splits);
        }
        TVAE_CHECK_LAUNCH();
        if (splits > 1) {
            int blocks = cdiv(per, 64);
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, S(stream), (const float*)ws, splits,
                               M, N, ep);
            TVAE_CHECK_LAUNCH();
        }
        return 0;
    }
    LoadConvDY al{dpre, (long)B * R * g.P, M, R, g.P};
    LoadConvPatchWgrad bl{y, g, N};
    return (int)launch_gemm(al, bl, ep, M, N, K, pick_splits(tiles, K), ws, ws_floats, S(stream));
}

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
    if (g_gemm_mode == 1) {
        LoadKContig8 al8{W, (long)K, M};
        return (int)launch_gemm_bf16x3(al8, bl, ep, M, N, K, 1, nullptr, 0, S(stream));
    }
    if (M % BM == 0 && N % BN == 0 && K % BK == 0 && ldx % 4 == 0 && aligned16(W) && aligned16(X)) {
        LoadKContigV4 af{W, (long)K, M};
        if (use_glds()) {
            const TileMap tm{M / BM, N / BN, 1};
            hipLaunchKernelGGL((gemm_f32_glds_kernel<LoadKContigV4>), dim3(tm.grid()), dim3(GEMM_THREADS), 0, S(stream),
                               af, X, ldx, ep, M, N, K, tm, vec_epilogue_ok(ep));
            TVAE_CHECK_LAUNCH();
            return 0;
        }
        LoadXContigV4 bf{X, ldx, N};
        return (int)launch_gemm(af, bf, ep, M, N, K, 1, nullptr, 0, S(stream));
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
    if (g_gemm_mode == 1) return (int)launch_gemm_bf16x3(al, bl, ep, K, N, M, 1, nullptr, 0, S(stream));
    if (use_glds() && K % BM == 0 && N % BN == 0 && M % BK == 0 && ldd % 4 == 0 && aligned16(W) && aligned16(dpre)) {
        // both operands are row-contiguous along their tile dimension: A(kout, m) = W[m][kout], B = dpre[m][n]
        const TileMap tm{K / BM, N / BN, 1};
        hipLaunchKernelGGL(gemm_f32_glds2_kernel, dim3(tm.grid()), dim3(GEMM_THREADS), 0, S(stream), W, (long)K, dpre,
                           ldd, ep, K, N, M, tm, vec_epilogue_ok(ep));
        TVAE_CHECK_LAUNCH();
        return 0;
    }
    if (K % BM == 0 && N % BN == 0 && M % BK == 0 && ldd % 4 == 0 && K % 4 == 0 && aligned16(W) && aligned16(dpre)) {
        LoadXContigV4 af{W, (long)K, K};
        LoadXContigV4 bf{dpre, ldd, N};
        return (int)launch_gemm(af, bf, ep, K, N, M, 1, nullptr, 0, S(stream));
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
    if (g_gemm_mode == 1) {
        LoadKContig8 al8{dpre, ldd, M};
        LoadKContig8 bl8{X, ldx, K};
        return (int)launch_gemm_bf16x3(al8, bl8, ep, M, K, N, pick_splits(tiles, N), ws, ws_floats, S(stream));
    }
    if (M % BM == 0 && K % BN == 0 && N % (BK * 1) == 0 && ldd % 4 == 0 && ldx % 4 == 0 && aligned16(dpre) &&
        aligned16(X)) {
        // split-K chunks are multiples of BK, and N % BK == 0, so every k-step of every slice is full
        LoadKContigV4 af{dpre, ldd, M};
        LoadKContigV4 bf{X, ldx, K};
        return (int)launch_gemm(af, bf, ep, M, K, N, pick_splits(tiles, N), ws, ws_floats, S(stream));
    }
    return (int)launch_gemm(al, bl, ep, M, K, N, pick_splits(tiles, N), ws, ws_floats, S(stream));
}

int tvae_rowdot_seg(const float* X, long ldx, const float* V, int no, int M, int N, int seglen, float* out,
                    tvae_stream_t stream) {
    if (seglen <= 0 || M <= 0) return (int)hipErrorInvalidValue;
    const int nseg = (N + seglen - 1) / seglen;
    dim3 grid(M, nseg), block(256);
    if (!V) no = 1;
    switch (no) {
        case 1: hipLaunchKernelGGL(rowdot_seg_kernel<1>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M); break;
        case 2: hipLaunchKernelGGL(rowdot_seg_kernel<2>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M); break;
        case 3: hipLaunchKernelGGL(rowdot_seg_kernel<3>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M); break;
        case 4: hipLaunchKernelGGL(rowdot_seg_kernel<4>, grid, block, 0, S(stream), X, ldx, V, N, seglen, out, M); break;
        default: return (int)hipErrorInvalidValue;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_seg_sum(const float* in, int S_, long L, float* out, float scale, int accumulate, tvae_stream_t stream) {
    hipLaunchKernelGGL(seg_sum_kernel, dim3(grid1d(L, 256)), dim3(256), 0, S(stream), in, S_, L, out, scale,
                       accumulate);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_coldot(const float* X, long ldx, int M, int N, const float* W, int wsm, int wso, const float* bias, int no,
                float* out, tvae_stream_t stream) {
    dim3 grid((N + 255) / 256), block(256);
    const size_t sh = (size_t)M * no * sizeof(float);
    switch (no) {
        case 1: hipLaunchKernelGGL(coldot_kernel<1>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        case 2: hipLaunchKernelGGL(coldot_kernel<2>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        case 3: hipLaunchKernelGGL(coldot_kernel<3>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        case 4: hipLaunchKernelGGL(coldot_kernel<4>, grid, block, sh, S(stream), X, ldx, M, N, W, wsm, wso, bias, out); break;
        default: return (int)hipErrorInvalidValue;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_outer_mask(const float* dy, int no, const float* W, int wsm, int wso, const float* H, long ldh, float* D,
                    long ldd, int M, int N, int act, float slope, tvae_stream_t stream) {
    dim3 grid((N + 255) / 256, (M + 15) / 16), block(256);
    switch (no) {
        case 1: hipLaunchKernelGGL(outer_mask_kernel<1>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        case 2: hipLaunchKernelGGL(outer_mask_kernel<2>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        case 3: hipLaunchKernelGGL(outer_mask_kernel<3>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        case 4: hipLaunchKernelGGL(outer_mask_kernel<4>, grid, block, 0, S(stream), dy, W, wsm, wso, H, ldh, D, ldd, M, N, act, slope); break;
        default: return (int)hipErrorInvalidValue;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

// ---- fused tails of the two MLPs (fused_tail_kernels.hpp) --------------------------------------------------

int tvae_dec_out_bwd(const float* gy, int n_out, const float* Wo, const float* H, long ldh, float* D, long ldd, int F,
                     long N, int act, float slope, float* part, long part_floats, float* tot, tvae_stream_t stream) {
    if (F <= 0 || N <= 0) return 0;
    const int np = panels_of(N, PANEL16);
    if (n_out < 1 || n_out > 4 || part_floats < (long)np * F * (1 + n_out)) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(H) && (!D || aligned16(D)) && ldh % 4 == 0 && ldd % 4 == 0) ? 1 : 0;   // D may be NULL
    dim3 grid(np), block(256);
    switch (n_out) {
        case 1: hipLaunchKernelGGL(dec_out_bwd_kernel<1>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
        case 2: hipLaunchKernelGGL(dec_out_bwd_kernel<2>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
        case 3: hipLaunchKernelGGL(dec_out_bwd_kernel<3>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
        default: hipLaunchKernelGGL(dec_out_bwd_kernel<4>, grid, block, 0, S(stream), gy, Wo, H, ldh, D, ldd, F, N, act, slope, part, vec); break;
    }
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(part_total_kernel, dim3(F), dim3(256), 0, S(stream), (const float*)part, np, F, 1 + n_out, tot);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_dec_in_bwd(const float* d, long ldd, const float* xr, const float* Wc, int F, int B, int Np, float* gxr,
                    float* Simg, float* dbc, float* dWc, float* part, long part_floats, tvae_stream_t stream) {
    if (F <= 0 || B <= 0 || Np <= 0) return 0;
    const int cpi = panels_of(Np, PANEL16);
    if (part_floats < (long)B * cpi * F * 3) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(d) && ldd % 4 == 0 && Np % 4 == 0) ? 1 : 0;
    hipLaunchKernelGGL(dec_in_bwd_kernel, dim3(B * cpi), dim3(256), 0, S(stream), d, ldd, xr, Wc, F, Np, cpi, gxr, part,
                       vec);
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(dec_in_total_kernel, dim3(F), dim3(256), 0, S(stream), (const float*)part, B, cpi, F, Simg, dbc,
                       dWc);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_heads_fwd(const float* W, const float* X, long ldx, const float* bias, float* Y, long ldy, int nh, int C,
                   long N, tvae_stream_t stream) {
    if (N <= 0 || C <= 0) return 0;
    if (nh < 1 || nh > 8) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(X) && aligned16(Y) && ldx % 4 == 0 && ldy % 4 == 0) ? 1 : 0;
    switch (nh) {
        case 1: launch_heads_fwd<1>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 2: launch_heads_fwd<2>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 3: launch_heads_fwd<3>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 4: launch_heads_fwd<4>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 5: launch_heads_fwd<5>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 6: launch_heads_fwd<6>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        case 7: launch_heads_fwd<7>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
        default: launch_heads_fwd<8>(W, X, ldx, bias, Y, ldy, C, N, vec, S(stream)); break;
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_heads_bwd(const float* W, const float* dY, long ldy, const float* X, long ldx, float* dX, long lddx, int nh,
                   int C, long N, int act, float slope, float* part, long part_floats, float* tot,
                   tvae_stream_t stream) {
    if (N <= 0 || C <= 0) return 0;
    const int np = panels_of(N, PANEL8);
    if (nh < 1 || nh > 8 || part_floats < (long)np * C * (nh + 1)) return (int)hipErrorInvalidValue;
    const int vec = (aligned16(X) && aligned16(dX) && aligned16(dY) && ldx % 4 == 0 && lddx % 4 == 0 && ldy % 4 == 0) ? 1 : 0;
    switch (nh) {
        case 1: launch_heads_bwd<1>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 2: launch_heads_bwd<2>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 3: launch_heads_bwd<3>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 4: launch_heads_bwd<4>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 5: launch_heads_bwd<5>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 6: launch_heads_bwd<6>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        case 7: launch_heads_bwd<7>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
        default: launch_heads_bwd<8>(W, dY, ldy, X, ldx, dX, lddx, C, N, act, slope, part, vec, S(stream)); break;
    }
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(part_total_kernel, dim3(C), dim3(256), 0, S(stream), (const float*)part, np, C, nh + 1, tot);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_act_bwd(const float* dY, const float* Y, float* dpre, long n, int act, float slope, tvae_stream_t stream) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d(n, 256)), dim3(256), 0, S(stream), dY, Y, dpre, n, act, slope);
    TVAE_CHECK_LAUNCH();
    return 0;
}

static HeadParams make_head(const float* heads, long ldh, const float* E, const float* eps_z, const float* eps_t,
                            const float* p_r, const float* off, const float* p_tr, const float* grid, int R, int P,
                            int zd, float sigma_p, float theta_off_scale) {
    HeadParams hp;
    hp.heads = heads; hp.ldh = ldh; hp.E = E; hp.eps_z = eps_z; hp.eps_t = eps_t;
    hp.p_r = p_r; hp.off = off; hp.p_tr = p_tr; hp.grid = grid;
    hp.R = R; hp.P = P; hp.zd = zd; hp.sigma_p = sigma_p; hp.theta_off_scale = theta_off_scale;
    return hp;
}

int tvae_attn_head_fwd(const float* heads, long ldh, const float* E, const float* eps_z, const float* eps_t,
                       const float* p_r, const float* off, const float* p_tr, const float* grid, int B, int R, int P,
                       int zd, float sigma_p, float theta_off_scale, float* attn, float* q, float* a, float* z,
                       float* theta, float* dx, float* kl, tvae_stream_t stream) {
    if (B <= 0) return 0;
    HeadParams hp = make_head(heads, ldh, E, eps_z, eps_t, p_r, off, p_tr, grid, R, P, zd, sigma_p, theta_off_scale);
    hipLaunchKernelGGL(attn_head_fwd_kernel, dim3(B), dim3(1024), 0, S(stream), hp, attn, q, a, z, theta, dx, kl);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_attn_head_bwd(const float* heads, long ldh, const float* q, const float* a, const float* eps_z,
                       const float* eps_t, const float* p_r, const float* off, const float* p_tr, const float* grid,
                       int B, int R, int P, int zd, float sigma_p, float theta_off_scale, const float* gz,
                       const float* gth, const float* gdx, const float* gkl, const float* g_attn, const float* g_q,
                       const float* g_a, float* dheads, tvae_stream_t stream) {
    if (B <= 0) return 0;
    HeadParams hp = make_head(heads, ldh, nullptr, eps_z, eps_t, p_r, off, p_tr, grid, R, P, zd, sigma_p,
                              theta_off_scale);
    hipLaunchKernelGGL(attn_head_bwd_kernel, dim3(B), dim3(1024), 0, S(stream), hp, q, a, gz, gth, gdx, gkl, g_attn,
                       g_q, g_a, dheads);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_get_latent(const float* heads, long ldh, const float* p_r, const float* off, const float* grid, int B, int R,
                    int P, int zd, float theta_off_scale, float* zc, float* theta_mu, float* dx, tvae_stream_t stream) {
    if (B <= 0) return 0;
    HeadParams hp = make_head(heads, ldh, nullptr, nullptr, nullptr, p_r, off, nullptr, grid, R, P, zd, 1.f,
                              theta_off_scale);
    hipLaunchKernelGGL(get_latent_kernel, dim3(B), dim3(1024), 0, S(stream), hp, zc, theta_mu, dx);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_coord_fwd(const float* xc, const float* dx, const float* theta, float* xr, int B, int Np,
                   tvae_stream_t stream) {
    hipLaunchKernelGGL(coord_fwd_kernel, dim3(grid1d((long)B * Np, 256)), dim3(256), 0, S(stream), xc, dx, theta, xr,
                       B, Np);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_coord_bwd(const float* xc, const float* dx, const float* theta, const float* gxr, float* gdx, float* gtheta,
                   int B, int Np, tvae_stream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(coord_bwd_kernel, dim3(B), dim3(256), 0, S(stream), xc, dx, theta, gxr, gdx, gtheta, Np);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_dec_l0_fwd(const float* xr, const float* Wc, const float* bc, const float* LB, float* h, long ldh, int F,
                    long Ntot, int Np, int act, float slope, tvae_stream_t stream) {
    dim3 grid((unsigned)((Ntot + 255) / 256), (F + 15) / 16), block(256);
    hipLaunchKernelGGL(dec_l0_fwd_kernel, grid, block, 0, S(stream), xr, Wc, bc, LB, h, ldh, F, Ntot, Np, act, slope);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_latent_bias(const float* Wl, const float* z, float* LB, int B, int F, int zd, tvae_stream_t stream) {
    hipLaunchKernelGGL(latent_bias_kernel, dim3((B * F + 255) / 256), dim3(256), 0, S(stream), Wl, z, LB, B, F, zd);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_latent_bwd(const float* S_, const float* Wl, const float* z, float* dWl, float* dz, int B, int F, int zd,
                    tvae_stream_t stream) {
    const int tot = (F * zd > B * zd) ? F * zd : B * zd;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((tot + 255) / 256), dim3(256), 0, S(stream), S_, Wl, z, dWl, dz, B, F,
                       zd);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_fourier_fwd(const float* xr, const float* Wf, const float* bf, float sigma, float* feat, long ld, int F,
                     long Ntot, tvae_stream_t stream) {
    dim3 grid((unsigned)((Ntot + 255) / 256), (F + 15) / 16), block(256);
    hipLaunchKernelGGL(fourier_fwd_kernel, grid, block, 0, S(stream), xr, Wf, bf, sigma, feat, ld, F, Ntot);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_fourier_bwd(const float* xr, const float* Wf, const float* bf, float sigma, const float* dfeat, long ld,
                     int F, long Ntot, float* gxr, tvae_stream_t stream) {
    hipLaunchKernelGGL(fourier_bwd_kernel, dim3((unsigned)((Ntot + 255) / 256)), dim3(256), 0, S(stream), xr, Wf, bf,
                       sigma, dfeat, ld, F, Ntot, gxr);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_fwd(const float* yh, const float* y, float* lp, int B, int L, int kind, tvae_stream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(loglik_fwd_kernel, dim3(B), dim3(256), 0, S(stream), yh, y, lp, L, kind);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_bwd(const float* yh, const float* y, const float* glp, float* gyh, int B, int L, int kind,
                    tvae_stream_t stream) {
    hipLaunchKernelGGL(loglik_bwd_kernel, dim3(grid1d((long)B * L, 256)), dim3(256), 0, S(stream), yh, y, glp, gyh, B,
                       L, kind);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_ctf_corr(const float* in, const float* ctf, float* out, int B, int n, int kc, int flip, tvae_stream_t stream) {
    if (B <= 0) return 0;
    if ((kc & 1) == 0) return (int)hipErrorInvalidValue;
    dim3 grid((n * n + 255) / 256, B), block(256);
    hipLaunchKernelGGL(ctf_corr_kernel, grid, block, 0, S(stream), in, ctf, out, n, kc, flip);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_masked_fwd(const float* yh, const float* y, const float* dx, float inv_spacing, float radius, int B,
                           int n, float* lp, tvae_stream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(loglik_masked_fwd_kernel, dim3(B), dim3(256), 0, S(stream), yh, y, dx, inv_spacing, radius, n,
                       lp);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_loglik_masked_bwd(const float* yh, const float* y, const float* dx, float inv_spacing, float radius, int B,
                           int n, const float* glp, float* gyh, tvae_stream_t stream) {
    hipLaunchKernelGGL(loglik_masked_bwd_kernel, dim3(grid1d((long)B * n * n, 256)), dim3(256), 0, S(stream), yh, y, dx,
                       inv_spacing, radius, n, glp, gyh, B);
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                   float bc1, float bc2_sqrt, float grad_scale, tvae_stream_t stream) {
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid1d(n, 256, 2048)), dim3(256), 0, S(stream), p, g, m, v, n, lr, b1,
                       b2, eps, bc1, bc2_sqrt, grad_scale);
    TVAE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
