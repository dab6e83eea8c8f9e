// libtvae_hip.so: forward / data-gradient GEMM of the wide dense layers on the bf16 matrix pipe with exactly split
// operands (dense_x6_kernels.hpp: dense_x6_kernel; its three operand variants are compiled in abi_dense_x6_v{0,1,2}.hip),
// plus the weight pre-pass.
#include "abi_dense_x6.hpp"

using namespace tvae;

namespace tvae {
int dense_x6_batched(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem, int rows_total,
                     int N, int K, const TileMap& tm, const DenseBatch& bt, int parts, hipStream_t st, H3Scale hs) {
    const int Rpad = x6_round_up(rows_total, DX6_ROWS), K8pad = dense_k8pad(K);
    if (N % 128 != 0 || !aligned16(w3)) return (int)hipErrorInvalidValue;
    if (parts != 1 && parts != 2 && parts != 3) return (int)hipErrorInvalidValue;
    if (parts == 2 && (!hs.amax_a || !hs.amax_x)) return (int)hipErrorInvalidValue;
    // the spectral contraction's own shape (plain column-tiled output, whole row tiles): lean store epilogue (round 6: the
    // generic one took ~40 % of these launches)
    const bool lean = !ep.bias && !ep.res && !ep.aux && ep.act == ACT_NONE && ep.mask == ACT_NONE && ep.ctile > 0 && ep.C &&
                      !ep.accumulate && !ep.amax_out && rows_per_problem % DX6_ROWS == 0 && ep.ldc * 8 * 4 < (1L << 31);
    if (lean)
        return TVAE_DX6_DISPATCH_E(0, 4, parts, (const uint4*)w3, X, ldx, ep, rows_per_problem, Rpad, N, K, K8pad, tm, bt,
                                   ColDot{nullptr, nullptr, nullptr},
                                   InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1},
                                   VirtGrad{nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0},
                                   VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, st, hs);
    return TVAE_DX6_DISPATCH(0, parts, (const uint4*)w3, X, ldx, ep, rows_per_problem, Rpad, N, K, K8pad, tm, bt,
                             ColDot{nullptr, nullptr, nullptr},
                             InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1},
                             VirtGrad{nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0}, VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, st,
                             hs);
}
int dense_x6_batched4(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem, int rows_total,
                      int N, int K, const TileMap& tm, const DenseBatch& bt, int parts, hipStream_t st, H3Scale hs, bool out_bf16) {
    const int Rpad = x6_round_up(rows_total, DX6_ROWS), K8pad = dense_k8pad(K);     // the cells keep their 512-row padding
    if (N % 128 != 0 || !aligned16(w3) || (parts != 1 && parts != 2 && parts != 3)) return (int)hipErrorInvalidValue;
    if (parts == 2 && (!hs.amax_a || !hs.amax_x)) return (int)hipErrorInvalidValue;
    // the spectral contraction's own shape (plain column-tiled output, whole row tiles) has a lean store epilogue
    const bool lean = !ep.bias && !ep.res && !ep.aux && ep.act == ACT_NONE && ep.mask == ACT_NONE && ep.ctile > 0 && ep.C &&
                      !ep.accumulate && rows_per_problem % DX4_ROWS == 0 && ep.ldc * 8 * 4 < (1L << 31);
#define TVAE_DX4_LAUNCH(NP_, E_)                                                                                       \
    hipLaunchKernelGGL((dense_x6_plain4_kernel<NP_, E_>), dim3(tm.grid()), dim3(DX4_THREADS), 0, st, (const uint4*)w3, X, ldx, \
                       ep, rows_per_problem, Rpad, N, K, K8pad, tm, bt, hs)
    if (out_bf16) {      // bf16 STORAGE of the output (one-part mode only; the caller reads 2-byte elements)
        if (!lean || parts != 1) return (int)hipErrorInvalidValue;
        TVAE_DX4_LAUNCH(1, 2);
        return (int)hipGetLastError();
    }
    if (parts == 3) { if (lean) TVAE_DX4_LAUNCH(3, 1); else TVAE_DX4_LAUNCH(3, 0); }
    else if (parts == 2) { if (lean) TVAE_DX4_LAUNCH(2, 1); else TVAE_DX4_LAUNCH(2, 0); }
    else { if (lean) TVAE_DX4_LAUNCH(1, 1); else TVAE_DX4_LAUNCH(1, 0); }
#undef TVAE_DX4_LAUNCH
    return (int)hipGetLastError();
}
// The same contraction with the streamed panel resident in LDS (dense_x6_xres_kernel): false when the shape is not its own
// (the caller then takes dense_x6_batched4).  ep as for the lean store epilogue; Mb = rows per problem in the cell array.
bool dense_x6_batched_xres(const void* w3, const float* X, long ldx, const Epilogue& ep, int rows_per_problem, int Mb, int nprob,
                           int N, int K, long x_stride, long c_stride, int parts, hipStream_t st, H3Scale hs, int* rc) {
    constexpr bool on = true;
    const int K8pad = dense_k8pad(K), nk = K8pad / 2;
    const size_t lds = (size_t)nk * parts * 256 * 16 + (size_t)(Mb + 128) * 4;
    const bool lean = !ep.bias && !ep.res && !ep.aux && ep.act == ACT_NONE && ep.mask == ACT_NONE && ep.ctile > 0 && ep.C &&
                      !ep.accumulate && rows_per_problem % DX6_ROWS == 0 && ep.ldc * 8 * 4 < (1L << 31);
    // twelve k-steps only (the 96-wide frame of the 64 x 64 configuration): at six steps (28 x 28: 44-wide frame) the stores per
    // step double and the kernel measured 213 us against dense_x6_plain4_kernel's 195; nine steps (50 x 50) were not measured.
    // Two parts (h3): 1.21 against 1.38 ms; three (x6): 1.76 against 1.80.
    // Round 5 (short frame: reduction 2 L = 160 at the 64 x 64 shape): ten steps take it as well.
    // ... and eight (mixed form: reduction 2 ksz = 128).
    const bool shape_ok = (parts == 2 || parts == 3) && (nk == 12 || nk == 10 || nk == 8);
    if (!on || !lean || !shape_ok || N % 128 != 0 || Mb < rows_per_problem ||
        lds > X6_LDS_MAX || !aligned16(w3) || (parts == 2 && (!hs.amax_a || !hs.amax_x)))
        return false;
    const int Rpad = x6_round_up(nprob * Mb, DX6_ROWS), tilesN = N / 128, nch = 4, cs = cdiv(tilesN, nch);
    const unsigned grid = 8u * cdiv(nprob * nch, 8) * cs;
#define TVAE_XRES_LAUNCH(NP_, NK_)                                                                                     \
    do {                                                                                                               \
        hipError_t e_ = allow_big_lds(dense_x6_xres_kernel<NP_, NK_>, lds);                                            \
        if (e_ != hipSuccess) { *rc = (int)e_; return true; }                                                          \
        hipLaunchKernelGGL((dense_x6_xres_kernel<NP_, NK_>), dim3(grid), dim3(256), lds, st, (const uint4*)w3, X, ldx, ep.C,  \
                           ep.ldc, ep.ctile, rows_per_problem, Mb, Rpad, K, nprob, tilesN, nch, x_stride, c_stride, hs);  \
    } while (0)
    if (nk == 12) { if (parts == 3) TVAE_XRES_LAUNCH(3, 12); else TVAE_XRES_LAUNCH(2, 12); }
    else if (nk == 10) { if (parts == 3) TVAE_XRES_LAUNCH(3, 10); else TVAE_XRES_LAUNCH(2, 10); }
    else { if (parts == 3) TVAE_XRES_LAUNCH(3, 8); else TVAE_XRES_LAUNCH(2, 8); }
#undef TVAE_XRES_LAUNCH
    *rc = (int)hipGetLastError();
    return true;
}
}  // namespace tvae

namespace {
// Totals of the per-tile row sums of dense_x6_kernel<4> (VirtGrad.rpart) and the algebra of the two-valued gradient: with
// S0[m] = sum_n gy[n] [H[m][n] > 0], S1[m] = sum_n gy[n] H[m][n] (tiles added in order: deterministic),
//   db[m]  = wo[m] * (slope * sum_n gy[n] + (1 - slope) * S0[m])     bias gradient of the layer that produced H
//   dwo[m] = S1[m]                                                   weight gradient of the single-output Linear
// One workgroup per row m.
__global__ void dgrad_rowsum_total_kernel(const float* __restrict__ part, int ntiles, int K, const float* __restrict__ wo,
                                          const float* __restrict__ gysum, float slope, float* __restrict__ db,
                                          float* __restrict__ dwo, const float* __restrict__ rowdot,
                                          const float* __restrict__ bias) {
    __shared__ float sm[2 * 16];
    const int m = blockIdx.x;
    float s[2] = {0.f, 0.f};
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((long)m * ntiles + t) * 2);
        s[0] += v.x;
        if (!rowdot) s[1] += v.y;                        // (the bits form writes only the first word)
    }
    block_sum<2>(s, sm);
    if (threadIdx.x == 0) {
        const float g0 = __fmaf_rn(1.f - slope, s[0], slope * gysum[0]);      // sum_n gy[n] act'(H[m][n])
        db[m] = wo[m] * g0;
        // dWo[m] = sum_n gy[n] H[m][n]: summed directly (s[1]), or -- H never stored -- from the layer's own weight gradient:
        // sum_k W[m][k] G[m][k] + b[m] g0[m]  (VirtGrad, dense_x6_kernels.hpp; rowdot from wgrad_lrf_finalize_kernel)
        dwo[m] = rowdot ? __fmaf_rn(bias ? bias[m] : 0.f, g0, rowdot[m]) : s[1];
    }
}
}  // namespace

extern "C" {

// ---- dense layers on the bf16 matrix pipe with exactly split operands (dense_x6_kernels.hpp) ---------------------
long tvae_dense_x6_bytes(int rows, int K) { return dense_x6_bytes(rows, K); }
// h3 cells (parts == 2) occupy two of the three part arrays the buffer is sized for; the third starts with one maximum per
// padded row ([Rpad], read again by the GEMM for its scales) and, for launches whose streamed operand is recomputed, the
// 4 + K bound words of dec_l0_bound_kernel behind them (written by the GEMM entry point).
static int dense_split(const float* W, long ldw, void* a3, long a3_bytes, int rows, int K, int transpose,
                       const float* scale, float* rowsum, int parts, tvae_stream_t stream) {
    if (rows <= 0 || K <= 0) return 0;
    if (a3_bytes < tvae_dense_x6_bytes(rows, K) || !aligned16(a3)) return (int)hipErrorInvalidValue;
    const int Rpad = x6_round_up(rows, DX6_ROWS), K8pad = dense_k8pad(K);
    const long total = (long)K8pad * Rpad;
    if (parts == 2) {
        float* tr = h3_trailer(a3, rows, K);             // [Rpad] row maxima, then the scratch words of the GEMM entry points
        if (transpose)
            hipLaunchKernelGGL(dense_rowmax_kernel, dim3(Rpad / 64), dim3(1024), 0, S(stream), W, ldw, rows, Rpad, K, transpose,
                               scale, tr);
        else             // row-major operand: one wave per row, lanes along k
            hipLaunchKernelGGL(dense_rowmax_rows_kernel, dim3(Rpad / 4), dim3(256), 0, S(stream), W, ldw, rows, Rpad, K, scale, tr);
        TVAE_CHECK_LAUNCH();
        if (!transpose && K8pad <= 48)                   // both sides coalesced through LDS (32 rows per workgroup, <= 48 KB)
            hipLaunchKernelGGL(dense_split2h_rows_kernel, dim3(Rpad / 32), dim3(256), (size_t)32 * K8pad * 32, S(stream), W, ldw,
                               (uint4*)a3, rows, Rpad, K, K8pad, scale, (const float*)tr);
        else
            hipLaunchKernelGGL(dense_split2h_kernel, dim3(grid1d(total, 256)), dim3(256), 0, S(stream), W, ldw, (uint4*)a3, rows,
                               Rpad, K, K8pad, transpose, scale, (const float*)tr);
    } else {
        hipLaunchKernelGGL(dense_split3_kernel, dim3(grid1d(total, 256)), dim3(256), 0, S(stream), W, ldw, (uint4*)a3, rows,
                           Rpad, K, K8pad, transpose, scale);
    }
    TVAE_CHECK_LAUNCH();
    if (rowsum) {
        hipLaunchKernelGGL(dense_rowsum_kernel, dim3((rows + 63) / 64), dim3(1024), 0, S(stream), W, ldw, rows, K, transpose,
                           scale, rowsum);
        TVAE_CHECK_LAUNCH();
    }
    return 0;
}
int tvae_dense_split3(const float* W, long ldw, void* a3, long a3_bytes, int rows, int K, int transpose,
                      const float* scale, float* rowsum, tvae_stream_t stream) {
    return dense_split(W, ldw, a3, a3_bytes, rows, K, transpose, scale, rowsum, 3, stream);
}
int tvae_dense_split2h(const float* W, long ldw, void* a3, long a3_bytes, int rows, int K, int transpose,
                       const float* scale, float* rowsum, tvae_stream_t stream) {
    return dense_split(W, ldw, a3, a3_bytes, rows, K, transpose, scale, rowsum, 2, stream);
}
static int launch_dense_x6(const void* a3, const float* X, long ldx, const Epilogue& ep, int rows, int N, int K, int parts,
                           hipStream_t st, ColDot cd = ColDot{nullptr, nullptr, nullptr},
                           InTail it = InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1},
                           VirtGrad vg = VirtGrad{nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0},
                           VirtAct va = VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, const float* x_amax = nullptr) {
    if ((cd.w || it.xr) && rows > DX6_ROWS) return (int)hipErrorInvalidValue;   // the fused tails need ONE row tile
    if (rows <= 0 || N <= 0) return 0;
    if (N % 128 != 0 || !aligned16(a3) || (parts != 1 && parts != 2 && parts != 3)) return (int)hipErrorInvalidValue;
    const int Rpad = x6_round_up(rows, DX6_ROWS), K8pad = dense_k8pad(K);
    TileMap tm{Rpad / DX6_ROWS, N / 128, 1};
    tm.pack = 1;                                         // eight adjacent column tiles per XCD (TileMap: the sign-bit lines)
    H3Scale hs = H3_NONE;
    if (parts == 2) {
        // h3 instances: the recomputed first-layer activation (its bound is formed here) and the exact 0 / 1 operand of the
        // two-valued gradient; an operand streamed from memory would need its maximum from its producer (not wired: x6)
        float* tr = h3_trailer(a3, rows, K);             // one maximum per row (tvae_dense_split2h)
        hs.amax_a = tr;
        hs.a_rows = 1;
        if (va.xr) {
            float* bw = tr + Rpad;                       // bound words of the recomputed operand: [4 + K]
            hipLaunchKernelGGL(h3_zero_slots_kernel, dim3(1), dim3(256), 0, st, bw, 4 + K);
            TVAE_CHECK_LAUNCH();
            const long nlb = va.lb ? (long)(N / va.Np) * K : 0;
            hipLaunchKernelGGL(dec_l0_bound_kernel, dim3(grid1d(N / 2, 256, 512)), dim3(256), 0, st, va.xr, 2L * N, va.wc,
                               va.bc, va.lb, nlb, K, bw);
            TVAE_CHECK_LAUNCH();
            hs.amax_x = bw;
        } else if (x_amax && !vg.wo && !vg.csum) {
            // an operand streamed from memory: the caller supplies max |X| or an upper bound of it (one device word) -- e.g.
            // |cos| <= 1 for Fourier features, the first-layer bound for a stored coordinate layer, a row-sum bound for the
            // output of a layer (tvae/ops.py: _h3_bounds)
            hs.amax_x = x_amax;
        } else if (!vg.csum) {
            return (int)hipErrorInvalidValue;
        }
    }
    // the recomputed operands need tiles inside one image and tables of <= 512 entries
    if ((va.xr && (K > 512 || va.Np % 128 != 0 || vg.wo || vg.csum)) || (vg.csum && (!vg.gy || vg.act != ACT_LRELU)) || (it.bc && it.Np % 128 != 0) ||
        (!va.xr && !X && !(vg.bits && vg.csum && vg.rpart)))
        return (int)hipErrorInvalidValue;
    const DenseBatch nb{0, 0, 0};
    if (vg.csum && vg.rpart && vg.bits) {                                // operand and row sums from the stored sign bits
        if (K > DX6_ROWS || N % 32 != 0) return (int)hipErrorInvalidValue;
        // the hot shape has its own lean instance: ONE full row tile, result not stored, fused first-layer backward with the
        // recomputed LeakyReLU mask, nothing else switched on
        if (rows == DX6_ROWS && !ep.C && it.xr && it.bc && !ep.res && ep.mask == ACT_LRELU && !cd.w && !cd.bits)
            return TVAE_DX6_DISPATCH_E(5, 2, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
        // ... its result stored under the LeakyReLU mask of a saved activation (Fourier decoders), whole row tiles: lean store
        if (rows % DX6_ROWS == 0 && ep.C && !it.xr && !ep.res && ep.mask == ACT_LRELU && ep.aux && !cd.w && !cd.bits && !ep.ctile &&
            ep.ldc * 8 * 4 < (1L << 31) && ep.ldaux * 8 * 4 < (1L << 31))
            return TVAE_DX6_DISPATCH_E(5, 3, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
        return TVAE_DX6_DISPATCH(5, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    }
    if (vg.csum && vg.rpart) {
        if (K > DX6_ROWS) return (int)hipErrorInvalidValue;              // the row sums live in one 512-row LDS table
        return TVAE_DX6_DISPATCH(4, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    }
    if (vg.csum) return TVAE_DX6_DISPATCH(3, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    // forward of the decoder's last hidden layer with its activation not stored: lean instance (one full row tile)
    // (cd.bits == nullptr with no output either: the inference-mode forward -- only the fused column dot leaves the launch)
    if (va.xr && rows == DX6_ROWS && !ep.C && cd.w && !it.xr && !ep.res && ep.mask == ACT_NONE && ep.act == ACT_LRELU)
        return TVAE_DX6_DISPATCH_E(2, 1, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    if (va.xr) return TVAE_DX6_DISPATCH(2, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    // the same layer with its input read from memory (28 x 28 shapes, Fourier decoders): lean instance of the plain operand
    if (!vg.wo && X && rows == DX6_ROWS && N % 128 == 0 && !ep.C && cd.w && !it.xr && !ep.res && ep.mask == ACT_NONE &&
        ep.act == ACT_LRELU)
        return TVAE_DX6_DISPATCH_E(0, 1, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    if (vg.wo) return TVAE_DX6_DISPATCH(1, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    // plain hidden layer that stores its output (round 6): whole 512-row tiles, bias + LeakyReLU | none (forward) or the
    // LeakyReLU mask of the saved activation (data gradient), nothing fused behind it: lean store epilogue
    if (X && rows % DX6_ROWS == 0 && ep.C && !ep.res && !cd.w && !cd.bits && !it.xr && !ep.ctile && !ep.accumulate && !ep.gbias &&
        ((ep.mask == ACT_NONE && (ep.act == ACT_LRELU || ep.act == ACT_NONE)) || (ep.mask == ACT_LRELU && ep.aux && ep.act == ACT_NONE && !ep.bias)) &&
        ep.ldc * 8 * 4 < (1L << 31) && ep.ldaux * 8 * 4 < (1L << 31))
        return TVAE_DX6_DISPATCH_E(0, 3, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
    return TVAE_DX6_DISPATCH(0, parts, (const uint4*)a3, X, ldx, ep, rows, Rpad, N, K, K8pad, tm, nb, cd, it, vg, va, st, hs);
}
int tvae_linear_fwd_x6(const void* w3, const float* X, const float* bias, const float* res, float* Y, int M, int N,
                       int K, long ldx, long ldy, int act, float slope, const float* col_w, const float* col_b,
                       float* col_y, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                       int va_np, void* sign_bits, int parts, const float* x_amax, float* y_amax, tvae_stream_t stream) {
    Epilogue ep;
    ep.C = Y; ep.ldc = ldy;
    // y_amax (ABI 7): max |Y| of what this launch stores, by atomic max into a word the caller has zeroed (generic epilogue only:
    // the launch must store its output and fuse nothing behind it)
    if (y_amax && (!Y || col_w || sign_bits)) return (int)hipErrorInvalidValue;
    ep.amax_out = y_amax;
    if (sign_bits && (act != ACT_LRELU || N % 32 != 0)) return (int)hipErrorInvalidValue;
    ep.bias = bias;
    ep.res = res; ep.ldres = ldy;
    ep.act = act; ep.slope = slope;
    return launch_dense_x6(w3, X, ldx, ep, M, N, K, parts, S(stream), ColDot{col_w, col_b, col_y, (unsigned*)sign_bits},
                           InTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1}, VirtGrad{nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0},
                           VirtAct{va_xr, va_wc, va_bc, va_lb, va_np > 0 ? va_np : 1, act, slope}, x_amax);
}
int tvae_linear_dgrad_x6(const void* w3t, const float* dpre, const float* add, const float* aux, float* dX, int M,
                         int N, int K, long ldd, long ldx, int mask, float slope, const float* in_xr,
                         const float* in_wc, float* in_gxr, float* in_part, long in_part_floats, const float* vg_wo,
                         const float* vg_gy, const float* vg_csum, const float* in_bc, const float* in_lb, int in_np,
                         float* rs_part, long rs_part_floats, const float* rs_wo, const float* rs_gysum, float* rs_db,
                         float* rs_dwo, int parts, const void* vg_bits, const float* rs_rowdot, const float* rs_bias,
                         const float* x_amax, float* y_amax, tvae_stream_t stream) {
    // dX[k][n] = act'(aux[k][n]) * (add[k][n] + sum_m W[m][k] dpre[m][n]): rows = K, reduction = M; w3t = split of W^T
    Epilogue ep;
    ep.C = dX; ep.ldc = ldx;                             // dX may be NULL when the fused first-layer backward consumes it
    if (y_amax && (!dX || in_xr)) return (int)hipErrorInvalidValue;      // y_amax (ABI 7): max |dX| as stored, see tvae_linear_fwd_x6
    ep.amax_out = y_amax;
    ep.res = add; ep.ldres = ldx;
    ep.aux = aux; ep.ldaux = ldx;
    ep.mask = (aux || in_bc) ? mask : ACT_NONE; ep.slope = slope;
    if (in_bc && !in_xr) return (int)hipErrorInvalidValue;               // the recomputed mask needs the coordinates
    if (in_xr) {
        if (!in_wc || !in_gxr || !in_part || in_part_floats < (long)(N / 128) * K * 3) return (int)hipErrorInvalidValue;
    } else if (!dX) {
        return (int)hipErrorInvalidValue;
    }
    if (vg_bits && !(vg_csum && rs_part && (rs_rowdot || !rs_db) && N % 32 == 0))     // the bits form: two-valued, with its row sums
        return (int)hipErrorInvalidValue;                                // and dWo from the weight-gradient identity
    if (rs_rowdot && !vg_bits) return (int)hipErrorInvalidValue;
    if (rs_part) {       // row sums of the streamed activation (two-valued form only; M = rows of H <= 512)
        // (rs_db == rs_dwo == NULL: partial sums only -- the caller totals them later with tvae_dgrad_rowsum_total, e.g. on the
        //  stream that also runs the weight gradient whose rd_rowdot the totals need: tvae/ops.py side stream)
        if (!vg_csum || !vg_gy || !rs_wo || !rs_gysum || (!rs_db != !rs_dwo) || M > DX6_ROWS || N % 128 != 0 ||
            rs_part_floats < (long)(N / 128) * M * 2 || (reinterpret_cast<size_t>(rs_part) & 7))
            return (int)hipErrorInvalidValue;
    }
    int rc = launch_dense_x6(w3t, dpre, ldd, ep, K, N, M, parts, S(stream), ColDot{nullptr, nullptr, nullptr},
                             InTail{in_xr, in_wc, in_gxr, in_part, in_bc, in_lb, in_np > 0 ? in_np : 1},
                             VirtGrad{vg_wo, vg_gy, mask, slope, vg_csum, (const unsigned*)vg_bits, rs_part, 0},
                             VirtAct{nullptr, nullptr, nullptr, nullptr, 1, 0, 0.f}, x_amax);
    if (rc || !rs_part || !rs_db || N <= 0 || K <= 0) return rc;
    hipLaunchKernelGGL(dgrad_rowsum_total_kernel, dim3(M), dim3(256), 0, S(stream), (const float*)rs_part, N / 128, M, rs_wo,
                       rs_gysum, slope, rs_db, rs_dwo, rs_rowdot, rs_bias);
    TVAE_CHECK_LAUNCH();
    return 0;
}
// The totals of tvae_linear_dgrad_x6's row sums as an entry point of their own (ABI 6): rs_part [M][ntiles][2] as that launch
// left it (called with rs_db = rs_dwo = NULL), everything else as there.  from_bits != 0: the bits form (dwo from rs_rowdot).
int tvae_dgrad_rowsum_total(const float* rs_part, int ntiles, int M, const float* rs_wo, const float* rs_gysum, float slope,
                            float* rs_db, float* rs_dwo, const float* rs_rowdot, const float* rs_bias, tvae_stream_t stream) {
    if (!rs_part || !rs_wo || !rs_gysum || !rs_db || !rs_dwo || ntiles <= 0 || M <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(dgrad_rowsum_total_kernel, dim3(M), dim3(256), 0, S(stream), rs_part, ntiles, M, rs_wo, rs_gysum, slope,
                       rs_db, rs_dwo, rs_rowdot, rs_bias);
    TVAE_CHECK_LAUNCH();
    return 0;
}


}  // extern "C"
