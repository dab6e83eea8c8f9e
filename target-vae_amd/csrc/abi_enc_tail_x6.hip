// libtvae_hip.so: the encoder tail (conv2 1x1x1 + the stacked head projection) fused per direction on the bf16 matrix
// pipe with exactly split operands (enc_tail_x6_kernels.hpp).
#include "abi_dense_x6.hpp"
#include "enc_tail_x6_kernels.hpp"

using namespace tvae;

namespace {
int cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}
// persistent grid: one workgroup per CU, never more workgroups than there are 64-column chunks for their eight waves
int et_grid(long N) {
    const long chunks = (N + ET_CHUNK - 1) / ET_CHUNK;
    const long want = (chunks + ET_THREADS / 64 - 1) / (ET_THREADS / 64);
    const int cus = cu_count();
    return (int)(want < cus ? want : cus);
}
// the kernels form lane offsets as 32-bit BYTE offsets of up to 32 rows: 32 * ld * 4 must stay below 2^32
constexpr long ET_MAX_LD = 1L << 25;
inline bool et_ld_ok(long N, long a, long b, long c) {
    return N < ET_MAX_LD && a < ET_MAX_LD && b < ET_MAX_LD && c < ET_MAX_LD;
}
}  // namespace

extern "C" {

int tvae_enc_tail_fwd_x6(const void* w3, const float* A1, long lda, const float* b2, const float* Wh, const float* bh,
                         int nh, float* H, long ldh, float* heads, long ldo, void* bits_h, void* bits_a, int C, long N,
                         int act, float slope, int parts, const float* amax_a1, tvae_stream_t stream) {
    if (N <= 0) return 0;
    if (parts == 2 && !amax_a1) return (int)hipErrorInvalidValue;        // h3 needs the per-channel maxima of A1 (C words) from A1's producer
    if (C != ET_C || nh < 1 || nh > ET_MAXH || !aligned16(w3) || (parts != 1 && parts != 2 && parts != 3) || !A1 || !heads || !Wh ||       // (H == NULL: inference-mode forward, heads only)
        !bh || ((bits_h || bits_a) && (act != ACT_LRELU || !aligned16(bits_h) || !aligned16(bits_a))) ||
        !et_ld_ok(N, lda, ldh, ldo))
        return (int)hipErrorInvalidValue;
    const int Rpad = x6_round_up(ET_C, DX6_ROWS);
    const size_t lds = (size_t)parts * 16 * ET_C * 16;
    const H3Scale hs = parts == 2 ? H3Scale{h3_trailer(w3, ET_C, ET_C), amax_a1, 1, 0, 0, 0, 0} : H3_NONE;
    hipError_t e;
    if (parts == 3) {
        e = allow_big_lds(enc_tail_fwd_x6_kernel<3>, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_fwd_x6_kernel<3>), dim3(et_grid(N)), dim3(ET_THREADS), lds, S(stream), (const uint4*)w3,
                           Rpad, A1, lda, b2, Wh, bh, nh, H, ldh, heads, ldo, N, act, slope, (uint4*)bits_h, (uint4*)bits_a, hs);
    } else if (parts == 2) {
        e = allow_big_lds(enc_tail_fwd_x6_kernel<2>, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_fwd_x6_kernel<2>), dim3(et_grid(N)), dim3(ET_THREADS), lds, S(stream), (const uint4*)w3,
                           Rpad, A1, lda, b2, Wh, bh, nh, H, ldh, heads, ldo, N, act, slope, (uint4*)bits_h, (uint4*)bits_a, hs);
    } else {
        e = allow_big_lds(enc_tail_fwd_x6_kernel<1>, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_fwd_x6_kernel<1>), dim3(et_grid(N)), dim3(ET_THREADS), lds, S(stream), (const uint4*)w3,
                           Rpad, A1, lda, b2, Wh, bh, nh, H, ldh, heads, ldo, N, act, slope, (uint4*)bits_h, (uint4*)bits_a, hs);
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

int tvae_enc_tail_dgrad_x6(const void* w3p, const void* wh3, const float* dheads, long ldd, int nh, const void* bits_h,
                           const void* bits_a, float* dA1, long lda, int C, long N, float slope, int parts,
                           tvae_stream_t stream) {
    if (N <= 0) return 0;
    if (C != ET_C || nh < 1 || nh > ET_MAXH || !aligned16(w3p) || !aligned16(wh3) || !aligned16(bits_h) || !aligned16(bits_a) ||
        (parts != 1 && parts != 2 && parts != 3) || !dheads || !bits_h || !bits_a || !dA1 || !et_ld_ok(N, ldd, lda, 0))
        return (int)hipErrorInvalidValue;
    const int Rpad = x6_round_up(ET_C, DX6_ROWS);
    // parts == 2 (h3): w3p = tvae_dense_split2h cells (two fp16 parts), wh3 = tvae_dense_split3 cells (the skinny GEMM stays exact)
    const size_t lds = ((size_t)parts * 16 + (parts == 2 ? 3 : parts) * 2) * ET_C * 16;
    const float* amax_a = parts == 2 ? h3_trailer(w3p, ET_C, ET_C) : nullptr;
    hipError_t e;
    if (parts == 3) {
        e = allow_big_lds(enc_tail_dgrad_x6_kernel<3>, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_dgrad_x6_kernel<3>), dim3(et_grid(N)), dim3(ET_THREADS), lds, S(stream),
                           (const uint4*)w3p, Rpad, (const uint4*)wh3, Rpad, dheads, ldd, nh, (const uint4*)bits_h,
                           (const uint4*)bits_a, dA1, lda, N, slope, amax_a);
    } else if (parts == 2) {
        e = allow_big_lds(enc_tail_dgrad_x6_kernel<2>, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_dgrad_x6_kernel<2>), dim3(et_grid(N)), dim3(ET_THREADS), lds, S(stream),
                           (const uint4*)w3p, Rpad, (const uint4*)wh3, Rpad, dheads, ldd, nh, (const uint4*)bits_h,
                           (const uint4*)bits_a, dA1, lda, N, slope, amax_a);
    } else {
        e = allow_big_lds(enc_tail_dgrad_x6_kernel<1>, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_dgrad_x6_kernel<1>), dim3(et_grid(N)), dim3(ET_THREADS), lds, S(stream),
                           (const uint4*)w3p, Rpad, (const uint4*)wh3, Rpad, dheads, ldd, nh, (const uint4*)bits_h,
                           (const uint4*)bits_a, dA1, lda, N, slope, amax_a);
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

static long ew_slab_floats(long N) {
    const long nchunks = N / EW_NC;
    const long g = nchunks < cu_count() ? nchunks : cu_count();
    return (g < 1 ? 1 : g) * (long)ET_C * ET_C;
}
long tvae_enc_tail_wgrad_x6_ws_floats(long N) { return ew_slab_floats(N) + 8; }     // + the operand maxima of the h3 arithmetic

int tvae_enc_tail_wgrad_x6(const float* A1, long lda, const float* dheads, long ldd, int nh, const void* bits_h,
                           const float* Wh, float* dW2, float* ws, long ws_floats, int C, long N, float slope, int parts,
                           const float* amax_a1, tvae_stream_t stream) {
    if (N <= 0) return 0;
    if (C != ET_C || nh < 1 || nh > ET_MAXH || N % EW_NC != 0 || !aligned16(A1) || !aligned16(dheads) || !aligned16(bits_h) ||
        lda % 4 != 0 || ldd % 4 != 0 || (parts != 1 && parts != 2 && parts != 3) || !A1 || !dheads || !bits_h || !Wh || !dW2 ||
        !ws || ws_floats < tvae_enc_tail_wgrad_x6_ws_floats(N) || (parts == 2 && !amax_a1))
        return (int)hipErrorInvalidValue;
    const long nchunks = N / EW_NC;
    const int grid = (int)(nchunks < cu_count() ? nchunks : cu_count());
    float* slots = ws + ew_slab_floats(N);
    hipError_t e;
    if (parts == 3) {
        e = allow_big_lds(enc_tail_wgrad_x6_kernel<3>, EW_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_wgrad_x6_kernel<3>), dim3(grid), dim3(ET_THREADS), EW_LDS, S(stream), A1, lda, dheads, ldd,
                           nh, (const uint4*)bits_h, Wh, ws, N, slope, amax_a1, (const float*)slots);
    } else if (parts == 2) {
        // max |dheads|: one pass over nh x N floats (62 MB at the bench shape)
        hipLaunchKernelGGL(h3_zero_slots_kernel, dim3(1), dim3(64), 0, S(stream), slots, 1);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(h3_absmax_rows_kernel, dim3(grid1d(N / 16 + 1, 256, 96), nh), dim3(256), 0, S(stream), dheads, ldd, N,
                           slots);
        TVAE_CHECK_LAUNCH();
        e = allow_big_lds(enc_tail_wgrad_x6_kernel<2>, EW_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_wgrad_x6_kernel<2>), dim3(grid), dim3(ET_THREADS), EW_LDS, S(stream), A1, lda, dheads, ldd,
                           nh, (const uint4*)bits_h, Wh, ws, N, slope, amax_a1, (const float*)slots);
    } else {
        e = allow_big_lds(enc_tail_wgrad_x6_kernel<1>, EW_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((enc_tail_wgrad_x6_kernel<1>), dim3(grid), dim3(ET_THREADS), EW_LDS, S(stream), A1, lda, dheads, ldd,
                           nh, (const uint4*)bits_h, Wh, ws, N, slope, amax_a1, (const float*)slots);
    }
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(enc_tail_wgrad_total_kernel, dim3(ET_C * ET_C / 64), dim3(256), 0, S(stream), (const float*)ws, grid, dW2);
    TVAE_CHECK_LAUNCH();
    return 0;
}

/* Round 6: dW[r][c] = sum_n D[r][n] A[c][n] for two STORED operands (encoder tail with many head rows: dW2 = dH A1^T and
 * dWh = dheads H^T).  dW is [128][128]; rows >= rows_d are unspecified. */
int tvae_enc_tail_wgrad_wide(const float* D, long ldd, int rows_d, const float* A, long lda, float* dW, float* ws, long ws_floats,
                             int C, long N, const float* amax_d, const float* amax_a, tvae_stream_t stream) {
    if (N <= 0) return 0;
    if (C != ET_C || rows_d < 1 || rows_d > ET_C || N % EW_NC != 0 || !D || !A || !dW || !ws || !amax_d || !amax_a ||
        !aligned16(D) || !aligned16(A) || lda % 4 != 0 || ldd % 4 != 0 || lda < N || ldd < N ||
        ws_floats < tvae_enc_tail_wgrad_x6_ws_floats(N))
        return (int)hipErrorInvalidValue;
    const long nchunks = N / EW_NC;
    const int grid = (int)(nchunks < cu_count() ? nchunks : cu_count());
    hipError_t e = allow_big_lds(enc_tail_wgrad_plain_kernel, PW_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(enc_tail_wgrad_plain_kernel, dim3(grid), dim3(ET_THREADS), PW_LDS, S(stream), D, ldd, rows_d, A, lda, ws, N,
                       amax_d, amax_a);
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(enc_tail_wgrad_total_kernel, dim3(ET_C * ET_C / 64), dim3(256), 0, S(stream), (const float*)ws, grid, dW);
    TVAE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
