// libtvae_hip.so: the encoder tail with 8 .. 128 head rows (z_dim up to 62; the galaxy configuration's 103 rows) as two
// chained split-pipe GEMMs per 32-column chunk, forward and data gradient (enc_tail_wide_kernels.hpp).
#include "abi_dense_x6.hpp"
#include "enc_tail_wide_kernels.hpp"

using namespace tvae;

namespace {
int cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}
int et_grid(long N) {      // persistent: one workgroup per CU, never more workgroups than there are chunks for their eight waves
    const long chunks = (N + ET_CHUNK - 1) / ET_CHUNK;
    const long want = (chunks + ET_THREADS / 64 - 1) / (ET_THREADS / 64);
    const int cus = cu_count();
    return (int)(want < cus ? want : cus);
}
constexpr long ET_MAX_LD = 1L << 25;   // lane offsets are 32-bit BYTE offsets of up to 32 rows
inline bool ld_ok(long a) { return a < ET_MAX_LD; }

template <int NP, bool DG, int ACT>
int launch(const EtWide& a, hipStream_t st) {
    const size_t lds = (size_t)2 * NP * 16 * ET_C * 16;
    hipError_t e = allow_big_lds(enc_tail_wide_kernel<NP, DG, ACT>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((enc_tail_wide_kernel<NP, DG, ACT>), dim3(et_grid(a.N)), dim3(ET_THREADS), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" {

int tvae_enc_tail_wide_max_rows(void) { return ET_C; }

int tvae_enc_tail_fwd_wide(const void* w3, const void* wh3, const float* A1, long lda, const float* b2, const float* bh, int nh,
                           float* H, long ldh, float* heads, long ldo, void* bits_h, void* bits_a, int C, long N, int act,
                           float slope, int parts, const float* amax_a1, tvae_stream_t stream) {
    if (N <= 0) return 0;
    if (C != ET_C || nh < 1 || nh > ET_C || (parts != 1 && parts != 2) || !aligned16(w3) || !aligned16(wh3) || !A1 || !heads ||
        !bh || (parts == 2 && !amax_a1) || (!bits_h != !bits_a) || (bits_h && (act != ACT_LRELU || !aligned16(bits_h) || !aligned16(bits_a))) ||
        !ld_ok(N) || !ld_ok(lda) || !ld_ok(ldh) || !ld_ok(ldo))
        return (int)hipErrorInvalidValue;
    EtWide a;
    a.Wa3 = (const uint4*)w3; a.RpadA = x6_round_up(ET_C, DX6_ROWS); a.K8a = dense_k8pad(ET_C);
    a.Wb3 = (const uint4*)wh3; a.RpadB = x6_round_up(nh, DX6_ROWS);
    a.X = A1; a.ldx = lda; a.kx = ET_C;
    a.b1 = b2; a.b2 = bh;
    a.Y1 = H; a.ld1 = ldh; a.Y2 = heads; a.ld2 = ldo; a.m2 = nh;
    a.bitsH = (uint4*)bits_h; a.bitsA = (uint4*)bits_a;
    a.N = N; a.slope = slope;
    a.amax_wa = parts == 2 ? h3_trailer(w3, ET_C, ET_C) : nullptr;
    a.amax_wb = parts == 2 ? h3_trailer(wh3, nh, ET_C) : nullptr;
    a.amax_x = amax_a1; a.nx = ET_C;
    const hipStream_t st = S(stream);
    if (parts == 2) {
        if (act == ACT_LRELU) return launch<2, false, ACT_LRELU>(a, st);
        if (act == ACT_TANH) return launch<2, false, ACT_TANH>(a, st);
        return launch<2, false, ACT_NONE>(a, st);
    }
    if (act == ACT_LRELU) return launch<1, false, ACT_LRELU>(a, st);
    if (act == ACT_TANH) return launch<1, false, ACT_TANH>(a, st);
    return launch<1, false, ACT_NONE>(a, st);
}

int tvae_enc_tail_dgrad_wide(const void* wht3, const void* w3p, const float* dheads, long ldd, int nh, const void* bits_h,
                             const void* bits_a, float* dH, long ldh, float* dA1, long lda, int C, long N, float slope,
                             int parts, const float* amax_dheads, tvae_stream_t stream) {
    if (N <= 0) return 0;
    if (C != ET_C || nh < 1 || nh > ET_C || (parts != 1 && parts != 2) || !aligned16(wht3) || !aligned16(w3p) || !dheads || !dA1 ||
        !bits_h || !bits_a || !aligned16(bits_h) || !aligned16(bits_a) || (parts == 2 && !amax_dheads) || !ld_ok(N) ||
        !ld_ok(ldd) || !ld_ok(ldh) || !ld_ok(lda))
        return (int)hipErrorInvalidValue;
    EtWide a;
    a.Wa3 = (const uint4*)wht3; a.RpadA = x6_round_up(ET_C, DX6_ROWS); a.K8a = dense_k8pad(nh);
    a.Wb3 = (const uint4*)w3p; a.RpadB = x6_round_up(ET_C, DX6_ROWS);
    a.X = dheads; a.ldx = ldd; a.kx = nh;
    a.b1 = nullptr; a.b2 = nullptr;
    a.Y1 = dH; a.ld1 = ldh; a.Y2 = dA1; a.ld2 = lda; a.m2 = ET_C;
    a.bitsH = (uint4*)const_cast<void*>(bits_h); a.bitsA = (uint4*)const_cast<void*>(bits_a);
    a.N = N; a.slope = slope;
    a.amax_wa = parts == 2 ? h3_trailer(wht3, ET_C, nh) : nullptr;
    a.amax_wb = parts == 2 ? h3_trailer(w3p, ET_C, ET_C) : nullptr;
    a.amax_x = amax_dheads; a.nx = 1;
    return parts == 2 ? launch<2, true, ACT_LRELU>(a, S(stream)) : launch<1, true, ACT_LRELU>(a, S(stream));
}

}  // extern "C"
