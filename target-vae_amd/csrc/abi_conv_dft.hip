// libtvae_hip.so, lifting convolution through the frequency domain (conv_dft_kernels.hpp): image / bank
// spectra, the spectral contraction as batched launches of the split-pipe dense GEMMs (units 5 and 6), the transforms
// along w on the fp32 matrix pipe, inverse transform of the weight gradient.
#include "abi_dense_x6.hpp"
#include "conv_dft_kernels.hpp"
#include "conv_dft_h3_kernels.hpp"

using namespace tvae;

// Zero fill as an ordinary kernel on the caller's stream.  hipMemsetAsync is NOT used: under GPU sharing (two processes
// time-slicing the device) its fill was observed to land late relative to the kernels queued behind it on the same
// stream -- rows of the spectral weight zeroed after dft_spectra had written them, and zeros appearing in blocks the
// caching allocator had already handed to other tensors (profiles/tools/stress_determinism.py; profiles/README.md).
static __global__ void dft_zero_kernel(float4* __restrict__ p, long n4, float* __restrict__ tail, int ntail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
static hipError_t dft_zero(float* p, long floats, hipStream_t st) {     // p is 16-byte aligned
    const long n4 = floats / 4;
    const int ntail = (int)(floats - 4 * n4);
    long blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dft_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(p), n4, p + 4 * n4,
                       ntail);
    return hipGetLastError();
}

static int dev_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

// Geometry / workspace of the frequency-domain lifting convolution (conv_dft_kernels.hpp).
struct DftPlan {
    int L, Lh, Ho, M, K2, Cin; // frame, half spectrum, output size, rows C*R, reduction 2*Ky*Cin
    bool mixed;                // frequency domain along x only, spatial taps along y (Ky = ksz); else both axes (Ky = L)
    int Ky;
    bool gen;                  // generic transforms along w (Lh > 64 or Ho > 64: no specialised instance)
    int ring;                  // ring (LDS-DMA) transforms along w: 0 = none, 1 / 2 / 3 = the 44- / 96- / 66-wide frame
    int LHP, NT, REM1;         // forward w-transform: frequencies processed, 32-row output tiles, extra row
    int NS, NRT;               // backward w-transform: k2-steps (pairs of w), 32-row tiles of (fx, ri)
    int FXB, nblk;             // spectra: frequencies per workgroup, blocks per plane
    int FXBf, nblkf;           // ... of the FILTER planes (mixed form: the images may take more, smaller blocks)
    int FXBd;                  // inverse transform of the weight gradient: frequencies per pass
    int splits;                // reduction slices of the weight-gradient GEMM
    int wsplits;               // the same for its exact-fit tile (dense_wgrad_x6_wide_kernel), 0: geometry not eligible
    int Mb;                    // rows per fx in the stacked spectral weight (2M rounded up to the 512-row tile)
    long NB, NBpad;            // (image, output row) columns
    long at_floats;            // A^T [Lh][K2][NBpad]
    long w_floats;             // W   [Lh][Mb][K2]
    long w3_floats;            // split cells of W
    long t_floats;             // T / S' [Lh][2M][NBpad]
    long eo_floats, ed_floats, tab_floats;
    // round 6: the transforms along w of LARGE frames on the 16-bit matrix pipe (conv_dft_h3_kernels.hpp; h3 arithmetic only):
    // k-steps / output tiles of the two GEMMs, their cell tables behind the fp32 tables, 0 = not this geometry
    int h3w, KS, NWT, WS, NKT;
    long eoc_floats, edc_floats;
    long g_floats;             // one split-K slab of G [Lh][2M][K2]
    size_t lds_sp, lds_db;
    // operand maxima of the h3 arithmetic, behind A^T (offsets in floats from the 16-byte aligned end of A^T):
    //   cmax [Lh][B]  per (frequency, image) bound of A^T's columns     fmax [Lh]  per frequency bound of A^T
    //   wmax [Lh][Mb] per stacked row of the spectral weight            smax [M]   per filter row of S'
    //   a1max [C]     per channel max |out| (LAST C words of the buffer: tvae_enc_tail_*_x6 take a pointer to them)
    long o_cmax, o_fmax, o_wmax, o_smax, o_a1max, trailer_floats;
    bool ok;
};
static size_t dft_lds_spectra(int S, int L, int FXB, bool mixed = false) {
    return (size_t)S * FXB * 8 + (mixed ? 0 : (size_t)L * FXB * 8) + (size_t)L * 8;
}
constexpr size_t DFT_LDS_TARGET = 50 * 1024;       // three workgroups per CU for the latency-bound direct-sum transforms
static DftPlan dft_plan(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    DftPlan q;
    q.Cin = Cin;
    // Frame of the circular correlation (round 5).  The reference pads the image with `pad` zeros on BOTH sides (frame n + 2 pad)
    // and the correlation theorem was first applied on that frame.  A circular frame of only L = n + pad is already alias free:
    // with the image at positions [pad, pad + n) of the frame, an index h + u (output row + tap) that runs past L lands on
    // positions [0, pad) -- the LEADING zeros -- exactly where the linear correlation meets the TRAILING zeros of the padded
    // image (h + u <= n + 2 pad - 1 < L + pad; one wrap only).  The two zero bands share their storage.  The filter (ksz taps)
    // must fit the frame as well: L = max(n + pad, ksz), rounded up to a multiple of 4 (radix-4 step of the spectra, whole
    // octets of the spectral reduction) but never beyond the reference's own n + 2 pad.  64 x 64 / k 64 / p 16: L = 80 instead of
    // 96 -- Lh 41 instead of 49 frequencies (T and S' 16 % smaller), reduction 2 L = 160 instead of 192 (30 % fewer matrix
    // FLOPs in both spectral GEMMs); 28 x 28 / k 28 / p 8: 36 instead of 44; the galaxy shape 160 instead of 192.  Exact: nothing
    // is approximated (tests/test_hip_primitives.py::test_conv1_dft_matches_fp64).  TVAE_DFT_FULL_FRAME=1 restores n + 2 pad.
    q.Ho = n + 2 * pad - ksz + 1;
    {
        static const bool full = getenv("TVAE_DFT_FULL_FRAME") && getenv("TVAE_DFT_FULL_FRAME")[0] == '1';
        int Lmin = n + pad > ksz ? n + pad : ksz;
        Lmin = (Lmin + 3) & ~3;
        q.L = (full || Lmin > n + 2 * pad) ? n + 2 * pad : Lmin;
    }
    q.Lh = q.L / 2 + 1;
    q.M = C * R;
    // Round 5: the mixed form (conv_dft_kernels.hpp: dft_spectra_x_kernel) -- the DFT runs along x only and the reduction of the
    // spectral GEMMs is (channel, re | im, tap row u < ksz) instead of (channel, re | im, frequency fy < L): ksz <= L, so it is
    // never the longer one (128 instead of 160 at the 64 x 64 shape, 384 instead of 960 at the galaxy shape).
    // TVAE_DFT_YSPECTRAL=1 restores the transform along both axes.
    {
        static const bool yspec = getenv("TVAE_DFT_YSPECTRAL") && getenv("TVAE_DFT_YSPECTRAL")[0] == '1';
        q.mixed = !yspec;
    }
    q.Ky = q.mixed ? ksz : q.L;
    q.K2 = 2 * q.Ky * Cin;
    q.NB = (long)B * q.Ho;
    q.NBpad = (q.NB + 127) / 128 * 128;
    q.Mb = x6_round_up(2 * q.M, DX6_ROWS);
    q.at_floats = (long)q.Lh * q.K2 * q.NBpad;
    q.w_floats = (long)q.Lh * q.Mb * q.K2;
    q.w3_floats = dense_x6_bytes(q.Lh * q.Mb, q.K2) / 4;
    q.t_floats = (long)q.Lh * 2 * q.M * q.NBpad;
    q.gen = q.Lh > 64 || q.Ho > DFT_WROWS;
    // frames of the reference configurations (28x28 k28 p8; 64x64 k64 p16; the 50x50 MNIST-U geometry k28 p8) take the
    // ring kernels (conv_dft_kernels.hpp); every other frame the register-staged / generic ones
    // ring ids: 1 / 2 / 3 = the 28x28, 64x64 and 50x50 (MNIST-U) geometries on the short frame (L = 36 / 80 / 60);
    //           4 / 5 / 6 = the same on the reference's full frame (L = 44 / 96 / 66; TVAE_DFT_FULL_FRAME=1)
    q.ring = (q.L == 36 && q.Ho == 17) ? 1 : (q.L == 80 && q.Ho == 33) ? 2 : (q.L == 60 && q.Ho == 39) ? 3 :
             (q.L == 44 && q.Ho == 17) ? 4 : (q.L == 96 && q.Ho == 33) ? 5 : (q.L == 66 && q.Ho == 39) ? 6 : 0;
    if (q.ring) {              // (LHP, NT, REM1) / (NS, NRT) of the instances: tables sized to match
        q.LHP = q.Lh; q.NT = q.Ho > 32 + 1 ? 2 : 1; q.REM1 = q.Ho == 33 ? 1 : 0;
        q.NS = (q.Ho + 1) / 2;
        q.NRT = (q.ring == 5) ? 3 : (2 * q.Lh + 31) / 32;      // (the 96-wide frame: 96 real rows + the Nyquist pair on the vector ALU)
    } else if (q.gen) {                                               // whole 32-row tiles, zero rows beyond Ho
        q.LHP = q.Lh; q.NT = (q.Ho + 31) / 32; q.REM1 = 0;
        q.NS = (q.Ho + 1) / 2; q.NRT = (2 * q.Lh + 31) / 32;
    } else {
        q.LHP = q.Lh == 23 ? 23 : (q.Lh == 49 ? 49 : 64);      // exact instances of the two reference frames, else generic
        q.NT = q.Ho <= 33 ? 1 : 2;
        q.REM1 = q.Ho == 33 ? 1 : 0;
        if (q.Ho <= 18 && q.Lh <= 32) { q.NS = 9; q.NRT = 2; }
        else if (q.Ho <= 34) { q.NS = 17; q.NRT = 4; }
        else { q.NS = 32; q.NRT = 4; }
    }
    q.eo_floats = (long)q.LHP * (q.NT + q.REM1) * 64;
    q.ed_floats = (long)q.NS * q.NRT * 64;
    // the h3 instances of the wide transforms: the galaxy frame (L = 160: Lh = 81, Ho = 129) and every other generic frame whose
    // step / tile counts they cover
    q.KS = (2 * q.Lh + 15) / 16; q.NWT = (q.Ho + 31) / 32; q.WS = (q.Ho + 15) / 16; q.NKT = (2 * q.Lh + 31) / 32;
    q.h3w = (q.gen && q.KS <= 11 && q.NWT <= 5 && q.WS <= 9 && q.NKT <= 6 && q.Ho >= 32) ? 1 : 0;
    q.eoc_floats = q.h3w ? (long)5 * 11 * 2 * 64 * 4 : 0;
    q.edc_floats = q.h3w ? (long)6 * 9 * 2 * 64 * 4 : 0;
    // EO + ED + per-row bias-gradient sums (+ the h3 cell tables, 16-byte aligned)
    q.tab_floats = ((q.eo_floats + 3) & ~3L) + ((q.ed_floats + 3) & ~3L) + ((q.M + 3) & ~3L) + q.eoc_floats + q.edc_floats;
    q.g_floats = (long)q.Lh * 2 * q.M * q.K2;
    // 8 reduction slices deal one to each XCD (TileMap: a slice lives on ONE XCD).  Round 4: also for short reductions --
    // with NBpad / 1024 slices a 32-image step (the per-GPU share of a 256-image global batch on 8 GPUs) ran its 392 tiles on
    // ONE slice = one XCD = 32 of the 256 CUs: 0.81 ms for an eighth of the work the full batch does in 1.56 ms.  Now every
    // slice of >= 2 k-steps gets its own XCD (0.81 -> 0.23 ms at B = 32; B = 256 unchanged: 8 slices either way).
    // (Problems with >= 8 column tiles -- several input channels, wide frames: K2 = 1 152 at the galaxy shape -- already
    // spread over the XCDs by their column tiles and keep few slices: a slab there is as large as S' itself.)
    q.splits = (int)(cdiv(q.K2, 128) >= 8 ? q.NBpad / 1024 : q.NBpad / 32);
    if (q.splits < 1) q.splits = 1;
    if (q.splits > 8) q.splits = 8;
    // exact-fit tile where the 2 L Cin columns are one and a half of the 128-wide tiles.  (Its slices are not pinned to XCDs, so
    // their number is free: 9 slices, whose groups fill 7 whole rounds of the 256 CUs where 8 run 6.125, measured the same.)
    // Wider problems take it in 192-column tiles when those pad no more than the 128-wide ones (1 152 columns at the galaxy
    // shape: 6 tiles instead of 9, i.e. S' read and split 6 times instead of 9).
    q.wsplits = ((2 * q.M) % WW_ROWS == 0 && q.K2 > 128 &&
                 (q.K2 <= 192 || cdiv(q.K2, 192) * 192 <= cdiv(q.K2, 128) * 128)) ? q.splits : 0;
    // Round 5: its slices are free in number, and every slice writes a whole slab of G that the inverse transform reads back.
    // Where a slab is as large as the operand (galaxy shape: 8 images, slab 0.51 GB against S' 1.53 GB) eight slices moved
    // 2 x 4 GB to read 1.5: as many slices as keep the slab traffic below the operand's, at least one, at most the 8 above.
    if (q.wsplits > 0) {
        const long by_bytes = q.t_floats / (2 * q.g_floats);
        const int cap = (int)(by_bytes < 1 ? 1 : (by_bytes > 8 ? 8 : by_bytes));
        if (q.wsplits > cap) q.wsplits = cap;
    }
    // spectra: as few frequency blocks per plane as LDS allows (images and filters share one launch)
    const int S = n > ksz ? n : ksz;
    q.nblk = 1;
    q.FXB = q.Lh;
    while (q.nblk < 16 && dft_lds_spectra(S, q.L, q.FXB, q.mixed) > DFT_LDS_TARGET) {
        ++q.nblk;
        q.FXB = (q.Lh + q.nblk - 1) / q.nblk;
    }
    // mixed form: an image's workgroup writes FXB * ksz * Ho operand values as 132-byte runs 33 KB apart; with all Lh frequencies
    // in one block that loop, not the transform, sets the kernel's duration (0.11 ms at the 64 x 64 shape, at ANY batch size: one
    // workgroup per image is the longest pole); more, smaller frequency blocks per plane spread it over more workgroups
    // (measured at the 64 x 64 shape: one block 114 us at any batch; four blocks 130 us at 256 images -- more workgroups than the
    //  write stream needs -- but 69 us at 32: as many blocks as give the IMAGES a workgroup per CU, at most 8)
    q.nblkf = q.nblk;
    q.FXBf = q.FXB;
    if (q.mixed) {
        int want = (256 + B * Cin - 1) / (B * Cin);
        if (want > 8) want = 8;
        while (q.nblk < want && q.nblk < q.Lh) {
            ++q.nblk;
            q.FXB = (q.Lh + q.nblk - 1) / q.nblk;
        }
    }
    q.lds_sp = dft_lds_spectra(S, q.L, q.FXBf, q.mixed);       // (the larger of the two block sizes: FXBf >= FXB)
    q.FXBd = q.Lh;
    auto lds_db = [&](int f) { return (size_t)q.L * f * 8 + (size_t)ksz * f * 8 + (size_t)q.L * 8 + (size_t)ksz * ksz * 4; };
    // the inverse transform runs one workgroup per (filter, channel): 1 024 of them at the bench shape = FOUR per CU, so its
    // LDS is sized for four residents (<= 39 KB) -- with three, a quarter of the workgroups ran in a second, mostly empty round
    constexpr size_t DFT_LDS_DBANK = 39 * 1024;
    for (int nb = 1; nb < 16 && lds_db(q.FXBd) > DFT_LDS_DBANK; ++nb) q.FXBd = (q.Lh + nb) / (nb + 1);
    q.lds_db = lds_db(q.FXBd);
    if (q.mixed) q.lds_db = ((size_t)ksz * q.Lh + q.L) * 8;      // dft_dbank_x_kernel: D [ksz][Lh] complex + twiddles
    q.o_cmax = 0;
    q.o_fmax = q.o_cmax + (long)q.Lh * B;
    q.o_wmax = q.o_fmax + q.Lh;
    q.o_smax = q.o_wmax + (long)q.Lh * q.Mb;
    q.o_a1max = (q.o_smax + q.M + 3) & ~3L;
    q.trailer_floats = q.o_a1max + C;
    q.ok = Cin >= 1 && q.Ho >= 1 && q.NT <= 5 && q.FXB <= 256 && q.lds_sp <= 152 * 1024 && q.lds_db <= 152 * 1024 &&
           (long)q.Lh * q.Mb < 2000000000L && (size_t)4 * 32 * ((2 * q.NS) | 1) * 4 <= 150 * 1024 &&
           (long)B * Cin * q.nblk + (long)q.M * Cin * q.nblk < 2000000000L && (long)q.M * (q.NBpad / 32) < 2000000000L;
    return q;
}

extern "C" {

// ---- lifting convolution through the frequency domain (conv_dft_kernels.hpp) -------------------------------------
int tvae_conv1_dft_supported(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    return dft_plan(B, Cin, n, ksz, pad, C, R).ok ? 1 : 0;
}
// A^T is followed by the operand maxima of the h3 arithmetic (DftPlan: cmax, fmax, wmax, smax, a1max), written by the kernels
// that produce those operands and read by the GEMMs that split them; the LAST C floats of the buffer are the per-channel
// maxima of the convolution's output (the streamed operand of the encoder tail that follows)
long tvae_conv1_dft_at_floats(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    return ((q.at_floats + 3) & ~3L) + q.trailer_floats;
}
// frame length L of the circular correlation, and which instance of the transforms along w a geometry takes (0: register-staged
// / generic; 1 .. 6: the LDS-DMA ring kernels, dft_plan)
int tvae_conv1_dft_frame(int B, int Cin, int n, int ksz, int pad, int C, int R) { return dft_plan(B, Cin, n, ksz, pad, C, R).L; }
int tvae_conv1_dft_ring(int B, int Cin, int n, int ksz, int pad, int C, int R) { return dft_plan(B, Cin, n, ksz, pad, C, R).ring; }
long tvae_conv1_dft_ws_floats(int B, int Cin, int n, int ksz, int pad, int C, int R) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    // forward: W + W3 + T + tables; backward: S' (= T) + split-K slabs of G + tables
    const long fwd = q.w_floats + q.w3_floats + q.t_floats + q.tab_floats + 64;
    const long bwd = q.t_floats + (q.splits > q.wsplits ? q.splits : q.wsplits) * q.g_floats + q.tab_floats + 64;
    return fwd > bwd ? fwd : bwd;
}

int tvae_conv1_fwd_dft(const float* y, const float* bank, const float* bias, float* out, float* at, float* ws,
                       long ws_floats, int B, int Cin, int n, int ksz, int pad, int C, int R, int act, float slope,
                       int parts, tvae_stream_t stream) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    if (!q.ok || ws_floats < tvae_conv1_dft_ws_floats(B, Cin, n, ksz, pad, C, R) || !aligned16(ws) || !aligned16(at))
        return (int)hipErrorInvalidValue;
    hipStream_t st = S(stream);
    float* W = ws;
    float* W3 = W + ((q.w_floats + 3) & ~3L);
    float* T = W3 + ((q.w3_floats + 3) & ~3L);
    float* tab = T + ((q.t_floats + 3) & ~3L);
    if (q.NBpad != q.NB) {
        hipError_t e = dft_zero(at, q.at_floats, st);
        if (e != hipSuccess) return (int)e;
    }
    // h3 arithmetic (two fp16 parts, three products): both forms of the spectral GEMM (four-wave tile for reductions <= 256,
    // eight-wave tile beyond: several channels, wide frames) take the operand maxima dft_spectra leaves behind A^T
    const bool h3 = parts == 2;
    // bf16 STORAGE of T (round 4): the one-part throughput mode on a ring geometry with whole 256-row tiles writes and reads T
    // as 2-byte elements (TVAE_BF16_STORE=0 keeps fp32 storage; the fp32-class arithmetics never take this path)
    constexpr bool bf16_store = true;
    // (ring 3, the 50 x 50 geometry on its 60-wide frame: a bf16 slot of 4 KB cannot hold the 32 x 39 transposition patch)
    const bool t16 = bf16_store && parts == 1 && q.ring && q.ring != 3 && q.K2 <= 256 && (2 * q.M) % DX4_ROWS == 0;
    // the maxima are produced in every arithmetic (one small fill and a few atomics): the weight gradient may run in
    // h3 after a forward that did not
    float* amax = at + ((q.at_floats + 3) & ~3L);
    {
        hipError_t ez = dft_zero(amax, q.trailer_floats, st);
        if (ez != hipSuccess) return (int)ez;
    }
    const DftMax mxp{amax + q.o_cmax, amax + q.o_fmax, amax + q.o_wmax};
    float* a1max = amax + q.o_a1max;
    float* EO = tab;
    float* ED = EO + ((q.eo_floats + 3) & ~3L);
    hipError_t e = q.mixed ? allow_big_lds(dft_spectra_x_kernel, q.lds_sp) : allow_big_lds(dft_spectra_kernel, q.lds_sp);
    if (e != hipSuccess) return (int)e;
    if (q.Mb != 2 * q.M) {                             // rows that pad 2M to the 512-row tile must be zero
        e = dft_zero(W, q.w_floats, st);
        if (e != hipSuccess) return (int)e;
    }
    constexpr bool spectra_mf = true;
    const int smax_ = n > ksz ? n : ksz;
    if (q.mixed && spectra_mf && smax_ <= 128) {         // row transform on the fp32 matrix pipe (dft_spectra_x_mf_kernel)
#define TVAE_SPX_MF(KSR_)                                                                                            \
    do {                                                                                                            \
        e = allow_big_lds(dft_spectra_x_mf_kernel<KSR_>, q.lds_sp);                                                 \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL(dft_spectra_x_mf_kernel<KSR_>, dim3((unsigned)(B * Cin * q.nblk + q.M * Cin * q.nblkf)), dim3(256), \
                           q.lds_sp, st, y, at, B, Cin, n, pad, q.Ho, q.NBpad, bank, W, ksz, q.M, q.Mb, q.L, q.Lh, q.FXB, q.nblk, \
                           q.FXBf, q.nblkf, mxp);                                                                   \
    } while (0)
        if (smax_ <= 32) TVAE_SPX_MF(16); else if (smax_ <= 64) TVAE_SPX_MF(32); else TVAE_SPX_MF(64);
#undef TVAE_SPX_MF
    } else if (q.mixed)
        hipLaunchKernelGGL(dft_spectra_x_kernel, dim3((unsigned)(B * Cin * q.nblk + q.M * Cin * q.nblkf)), dim3(256), q.lds_sp, st,
                           y, at, B, Cin, n, pad, q.Ho, q.NBpad, bank, W, ksz, q.M, q.Mb, q.L, q.Lh, q.FXB, q.nblk, q.FXBf, q.nblkf,
                           mxp);
    else
    hipLaunchKernelGGL(dft_spectra_kernel, dim3((unsigned)((B + q.M) * Cin * q.nblk)), dim3(256), q.lds_sp, st, y, at, B, Cin,
                       n, pad, q.Ho, q.NBpad, bank, W, ksz, q.M, q.Mb, q.L, q.Lh, q.FXB, q.nblk, mxp);
    TVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(dft_fmax_kernel, dim3(q.Lh), dim3(256), 0, st, (const float*)mxp.cmax, mxp.fmax, B);
    TVAE_CHECK_LAUNCH();
    // split the stacked spectral weights [Lh*2M rows][2L] into cells, then ONE batched launch of the split dense GEMM
    const int rows = q.Lh * q.Mb;
    int rc = 0;
    if (h3) {
        const int Rp = x6_round_up(rows, DX6_ROWS), K8p = dense_k8pad(q.K2);
        if (K8p <= 48)       // both sides coalesced through LDS (<= 48 KB)
            hipLaunchKernelGGL(dense_split2h_rows_kernel, dim3(Rp / 32), dim3(256), (size_t)32 * K8p * 32, st, (const float*)W,
                               (long)q.K2, (uint4*)W3, rows, Rp, q.K2, K8p, (const float*)nullptr, (const float*)mxp.wmax);
        else
            hipLaunchKernelGGL(dense_split2h_kernel, dim3(grid1d((long)K8p * Rp, 256)), dim3(256), 0, st, (const float*)W,
                               (long)q.K2, (uint4*)W3, rows, Rp, q.K2, K8p, 0, (const float*)nullptr, (const float*)mxp.wmax);
        TVAE_CHECK_LAUNCH();
    } else {
        rc = tvae_dense_split3(W, q.K2, W3, q.w3_floats * 4, rows, q.K2, 0, nullptr, nullptr, stream);
    }
    if (rc) return rc;
    {
        Epilogue ep;
        ep.C = T; ep.ldc = (long)q.Lh * 128;              // T is [n >> 7][m'][fx][n & 127] (dft_t_off)
        ep.ctile = (long)2 * q.M * q.Lh * 128;
        const int Rpad = x6_round_up(rows, DX6_ROWS);
        // short reductions (K2 <= 256: twelve k-steps at the 64 x 64 shape): 256-row tiles, two 4-wave workgroups per CU
        const int TR = q.K2 <= 256 ? DX4_ROWS : DX6_ROWS;
        TileMap tm{Rpad / TR, (int)(q.NBpad / 128), 1};
        tm.bt = q.Mb / TR;                             // group = (fx, quarter of the column tiles): 4*Lh groups over 8 XCDs
        tm.nch = 4;
        const DenseBatch bt{q.Mb / TR, (long)q.K2 * q.NBpad, 128};
        // h3: one scale per stacked row (fx, m') of the spectral weight, one per (fx, image) of A^T's columns
        const H3Scale hs{mxp.wmax, mxp.cmax, 1, 0, 0, q.Ho, B};
        // the panel of A^T resident in LDS, weight cells streamed by free-running waves (dense_x6_xres_kernel)
        if (TR == DX4_ROWS && !t16 &&
            dense_x6_batched_xres(W3, at, q.NBpad, ep, 2 * q.M, q.Mb, q.Lh, (int)q.NBpad, q.K2, bt.x_stride, 128, parts, st, hs, &rc)) {
            if (rc) return rc;
        } else
        rc = TR == DX4_ROWS ? dense_x6_batched4(W3, at, q.NBpad, ep, 2 * q.M, rows, (int)q.NBpad, q.K2, tm, bt, parts, st, hs, t16)
                            : dense_x6_batched(W3, at, q.NBpad, ep, 2 * q.M, rows, (int)q.NBpad, q.K2, tm, bt, parts, st, hs);
        if (rc) return rc;
    }
    {
        // contraction over fx on the fp32 matrix pipe
        const int NTT = q.NT + q.REM1;
        hipLaunchKernelGGL(dft_wtab_kernel, dim3(64), dim3(256), 0, st, EO, ED, q.L, q.Lh, q.Ho, q.LHP, q.NT, NTT, q.NS,
                           q.NRT, q.mixed ? 1.f / (float)q.L : 1.f / ((float)q.L * (float)q.L));
        TVAE_CHECK_LAUNCH();
        const long ntiles = (long)q.M * (q.NBpad / 32);
        if (q.ring) {
            const long vt = (long)q.M * ((q.NB + 31) / 32);           // tiles with at least one real column
            const int cus = dev_cu_count();
            const int grid = (int)((vt + 3) / 4 < cus ? (vt + 3) / 4 : cus);
#define TVAE_OUT_RING_ONE(L_, N_, R_, H_, TH_, T16_)                                                                  \
    do {                                                                                                            \
        const size_t lds_r = (size_t)4 * 3 * ((T16_) ? (2 * L_ + 15) / 16 : (L_ + 3) / 4) * 1024;                   \
        e = allow_big_lds(dft_out_ring_kernel<L_, N_, R_, H_, TH_, T16_>, lds_r);                                   \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((dft_out_ring_kernel<L_, N_, R_, H_, TH_, T16_>), dim3(grid), dim3(256), lds_r, st,      \
                           (const float*)T, (const float*)EO, bias, out, q.M, R, B, q.Lh, act, slope, a1max);       \
    } while (0)
#define TVAE_OUT_RING(L_, N_, R_, H_)                                                                               \
    do {                                                                                                            \
        if (act == ACT_TANH) { if (t16) TVAE_OUT_RING_ONE(L_, N_, R_, H_, true, true); else TVAE_OUT_RING_ONE(L_, N_, R_, H_, true, false); } \
        else { if (t16) TVAE_OUT_RING_ONE(L_, N_, R_, H_, false, true); else TVAE_OUT_RING_ONE(L_, N_, R_, H_, false, false); } \
    } while (0)
            if (q.ring == 1) TVAE_OUT_RING(19, 1, false, 17);
            else if (q.ring == 2) TVAE_OUT_RING(41, 1, true, 33);
            else if (q.ring == 3) { if (act == ACT_TANH) TVAE_OUT_RING_ONE(31, 2, false, 39, true, false); else TVAE_OUT_RING_ONE(31, 2, false, 39, false, false); }
            else if (q.ring == 4) TVAE_OUT_RING(23, 1, false, 17);
            else if (q.ring == 5) TVAE_OUT_RING(49, 1, true, 33);
            else TVAE_OUT_RING(34, 2, false, 39);
#undef TVAE_OUT_RING
#undef TVAE_OUT_RING_ONE
            TVAE_CHECK_LAUNCH();
            return 0;
        }
        constexpr bool wide_gen = true, wide_h3 = true;
        if (q.h3w && h3 && wide_h3 && !t16) {
            // round 6: large frames in the h3 arithmetic leave the fp32 matrix pipe (which bounds them: conv_dft_h3_kernels.hpp)
            uint4* EOc = reinterpret_cast<uint4*>(ED + ((q.ed_floats + 3) & ~3L) + ((q.M + 3) & ~3L));
            uint4* EDc = EOc + q.eoc_floats / 4;
            const float norm = q.mixed ? 1.f / (float)q.L : 1.f / ((float)q.L * (float)q.L);
            hipLaunchKernelGGL(dft_wtab_h3_kernel, dim3(64), dim3(256), 0, st, EOc, EDc, q.L, q.Lh, q.Ho, 11, 5, 9, 6, norm);
            TVAE_CHECK_LAUNCH();
            int ex = 0;
            frexpf(2.f * norm, &ex);                     // 2 norm in [2^(ex-1), 2^ex): h3_scale = 2^(15 - ex)
            const float eo_inv = ldexpf(1.f, ex - 15);
            constexpr size_t lds_h = ((size_t)16 * 11 * 32 + 2 * 2 * 11 * 32 * 4 + 32 + 5 * 32 * 33) * 4;
            e = allow_big_lds(dft_out_h3_kernel<11, 5, 5>, lds_h);
            if (e != hipSuccess) return (int)e;
            const int cus = dev_cu_count();
            const int grid = (int)(ntiles < 2L * cus ? ntiles : 2L * cus);
            hipLaunchKernelGGL((dft_out_h3_kernel<11, 5, 5>), dim3(grid), dim3(320), lds_h, st, (const float*)T,
                               (const uint4*)EOc, bias, out, q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope, eo_inv, a1max);
            TVAE_CHECK_LAUNCH();
            return 0;
        }
        // large frames (galaxy shape): a workgroup per tile, its waves split the output rows (dft_out_wide_kernel); the single
        // last column of Ho = 32 k + 1 outputs goes to the vector ALU instead of a whole wave
        const int NTW = (q.Ho % 32 == 1 && q.NT > 1) ? q.NT - 1 : q.NT;
        if (q.gen && wide_gen && q.Lh <= DFT_WIDE_LH && NTW <= 8 && q.REM1 == 0 && q.Lh * 16 <= 8 * 64 * NTW) {
            const int LHR = (q.Lh <= 82 && cdiv(16 * q.Lh, 64 * NTW) <= 6) ? 82 : DFT_WIDE_LH;      // the instance's padded slot
            const size_t lds_w = ((size_t)2 * LHR * 64 + (size_t)NTW * 32 * 33 + 2 * q.Lh) * 4;
            if (lds_w <= 150 * 1024) {
                const int cus = dev_cu_count();
                const long fit = (long)(150 * 1024 / lds_w);
                const long wg_per_cu = fit < 2 ? fit : 2;
                const int grid = (int)(ntiles < wg_per_cu * cus ? ntiles : wg_per_cu * cus);
#define TVAE_OUT_WIDE(LHR_, NLD_)                                                                                    \
    do {                                                                                                            \
        e = allow_big_lds(dft_out_wide_kernel<LHR_, NLD_>, lds_w);                                                  \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((dft_out_wide_kernel<LHR_, NLD_>), dim3(grid), dim3(64 * NTW), lds_w, st, (const float*)T, \
                           (const float*)EO, bias, out, q.M, R, B, q.Ho, q.Lh, q.NBpad, q.NT, NTW, act, slope, a1max); \
    } while (0)
                const int nld = cdiv(16 * q.Lh, 64 * NTW);            // float4 pieces per thread and tile
                if (q.Lh <= 82 && nld <= 6) TVAE_OUT_WIDE(82, 6); else TVAE_OUT_WIDE(DFT_WIDE_LH, 8);
#undef TVAE_OUT_WIDE
                TVAE_CHECK_LAUNCH();
                return 0;
            }
        }
        if (q.gen) {
            const int grid = (int)((ntiles + 3) / 4 < 4096 ? (ntiles + 3) / 4 : 4096);
#define TVAE_OUT_GEN(N_)                                                                                            \
    hipLaunchKernelGGL(dft_out_gen_kernel<N_>, dim3(grid), dim3(256), 0, st, (const float*)T, (const float*)EO, bias, out, \
                       q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope, a1max)
            switch (q.NT) {
                case 1: TVAE_OUT_GEN(1); break;
                case 2: TVAE_OUT_GEN(2); break;
                case 3: TVAE_OUT_GEN(3); break;
                case 4: TVAE_OUT_GEN(4); break;
                default: TVAE_OUT_GEN(5); break;
            }
#undef TVAE_OUT_GEN
            TVAE_CHECK_LAUNCH();
            return 0;
        }
        const size_t lds_o = ((size_t)q.LHP * NTT * 64 + (size_t)4 * 32 * (q.Ho | 1)) * 4;
        const int grid = (int)((ntiles + 3) / 4 < 768 ? (ntiles + 3) / 4 : 768);
        const int iters = (int)((ntiles + 4L * grid - 1) / (4L * grid));
#define TVAE_OUT_MF(L_, N_, R_)                                                                                     \
    do {                                                                                                            \
        e = allow_big_lds(dft_out_mf_kernel<L_, N_, R_>, lds_o);                                                    \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((dft_out_mf_kernel<L_, N_, R_>), dim3(grid), dim3(256), lds_o, st, (const float*)T,      \
                           (const float*)EO, bias, out, q.M, R, B, q.Ho, q.Lh, q.NBpad, act, slope, iters, a1max); \
    } while (0)
#define TVAE_OUT_MF_L(L_)                                                                                           \
    do {                                                                                                            \
        if (q.REM1) TVAE_OUT_MF(L_, 1, true); else if (q.NT == 1) TVAE_OUT_MF(L_, 1, false); else TVAE_OUT_MF(L_, 2, false); \
    } while (0)
        if (q.LHP == 23) TVAE_OUT_MF_L(23); else if (q.LHP == 49) TVAE_OUT_MF_L(49); else TVAE_OUT_MF_L(64);
#undef TVAE_OUT_MF_L
#undef TVAE_OUT_MF
        TVAE_CHECK_LAUNCH();
    }
    return 0;
}

int tvae_conv1_wgrad_dft(const float* dpre, const float* at, float* dbank, float* dbias, float* ws, long ws_floats, int B,
                         int Cin, int n, int ksz, int pad, int C, int R, int parts, tvae_stream_t stream) {
    const DftPlan q = dft_plan(B, Cin, n, ksz, pad, C, R);
    if (!q.ok || ws_floats < tvae_conv1_dft_ws_floats(B, Cin, n, ksz, pad, C, R) || !aligned16(ws) || !aligned16(at))
        return (int)hipErrorInvalidValue;
    hipStream_t st = S(stream);
    // h3 arithmetic: the bounds of A^T were left behind it by the forward; the per-row maxima of S' come from the transform
    // along w below (every instance measures them, in every arithmetic: the ring kernels count the atomic in their waits)
    float* amax = const_cast<float*>(at) + ((q.at_floats + 3) & ~3L);
    float* smax = amax + q.o_smax;
    // bf16 STORAGE of S' (round 4): one-part mode on a ring geometry (see tvae_conv1_fwd_dft)
    constexpr bool bf16_store = true;
    const bool s16 = bf16_store && parts == 1 && q.ring;
    hipLaunchKernelGGL(h3_zero_slots_kernel, dim3(1), dim3(256), 0, st, smax, q.M);
    TVAE_CHECK_LAUNCH();
    float* Sp = ws;
    float* slabs = Sp + ((q.t_floats + 3) & ~3L);
    float* tab = slabs + (((long)(q.splits > q.wsplits ? q.splits : q.wsplits) * q.g_floats + 3) & ~3L);
    float* EO = tab;
    float* ED = EO + ((q.eo_floats + 3) & ~3L);
    {
        hipLaunchKernelGGL(dft_wtab_kernel, dim3(64), dim3(256), 0, st, EO, ED, q.L, q.Lh, q.Ho, q.LHP, q.NT,
                           q.NT + q.REM1, q.NS, q.NRT, q.mixed ? 1.f / (float)q.L : 1.f / ((float)q.L * (float)q.L));
        TVAE_CHECK_LAUNCH();
        const long ntiles = (long)q.M * (q.NBpad / 32);
        if (q.ring) {
            const int cus = dev_cu_count();
            const int grid = (int)((ntiles + 3) / 4 < 2 * cus ? (ntiles + 3) / 4 : 2 * cus);
            hipError_t er = hipSuccess;
#define TVAE_DY_RING_ONE(S_, T_, L2_, H_, Q_, S16_)                                                                  \
    do {                                                                                                            \
        const size_t lds_r = (size_t)4 * 2 * ((32 * H_ + 63) / 64) * 256;                                           \
        er = allow_big_lds(dft_dy_ring_kernel<S_, T_, L2_, H_, Q_, S16_>, lds_r);                                   \
        if (er != hipSuccess) return (int)er;                                                                       \
        hipLaunchKernelGGL((dft_dy_ring_kernel<S_, T_, L2_, H_, Q_, S16_>), dim3(grid), dim3(256), lds_r, st, dpre, \
                           (const float*)ED, Sp, q.M, R, B, q.Lh, q.NBpad, smax);                                   \
    } while (0)
#define TVAE_DY_RING(S_, T_, L2_, H_, Q_)                                                                           \
    do { if (s16) TVAE_DY_RING_ONE(S_, T_, L2_, H_, Q_, true); else TVAE_DY_RING_ONE(S_, T_, L2_, H_, Q_, false); } while (0)
            if (q.ring == 1) TVAE_DY_RING(9, 2, 38, 17, false);
            else if (q.ring == 2) TVAE_DY_RING(17, 3, 82, 33, false);
            else if (q.ring == 3) TVAE_DY_RING(20, 2, 62, 39, false);
            else if (q.ring == 4) TVAE_DY_RING(9, 2, 46, 17, false);
            else if (q.ring == 5) TVAE_DY_RING(17, 3, 98, 33, true);
            else TVAE_DY_RING(20, 3, 68, 39, false);
#undef TVAE_DY_RING
#undef TVAE_DY_RING_ONE
            TVAE_CHECK_LAUNCH();
        } else if (q.h3w && parts == 2) {
            // round 6: the same transform on the 16-bit matrix pipe (conv_dft_h3_kernels.hpp)
            uint4* EOc = reinterpret_cast<uint4*>(ED + ((q.ed_floats + 3) & ~3L) + ((q.M + 3) & ~3L));
            uint4* EDc = EOc + q.eoc_floats / 4;
            hipLaunchKernelGGL(dft_wtab_h3_kernel, dim3(64), dim3(256), 0, st, EOc, EDc, q.L, q.Lh, q.Ho, 11, 5, 9, 6,
                               q.mixed ? 1.f / (float)q.L : 1.f / ((float)q.L * (float)q.L));
            TVAE_CHECK_LAUNCH();
            const size_t lds_h = ((size_t)((32 * (q.Ho | 1) + 3) & ~3) + 2 * 2 * 9 * 32 * 4 + 32) * 4;
            hipError_t eh = allow_big_lds(dft_dy_h3_kernel<9, 6, 11>, lds_h);
            if (eh != hipSuccess) return (int)eh;
            const int cus = dev_cu_count();
            const int gridh = (int)(ntiles < 2L * cus ? ntiles : 2L * cus);
            hipLaunchKernelGGL((dft_dy_h3_kernel<9, 6, 11>), dim3(gridh), dim3(384), lds_h, st, dpre, (const uint4*)EDc, Sp, q.M,
                               R, B, q.Ho, q.Lh, q.NBpad, ldexpf(1.f, -14), smax);
            TVAE_CHECK_LAUNCH();
        } else if (q.gen && q.NS <= DFT_WIDE_NS &&
                   q.NRT <= 8 && 32 * q.Ho <= 16 * 64 * q.NRT && (size_t)2 * (32 * (q.Ho | 1) + 1) * 4 <= 150 * 1024) {
            // large frames (galaxy shape): a workgroup per tile, its waves split the rows of S' (dft_dy_wide_kernel)
            const int NSR = q.NS <= 50 ? 50 : (q.NS <= 66 ? 66 : DFT_WIDE_NS);      // the instance's padded slot
            const size_t lds_w = (size_t)2 * (32 * (q.Ho | 1) + 2 * NSR + 2) * 4;
            const int cus = dev_cu_count();
            const long fit = (long)(150 * 1024 / lds_w);
            const long wg_per_cu = fit < 2 ? fit : 2;
            const int gridw = (int)(ntiles < wg_per_cu * cus ? ntiles : wg_per_cu * cus);
            hipError_t ew = hipSuccess;
#define TVAE_DY_WIDE(NSR_, NLD_, WPE_)                                                                               \
    do {                                                                                                            \
        ew = allow_big_lds(dft_dy_wide_kernel<NSR_, NLD_, WPE_>, lds_w);                                            \
        if (ew != hipSuccess) return (int)ew;                                                                       \
        hipLaunchKernelGGL((dft_dy_wide_kernel<NSR_, NLD_, WPE_>), dim3(gridw), dim3(64 * q.NRT), lds_w, st, dpre,  \
                           (const float*)ED, Sp, q.M, R, B, q.Ho, q.Lh, q.NBpad, q.NS, q.NRT, smax);                \
    } while (0)
            // (an instance at three waves per SIMD -- two of the galaxy shape's six-wave workgroups per CU, 168 registers with 17
            //  spilled -- measured 2.17 ms against 2.05 for this one)
            if (q.NS <= 50) TVAE_DY_WIDE(50, 16, 2); else if (q.NS <= 66) TVAE_DY_WIDE(66, 16, 2); else TVAE_DY_WIDE(DFT_WIDE_NS, 16, 2);
#undef TVAE_DY_WIDE
            TVAE_CHECK_LAUNCH();
        } else if (q.gen) {
            const size_t lds_g = (size_t)4 * 32 * ((2 * q.NS) | 1) * 4;
            hipError_t eg = allow_big_lds(dft_dy_gen_kernel, lds_g);
            if (eg != hipSuccess) return (int)eg;
            const int gridg = (int)((ntiles + 3) / 4 < 4096 ? (ntiles + 3) / 4 : 4096);
            hipLaunchKernelGGL(dft_dy_gen_kernel, dim3(gridg), dim3(256), lds_g, st, dpre, (const float*)ED, Sp, q.M, R, B,
                               q.Ho, q.Lh, q.NBpad, q.NS, q.NRT, smax);
            TVAE_CHECK_LAUNCH();
        } else {
        const size_t lds_d = ((size_t)q.NS * q.NRT * 64 + (size_t)4 * (32 * ((2 * q.NS) | 1) + 64)) * 4;
        const int grid = (int)((ntiles + 3) / 4 < 768 ? (ntiles + 3) / 4 : 768);
        const int iters = (int)((ntiles + 4L * grid - 1) / (4L * grid));
        hipError_t e0 = hipSuccess;
#define TVAE_DY_MF(S_, T_, L2_, A_)                                                                                 \
    do {                                                                                                            \
        e0 = allow_big_lds(dft_dy_mf_kernel<S_, T_, L2_, A_>, lds_d);                                               \
        if (e0 != hipSuccess) return (int)e0;                                                                       \
        hipLaunchKernelGGL((dft_dy_mf_kernel<S_, T_, L2_, A_>), dim3(grid), dim3(256), lds_d, st, dpre,             \
                           (const float*)ED, Sp, q.M, R, B, q.Ho, q.Lh, q.NBpad, iters,                             \
                           smax);                                                        \
    } while (0)
        if (q.NS == 9) { if (q.Lh == 23) TVAE_DY_MF(9, 2, 46, true); else TVAE_DY_MF(9, 2, 0, true); }
        else if (q.NS == 17) { if (q.Lh == 49) TVAE_DY_MF(17, 4, 98, true); else TVAE_DY_MF(17, 4, 0, true); }
        else TVAE_DY_MF(32, 4, 0, false);
#undef TVAE_DY_MF
        TVAE_CHECK_LAUNCH();
        }
    }
    if (dbias) {
        float* dbpart = ED + ((q.ed_floats + 3) & ~3L);           // M floats behind the transform tables (then the h3 cell tables)
        hipLaunchKernelGGL(dft_dbias_rows_kernel, dim3(q.M), dim3(256), 0, st, (const float*)Sp, dbpart, q.Lh, q.NB, q.M,
                           s16 ? 1 : 0);
        TVAE_CHECK_LAUNCH();
        hipLaunchKernelGGL(dft_dbias_kernel, dim3((C + 63) / 64), dim3(64), 0, st, (const float*)dbpart, dbias, R, C);
        TVAE_CHECK_LAUNCH();
    }
    // G[fx][m'][k] = sum_n S'[fx][m'][n] A^T[fx][k][n]: batched split-pipe weight-gradient GEMM into reduction-slice slabs
    // exact-fit tile where the 2 L Cin columns are one and a half of the 128-wide tiles (dense_wgrad_x6_wide_kernel)
    constexpr bool wide_on = true;
    const bool wide = wide_on && parts != 1 && q.wsplits > 0;       // (the one-part mode keeps its bf16-stored S' path)
    const int nslabs = wide ? q.wsplits : q.splits;
    {
        const int M2 = 2 * q.M;
        const int tiles_b = wide ? M2 / WW_ROWS : q.Mb / DX6_ROWS, tilesM = q.Lh * tiles_b;
        const int tilesK = wide ? (q.K2 <= 160 ? 1 : cdiv(q.K2, 192)) : cdiv(q.K2, 128);
        // 8 reduction slices at the 64x64 configuration: TileMap deals the slices round-robin to the 8 XCDs
        const int splits = nslabs;
        const int nchunk = cdiv(cdiv((int)q.NBpad, splits), 16) * 16;
        const TileMap tmk{tilesM, tilesK, splits};
        const DenseBatch bt{tiles_b, (long)q.K2 * q.NBpad, 0};
        int rc = dense_wgrad_x6_batched(Sp, (long)q.Lh * 128, at, q.NBpad, slabs, M2, q.K2, (int)q.NBpad, nchunk, tmk, bt,
                                        128L, ATile{7, 127, (long)M2 * q.Lh * 128}, parts, st,
                                        // h3: one scale per filter row of S' (rows m and M + m), one per frequency of A^T
                                        H3Scale{smax, amax + q.o_fmax, 1, 0, q.M, 1 << 30, 1}, s16, wide);
        if (rc) return rc;
    }
    // both contractions on the fp32 matrix pipe where the tiles are not mostly padding (ksz <= 64, >= 17 frequencies per block)
    constexpr bool dbank_mf = true;
    if (q.mixed) {       // the tap rows are spatial already: x stage only
        hipError_t e = allow_big_lds(dft_dbank_x_kernel, q.lds_db);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(dft_dbank_x_kernel, dim3(q.M * Cin), dim3(256), q.lds_db, st, (const float*)slabs, nslabs, q.g_floats,
                           dbank, ksz, q.L, q.Lh, q.M, Cin);
    } else if (dbank_mf && ksz <= 64 && q.FXBd >= 17 && q.FXBd <= 64) {
        hipError_t e = allow_big_lds(dft_dbank_mf_kernel, q.lds_db);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(dft_dbank_mf_kernel, dim3(q.M * Cin), dim3(256), q.lds_db, st, (const float*)slabs, nslabs,
                           q.g_floats, dbank, ksz, q.L, q.Lh, q.M, Cin, q.FXBd);
    } else {
        hipError_t e = allow_big_lds(dft_dbank_kernel, q.lds_db);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(dft_dbank_kernel, dim3(q.M * Cin), dim3(256), q.lds_db, st, (const float*)slabs, nslabs, q.g_floats,
                           dbank, ksz, q.L, q.Lh, q.M, Cin, q.FXBd);
    }
    TVAE_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
