// Lifting convolution through the frequency domain, with the DFTs and the spectral contraction written as GEMMs.
//
// The reference's conv1 correlates a 64x64-tap filter bank with a 96x96 padded image to get 33x33 outputs: 8.9 MFLOP
// per (image, filter) in the direct form, and the direct kernels sit on the board power limit of the matrix pipe.
// With the circular-correlation theorem on an L x L frame (L = n + 2*pad; outputs h, w <= L - ksz are alias free)
//     out[b][m][h][w] = 1/L^2 * Re sum_{fx < Lh} c_fx e^{+2 pi i fx w / L}  T[fx][(b,h)][m]
//     T[fx][(b,h)][m] = sum_{fy < L} ( Yh[b][fy][fx] e^{+2 pi i fy h / L} ) * conj( Kh[m][fy][fx] )
// (Lh = L/2 + 1 by Hermitian symmetry, c_fx = 1 for fx = 0 and fx = L/2, else 2) the heavy part is, for every fx,
// ONE complex GEMM  [(b,h): B*Ho] x [fy: L] x [m: C*R]  = 326 GFLOP per launch instead of 2 339 GFLOP, and it runs
// on the split-bf16 dense kernels (dense_x6_kernels.hpp) as a real GEMM with the complex structure folded into the
// operands:  [Tr; Ti] = [[Kr, Ki], [-Ki, Kr]] . [Ar; Ai]   (rows m | M+m, reduction index (re/im, fy)).
// The weight gradient is the same contraction transposed:
//     dKh'[m][fy][fx] = sum_{(b,h)} conj(S[fx][m][(b,h)]) * A[fx][(b,h)][fy],   S = DFT over w of dY,
// followed by a small inverse DFT per filter.  Everything else here is small streaming work:
//   dft_spectra_kernel y -> A^T[fx][(re/im, fy)][(b,h)]           (one workgroup per image, DFT by direct sums in LDS)
//                      bank -> W[fx][m | M+m][(re/im, fy)]         (one workgroup per filter; same launch)
//   dft_out_mf_kernel  T -> out (+bias, activation)                (contraction over fx: fp32 MFMA, constant operand in LDS)
//   dft_dy_mf_kernel   dY -> S'[m | M+m][fx][(b,h)]                (DFT over w: fp32 MFMA, constant operand in registers)
//   dft_dbank_kernel   G[fx][m | M+m][(re/im, fy)] -> dbank        (inverse DFT per filter, cropped to ksz x ksz)
// Single input channel (Cin = 1: MNIST / particle configurations).  Twiddles: sincospi of exactly reduced angles.
#pragma once
#include <hip/hip_runtime.h>
#include "small_kernels.hpp"
#include "conv_x6_kernels.hpp"

namespace tvae {

// T / S' layout: [n >> 7][row m' < 2M][fx < Lh][n & 127] -- the 2*Lh frequency rows of one (m, 128 columns) are one
// contiguous run for the transforms along w, and a GEMM tile (512 rows x 128 columns of one fx) touches 512-byte runs
// 512*Lh bytes apart instead of Lh*NBpad*4 bytes apart.
__device__ __forceinline__ long dft_t_off(long n, int row, int rows2, int Lh) {
    return (((n >> 7) * rows2 + row) * Lh) * 128 + (n & 127);            // + fx * 128
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// tw[j] = e^{+2 pi i j / L}, j < L, into LDS
__device__ __forceinline__ void fill_twiddles(float2* tw, int L) {
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)j / (float)L, &s, &c);
        tw[j] = make_float2(c, s);
    }
}

// ------------------------------------------------------------------------------------------
// Spectra of the images and of the rotated filters, ONE launch.  A workgroup owns one (image b | filter m, input channel
// ci, block of FXB frequencies fx0 .. fx0+FXB): it transforms its plane by direct sums in LDS (rows, then columns) and
// writes the operands of the spectral GEMM.  The reduction index of that GEMM is k = (ci*2 + ri)*L + fy (K2 = 2*L*Cin):
//   images : AT[fx][k][b*Ho + h] = Re / Im ( Yh[b][ci][fy][fx] e^{+2 pi i fy h / L} )
//   filters: W[fx][m][k] = (Kr | Ki),  W[fx][M+m][k] = (-Ki | Kr)    (the complex product folded into a real GEMM)
// One block of all Lh frequencies for the 96-wide frame of the 64x64 configuration (80 KB of LDS); the 192-wide frame of
// the galaxy configuration (n = 128) takes more, smaller blocks (the host sizes FXB for >= 3 workgroups per CU).  LDS: row
// transform S*FXB complex, plane spectrum L*FXB complex, twiddles L complex (S = n for images, ksz for filters).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void dft_plane_spectrum(float* sm_dft, const float* __restrict__ pl, int S, int pad, int L,
                                                   int fx0, int nfx, int FXB, float2*& Yh, float2*& tw) {
    float2* Rw = reinterpret_cast<float2*>(sm_dft);
    Yh = Rw + S * FXB;
    tw = Yh + L * FXB;
    fill_twiddles(tw, L);
    __syncthreads();
    // Rw[yy][f] = sum_x pl[yy][x] e^{-2 pi i fx (x+pad) / L},  fx = fx0 + f.  The plane is read straight from memory:
    // consecutive threads share yy, so a load is one broadcast line (L1), and the 16-64 KB an LDS copy would take
    // decides how many workgroups fit a CU -- these loops are latency chains that need the occupancy.
    for (int i = threadIdx.x; i < S * nfx; i += blockDim.x) {
        const int yy = i / nfx, f = i - yy * nfx, fx = fx0 + f;
        const float* row = pl + yy * S;
        float re = 0.f, im = 0.f;
        int ph = (fx * pad) % L;
#pragma unroll 4
        for (int x = 0; x < S; ++x) {
            const float v = row[x];
            const float2 t = tw[ph];
            re += v * t.x;
            im -= v * t.y;
            ph += fx;
            if (ph >= L) ph -= L;
        }
        Rw[yy * FXB + f] = make_float2(re, im);
    }
    __syncthreads();
#ifdef TVAE_SPECTRA_EXTRA_SYNC
    __builtin_amdgcn_s_waitcnt(0);                       // investigation build (profiles/README.md round 4): drain every counter,
    __syncthreads();                                     // a second barrier after the row transform
#endif
    // Yh[fy][f] = sum_y Rw[y][f] e^{-2 pi i fy (y+pad) / L}
    if ((L & 3) == 0) {
        // radix-4 step over the output index: the four outputs fy + q L/4 share their twiddles up to (-i)^(q (y+pad)), so
        // one pass builds the four partial sums by (y + pad) mod 4 and a 4-point DFT combines them (a quarter of the MACs)
        const int Lq = L >> 2;
        for (int i = threadIdx.x; i < Lq * nfx; i += blockDim.x) {
            const int fy = i / nfx, f = i - fy * nfx;
            // the four partial sums are NAMED registers and a term with a run-time residue is added through selects: an
            // array / by-reference capture here ended up in scratch memory (40 bytes per lane), and scratch is not
            // dependable when several processes time-slice the GPU (profiles/README.md, round 3: the filter spectra of
            // one workgroup in ~10^-4 came out wrong under sharing, and nothing else in that configuration used scratch)
            float a0x = 0.f, a0y = 0.f, a1x = 0.f, a1y = 0.f, a2x = 0.f, a2y = 0.f, a3x = 0.f, a3y = 0.f;
            int ph = (fy * pad) % L;
            int yy = 0;
#define TVAE_DFT_TERM_P(P_)                                                                      \
    const float2 P_ = cmul(Rw[yy * FXB + f], make_float2(tw[ph].x, -tw[ph].y));                  \
    ph += fy;                                                                                    \
    if (ph >= L) ph -= L;                                                                        \
    ++yy;
#define TVAE_DFT_TERM_C(AX_, AY_) { TVAE_DFT_TERM_P(p_) AX_ += p_.x; AY_ += p_.y; }
#define TVAE_DFT_TERM_R(R_)                                                                      \
    {                                                                                            \
        TVAE_DFT_TERM_P(p_)                                                                      \
        a0x += (R_) == 0 ? p_.x : 0.f; a0y += (R_) == 0 ? p_.y : 0.f;                            \
        a1x += (R_) == 1 ? p_.x : 0.f; a1y += (R_) == 1 ? p_.y : 0.f;                            \
        a2x += (R_) == 2 ? p_.x : 0.f; a2y += (R_) == 2 ? p_.y : 0.f;                            \
        a3x += (R_) == 3 ? p_.x : 0.f; a3y += (R_) == 3 ? p_.y : 0.f;                            \
    }
            // bring (yy + pad) to a multiple of 4 first: the residue of every later term is then known at compile time
            int r0 = pad & 3;
            while (r0 != 0 && r0 < 4 && yy < S) { TVAE_DFT_TERM_R(r0) ++r0; }
            while (yy + 3 < S) {
                TVAE_DFT_TERM_C(a0x, a0y) TVAE_DFT_TERM_C(a1x, a1y) TVAE_DFT_TERM_C(a2x, a2y) TVAE_DFT_TERM_C(a3x, a3y)
            }
            r0 = 0;
            while (yy < S) { TVAE_DFT_TERM_R(r0) ++r0; }
#undef TVAE_DFT_TERM_R
#undef TVAE_DFT_TERM_C
#undef TVAE_DFT_TERM_P
            const float2 a[4] = {make_float2(a0x, a0y), make_float2(a1x, a1y), make_float2(a2x, a2y), make_float2(a3x, a3y)};
            // Y[fy + q Lq] = sum_r a[r] (-i)^(q r):   (-i)^0 = 1, (-i)^1 = -i, (-i)^2 = -1, (-i)^3 = i
            const float2 s02 = make_float2(a[0].x + a[2].x, a[0].y + a[2].y), d02 = make_float2(a[0].x - a[2].x, a[0].y - a[2].y);
            const float2 s13 = make_float2(a[1].x + a[3].x, a[1].y + a[3].y), d13 = make_float2(a[1].x - a[3].x, a[1].y - a[3].y);
            Yh[fy * FXB + f] = make_float2(s02.x + s13.x, s02.y + s13.y);
            Yh[(fy + Lq) * FXB + f] = make_float2(d02.x + d13.y, d02.y - d13.x);          // a0 - i a1 - a2 + i a3
            Yh[(fy + 2 * Lq) * FXB + f] = make_float2(s02.x - s13.x, s02.y - s13.y);
            Yh[(fy + 3 * Lq) * FXB + f] = make_float2(d02.x - d13.y, d02.y + d13.x);      // a0 + i a1 - a2 - i a3
        }
    } else {
        for (int i = threadIdx.x; i < L * nfx; i += blockDim.x) {
            const int fy = i / nfx, f = i - fy * nfx;
            float2 acc = make_float2(0.f, 0.f);
            int ph = (fy * pad) % L;
#pragma unroll 4
            for (int yy = 0; yy < S; ++yy) {
                const float2 p = cmul(Rw[yy * FXB + f], make_float2(tw[ph].x, -tw[ph].y));
                acc.x += p.x;
                acc.y += p.y;
                ph += fy;
                if (ph >= L) ph -= L;
            }
            Yh[fy * FXB + f] = acc;
        }
    }
    __syncthreads();
}

// h3 arithmetic of the spectral GEMMs (round 4: one power-of-two scale per row / column group instead of one per tensor):
// the workgroup that owns a plane and a block of frequencies also leaves, per frequency fx of its block,
//   images : cmax[fx * B + b]   >= max over (ci, fy, h) |A^T[fx][(ci, ri, fy)][(b, h)]|  (the modulus |Yh| bounds both parts of
//            Yh e^{i..}): the scale of image b's columns in problem fx of the forward GEMM;  fmax[fx] = max over b
//            (dft_fmax_kernel): the scale of the rows of A^T in problem fx of the weight-gradient GEMM;
//   filters: wmax[fx * Mb + m] = wmax[fx * Mb + M + m] = max over (ci, fy) of |Kr|, |Ki|: the scale of the two stacked rows.
// (atomic maxima on the bit patterns: Cin planes meet in a slot; the slots are zeroed by the entry point.)  An image
// 2^-24 as bright as its neighbours, a filter that has not started to train, the DC plane of un-normalised data: each gets
// the full two-part precision relative to ITSELF.
struct DftMax {
    float* cmax;    // [Lh][B]
    float* fmax;    // [Lh]   (dft_fmax_kernel: max over the images of cmax)
    float* wmax;    // [Lh][Mb]
};
// a slot has one writer per input channel: a plain store with one channel, else a fire-and-forget atomic maximum (no read
// first: a dependent load per frequency in lane 0 cost the spectra kernel 0.17 ms)
__device__ __forceinline__ void dft_max_slot(float* slot, float v, bool single) {
    if (single) *slot = v;
    else __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(slot), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// fmax[fx] = max_b cmax[fx][b]: one workgroup per frequency (256 atomics from as many workgroups on two cache lines would
// serialise in the memory system)
static __global__ void dft_fmax_kernel(const float* __restrict__ cmax, float* __restrict__ fmax, int B) {
    float m = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) m = fmaxf(m, cmax[(long)blockIdx.x * B + b]);
    h3_block_amax(m, fmax + blockIdx.x);
}
static __global__ void dft_spectra_kernel(const float* __restrict__ y, float* __restrict__ AT, int B, int Cin, int n, int pad,
                                          int Ho, long NBpad, const float* __restrict__ bank, float* __restrict__ W, int ksz,
                                          int M, int Mb, int L, int Lh, int FXB, int nblk, DftMax mxp) {
    extern __shared__ float sm_dft[];
    const int nimg = B * Cin * nblk;
    int id = blockIdx.x;
    const bool is_img = id < nimg;
    if (!is_img) id -= nimg;
    const int blk = id % nblk;
    const int ci = (id / nblk) % Cin;
    const int outer = id / (nblk * Cin);                 // image b or filter m
    const int fx0 = blk * FXB, nfx = min(FXB, Lh - fx0);
    float2 *Yh, *tw;
    if (is_img) {
        dft_plane_spectrum(sm_dft, y + ((long)outer * Cin + ci) * n * n, n, pad, L, fx0, nfx, FXB, Yh, tw);
        const int total = nfx * L * Ho;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int h = i % Ho;
            const int t2 = i / Ho;
            const int fy = t2 % L, f = t2 / L;
            const float2 v = cmul(Yh[fy * FXB + f], tw[(fy * h) % L]);
            float* dst = AT + ((long)(fx0 + f) * 2 * L * Cin + (long)(2 * ci) * L + fy) * NBpad + (long)outer * Ho + h;
            dst[0] = v.x;
            dst[(long)L * NBpad] = v.y;
        }
        // one frequency per wave and turn: lanes over fy, wave maximum of the SQUARED modulus (it bounds |Re|, |Im| of
        // Yh e^{i phi} for every phi), rounded up a little after the root
        for (int f = threadIdx.x >> 6; f < nfx; f += blockDim.x >> 6) {
            float q = 0.f;
            for (int fy = threadIdx.x & 63; fy < L; fy += 64) {
                const float2 k = Yh[fy * FXB + f];
                q = fmaxf(q, __fmaf_rn(k.x, k.x, k.y * k.y));
            }
            q = h3_wave_max(q);
            if ((threadIdx.x & 63) == 0) dft_max_slot(mxp.cmax + (long)(fx0 + f) * B + outer, sqrtf(q) * 1.000001f, Cin == 1);
        }
    } else {
        const int m = outer;
        dft_plane_spectrum(sm_dft, bank + ((long)m * Cin + ci) * ksz * ksz, ksz, 0, L, fx0, nfx, FXB, Yh, tw);
        const long rowlen = 2L * L * Cin;
        for (int i = threadIdx.x; i < nfx * L; i += blockDim.x) {
            const int fy = i % L, f = i / L;
            const float2 k = Yh[fy * FXB + f];
            float* r0 = W + ((long)(fx0 + f) * Mb + m) * rowlen + (long)(2 * ci) * L;   // Mb >= 2M rows per fx (padding rows stay zero)
            float* r1 = W + ((long)(fx0 + f) * Mb + M + m) * rowlen + (long)(2 * ci) * L;
            r0[fy] = k.x;
            r0[L + fy] = k.y;
            r1[fy] = -k.y;
            r1[L + fy] = k.x;
        }
        for (int f = threadIdx.x >> 6; f < nfx; f += blockDim.x >> 6) {
            float q = 0.f;
            for (int fy = threadIdx.x & 63; fy < L; fy += 64) {
                const float2 k = Yh[fy * FXB + f];
                q = fmaxf(q, fmaxf(fabsf(k.x), fabsf(k.y)));
            }
            q = h3_wave_max(q);
            if ((threadIdx.x & 63) == 0) {
                dft_max_slot(mxp.wmax + (long)(fx0 + f) * Mb + m, q, Cin == 1);
                dft_max_slot(mxp.wmax + (long)(fx0 + f) * Mb + M + m, q, Cin == 1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 5: the MIXED form -- frequency domain along x only, the correlation along y stays spatial.
//     out[b][m][h][w] = 1/L * Re sum_{fx < Lh} c_fx e^{+2 pi i fx w / L}  T[fx][(b,h)][m]
//     T[fx][(b,h)][m] = sum_{ci} sum_{u < ksz} X[b][ci][h + u - pad][fx] * conj( Kx[m][ci][u][fx] )
// with X / Kx the 1-D DFTs ALONG x of the image rows (at frame positions x + pad) and of the filter rows.  The reduction of the
// spectral GEMM is then k = (ci*2 + ri)*ksz + u: K2 = 2 ksz Cin instead of 2 L Cin (128 instead of 160 at the 64 x 64 shape, 384
// instead of 960 at the galaxy shape) -- ksz <= L always, so it never loses -- T, S' and the transforms along w are unchanged,
// the spectra need no transform along y and no phase per output row (the GEMM operand is a Toeplitz view of the row spectra:
// AT[fx][k][(b,h)] = X[b][ci][h + u - pad][fx], zero outside the image), and the inverse transform of the weight gradient
// needs its x stage only.  Same launch shape as dft_spectra_kernel (a workgroup per (plane, block of frequencies)).
// LDS: row spectra S*FXB complex + twiddles L complex.
// ------------------------------------------------------------------------------------------
// tail shared by both forms of the row transform: GEMM operands and operand maxima from the row spectra Rw [row][f]
__device__ __forceinline__ void dft_spectra_x_tail(const float2* Rw, bool is_img, int outer, int ci, int fx0, int nfx, int FXB,
                                                   float* __restrict__ AT, int B, int Cin, int n, int pad, int Ho, long NBpad,
                                                   float* __restrict__ W, int ksz, int M, int Mb, DftMax mxp) {
    const long rowlen = 2L * ksz * Cin;
    if (is_img) {
        // AT[fx][(2 ci + ri) ksz + u][b Ho + h] = Re / Im X[h + u - pad][fx]  (h fastest: coalesced runs of Ho floats)
        // element i = (f * ksz + u) * Ho + h, h fastest (consecutive threads write consecutive addresses); the indices advance
        // incrementally: two integer divisions per element made this loop, not the transform, the kernel's critical path
        const int total = nfx * ksz * Ho, nth = blockDim.x;
        const int dh = nth % Ho, dt = nth / Ho;          // i += nth:  h += dh (carry into t2), t2 += dt
        int h = threadIdx.x % Ho, t2 = threadIdx.x / Ho;
        int u = t2 % ksz, f = t2 / ksz;
        for (int i = threadIdx.x; i < total; i += nth) {
            const int r = h + u - pad;
            const float2 v = (r >= 0 && r < n) ? Rw[r * FXB + f] : make_float2(0.f, 0.f);
            float* dst = AT + ((long)(fx0 + f) * rowlen + (long)(2 * ci) * ksz + u) * NBpad + (long)outer * Ho + h;
            dst[0] = v.x;
            dst[(long)ksz * NBpad] = v.y;
            h += dh;
            int du = dt;
            if (h >= Ho) { h -= Ho; ++du; }
            u += du;
            while (u >= ksz) { u -= ksz; ++f; }
        }
        for (int f = threadIdx.x >> 6; f < nfx; f += blockDim.x >> 6) {       // one frequency per wave and turn
            float q = 0.f;
            for (int r = threadIdx.x & 63; r < n; r += 64) {
                const float2 k = Rw[r * FXB + f];
                q = fmaxf(q, fmaxf(fabsf(k.x), fabsf(k.y)));
            }
            q = h3_wave_max(q);
            if ((threadIdx.x & 63) == 0) dft_max_slot(mxp.cmax + (long)(fx0 + f) * B + outer, q, Cin == 1);
        }
    } else {
        const int m = outer;
        for (int i = threadIdx.x; i < nfx * ksz; i += blockDim.x) {
            const int u = i % ksz, f = i / ksz;
            const float2 k = Rw[u * FXB + f];
            float* r0 = W + ((long)(fx0 + f) * Mb + m) * rowlen + (long)(2 * ci) * ksz;
            float* r1 = W + ((long)(fx0 + f) * Mb + M + m) * rowlen + (long)(2 * ci) * ksz;
            r0[u] = k.x;
            r0[ksz + u] = k.y;
            r1[u] = -k.y;
            r1[ksz + u] = k.x;
        }
        for (int f = threadIdx.x >> 6; f < nfx; f += blockDim.x >> 6) {
            float q = 0.f;
            for (int u = threadIdx.x & 63; u < ksz; u += 64) {
                const float2 k = Rw[u * FXB + f];
                q = fmaxf(q, fmaxf(fabsf(k.x), fabsf(k.y)));
            }
            q = h3_wave_max(q);
            if ((threadIdx.x & 63) == 0) {
                dft_max_slot(mxp.wmax + (long)(fx0 + f) * Mb + m, q, Cin == 1);
                dft_max_slot(mxp.wmax + (long)(fx0 + f) * Mb + M + m, q, Cin == 1);
            }
        }
    }
}

static __global__ void dft_spectra_x_kernel(const float* __restrict__ y, float* __restrict__ AT, int B, int Cin, int n, int pad,
                                            int Ho, long NBpad, const float* __restrict__ bank, float* __restrict__ W, int ksz,
                                            int M, int Mb, int L, int Lh, int FXB, int nblk, int FXBf, int nblkf, DftMax mxp) {
    extern __shared__ float sm_dft[];
    // images and filters have their own number of frequency blocks per plane (a filter's workgroup is cheap and there are
    // C R Cin of them: splitting those as well only multiplies fixed costs)
    const int nimg = B * Cin * nblk;
    int id = blockIdx.x;
    const bool is_img = id < nimg;
    if (!is_img) {
        id -= nimg;
        nblk = nblkf;
        FXB = FXBf;
    }
    const int blk = id % nblk;
    const int ci = (id / nblk) % Cin;
    const int outer = id / (nblk * Cin);                 // image b or filter m
    const int fx0 = blk * FXB, nfx = min(FXB, Lh - fx0);
    const int S = is_img ? n : ksz, pd = is_img ? pad : 0;
    const float* pl = is_img ? y + ((long)outer * Cin + ci) * n * n : bank + ((long)outer * Cin + ci) * ksz * ksz;
    float2* Rw = reinterpret_cast<float2*>(sm_dft);      // [row][f]
    float2* tw = Rw + S * FXB;
    fill_twiddles(tw, L);
    __syncthreads();
    // Rw[yy][f] = sum_x pl[yy][x] e^{-2 pi i fx (x + pd) / L}  (the plane is read straight from memory: see dft_plane_spectrum)
    for (int i = threadIdx.x; i < S * nfx; i += blockDim.x) {
        const int yy = i / nfx, f = i - yy * nfx, fx = fx0 + f;
        const float* row = pl + yy * S;
        float re = 0.f, im = 0.f;
        int ph = (fx * pd) % L;
#pragma unroll 4
        for (int x = 0; x < S; ++x) {
            const float v = row[x];
            const float2 t = tw[ph];
            re += v * t.x;
            im -= v * t.y;
            ph += fx;
            if (ph >= L) ph -= L;
        }
        Rw[yy * FXB + f] = make_float2(re, im);
    }
    __syncthreads();
    dft_spectra_x_tail(Rw, is_img, outer, ci, fx0, nfx, FXB, AT, B, Cin, n, pad, Ho, NBpad, W, ksz, M, Mb, mxp);
}

// The same with the row transform on the fp32 matrix pipe (round 5): Rw[yy][(f, re | im)] = sum_x pl[yy][x] E[x][(f, re | im)] is
// a small GEMM per plane (64 x 64 x 82 at the 64 x 64 shape: 192 v_mfma_f32_32x32x2_f32 per plane, spread over four waves) where the
// direct sums above walk a dependent twiddle index per tap (0.11 ms of FIXED cost per step -- the 1 024 filter planes do not
// shrink with the batch).  A wave owns (row tile, column tile) pairs: its 32 plane rows sit in registers (KSR pairs of taps,
// zero padded: the chain is branch free), the twiddle operand comes from the LDS table with a running phase per lane.
template <int KSR>
static __global__ __launch_bounds__(256) void dft_spectra_x_mf_kernel(const float* __restrict__ y, float* __restrict__ AT, int B,
                                                                      int Cin, int n, int pad, int Ho, long NBpad,
                                                                      const float* __restrict__ bank, float* __restrict__ W,
                                                                      int ksz, int M, int Mb, int L, int Lh, int FXB, int nblk, int FXBf, int nblkf,
                                                                      DftMax mxp) {
    extern __shared__ float sm_dft[];
    // images and filters have their own number of frequency blocks per plane (a filter's workgroup is cheap and there are
    // C R Cin of them: splitting those as well only multiplies fixed costs)
    const int nimg = B * Cin * nblk;
    int id = blockIdx.x;
    const bool is_img = id < nimg;
    if (!is_img) {
        id -= nimg;
        nblk = nblkf;
        FXB = FXBf;
    }
    const int blk = id % nblk;
    const int ci = (id / nblk) % Cin;
    const int outer = id / (nblk * Cin);                 // image b or filter m
    const int fx0 = blk * FXB, nfx = min(FXB, Lh - fx0);
    const int S = is_img ? n : ksz, pd = is_img ? pad : 0;
    const float* pl = is_img ? y + ((long)outer * Cin + ci) * n * n : bank + ((long)outer * Cin + ci) * ksz * ksz;
    float2* Rw = reinterpret_cast<float2*>(sm_dft);      // [row][f]
    float* Rwf = sm_dft;                                 // the same as floats: [(row * FXB + f) * 2 + (re | im)]
    float2* tw = Rw + S * FXB;
    fill_twiddles(tw, L);
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int MT = (S + 31) / 32, NC = (2 * nfx + 31) / 32;
    int mt_have = -1;
    float a[KSR];
    for (int pair = wave; pair < MT * NC; pair += 4) {   // (wave uniform)
        const int mt = pair % MT, nt = pair / MT;
        if (mt != mt_have) {                             // this lane's plane row, taps 2 t + kh
            const int row = 32 * mt + li;
            const float* rp = pl + (long)min(row, S - 1) * S;
            const float ok = row < S ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < KSR; ++t) {
                const int x = 2 * t + kh;
                a[t] = x < S ? rp[x] * ok : 0.f;
            }
            mt_have = mt;
        }
        const int ncol = 32 * nt + li, f = ncol >> 1, ri = ncol & 1;
        const int fx = fx0 + min(f, nfx - 1);
        const float bs = f < nfx ? 1.f : 0.f;
        int ph = (int)(((long)fx * (kh + pd)) % L);
        const int dph = (2 * fx) % L;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < KSR; ++t) {
            const float2 w = tw[ph];
            const float bv = (ri ? -w.y : w.x) * bs;     // e^{-i theta}: (cos, -sin)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc, 0, 0, 0);
            ph += dph;
            if (ph >= L) ph -= L;
        }
        if (f < nfx) {                                   // D: lane = column (f, ri), registers = rows
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int yy = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (yy < S) Rwf[((long)yy * FXB + f) * 2 + ri] = acc[r];
            }
        }
    }
    __syncthreads();
    dft_spectra_x_tail(Rw, is_img, outer, ci, fx0, nfx, FXB, AT, B, Cin, n, pad, Ho, NBpad, W, ksz, M, Mb, mxp);
}

// ==========================================================================================
// The two transforms along w on the matrix pipe, in plain fp32 (v_mfma_f32_32x32x2_f32: exact products, fp32
// accumulate -- no operand splitting, so the vector ALU only moves data).  They are small GEMMs with a constant operand,
//     out[w][(m,n)] = sum_{(fx,ri)} E[w][(fx,ri)] T[(fx,ri)][(m,n)]      (K = 2*Lh, 32 output rows per MFMA tile)
//     S'[(fx,ri)][(m,n)] = sum_w E'[(fx,ri)][w] dY[(m,n)][w]              (K = Ho)
// and one wave owns a tile of 32 consecutive columns n of one filter row m.  A vector-ALU version issued
// 3 234 FMAs per (m,n) with one scalar table load per FMA pair and waited on the scalar cache 60 % of the time; here a
// tile costs 49 (resp. 68) MFMAs and the constant operand comes from LDS (one ds_read_b32 per 64-cycle MFMA).
// The values of the NEXT tile are loaded into registers while the current one is on the matrix pipe.
// ==========================================================================================
constexpr int DFT_WROWS = 64;          // largest output width of the matrix-pipe transforms (two 32-row tiles)

// EO[fx < LHP][t < NTT][lane]: t < NT:  E[w = 32t + (lane & 31)][(fx, ri = lane >> 5)] = c_fx/L^2 * (ri ? -sin : cos)(2 pi fx w / L)
//                              t == NT (REM1 only): the same for the single extra row w = 32*NT, identical in all 32 lanes
// ED[s < NS][rt < NRT][lane]:  E'[kk = 32 rt + (lane & 31)][w = 2s + (lane >> 5)], kk = 2 fx + ri: ri ? -sin : cos
// Entries outside fx < Lh, w < Ho are zero (they pad the loops of the kernels).
// norm: 1/L^2 (both axes in the frequency domain) or 1/L (mixed form: x only)
static __global__ void dft_wtab_kernel(float* __restrict__ EO, float* __restrict__ ED, int L, int Lh, int Ho, int LHP, int NT,
                                int NTT, int NS, int NRT, float norm) {
    const int nEO = LHP * NTT * 64, nED = NS * NRT * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nEO + nED; i += gridDim.x * blockDim.x) {
        int fx, w, ri;
        bool fwd;
        if (i < nEO) {
            const int lane = i & 63, t = (i >> 6) % NTT;
            fx = (i >> 6) / NTT;
            ri = lane >> 5;
            w = t < NT ? 32 * t + (lane & 31) : 32 * NT;
            fwd = true;
        } else {
            const int ii = i - nEO, lane = ii & 63, rt = (ii >> 6) % NRT, s = (ii >> 6) / NRT;
            const int kk = 32 * rt + (lane & 31);
            fx = kk >> 1;
            ri = kk & 1;
            w = 2 * s + (lane >> 5);
            fwd = false;
        }
        float v = 0.f;
        if (fx < Lh && w < Ho) {
            float sn, cs;
            sincospif(2.0f * (float)((fx * w) % L) / (float)L, &sn, &cs);
            v = ri ? -sn : cs;
            if (fwd) v *= (((fx == 0) || (2 * fx == L)) ? 1.f : 2.f) * norm;
        }
        if (i < nEO) EO[i] = v; else ED[i - nEO] = v;
    }
}

// out[c][b][r][h][w] = act(bias[c] + sum_k E[w][k] T[k][(m,n)]).  Block of 4 waves; wave q of block x walks the tiles
// (it*gridDim.x + x)*4 + q.  LHP >= Lh frequencies are processed (zero table rows beyond Lh), NT output tiles of 32 rows
// on the matrix pipe and, with REM1, the single extra row w = 32*NT as a dot product on the vector ALU.
template <int LHP, int NT, bool REM1>
static __global__ __launch_bounds__(256, 2) void dft_out_mf_kernel(const float* __restrict__ T, const float* __restrict__ EO,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int M, int R, int B, int Ho, int Lh, long NBpad, int act,
                                                            float slope, int iters, float* __restrict__ amax) {
    float amx = 0.f;                                             // max |out| per channel -> amax[C] (h3 scales of the encoder tail)
    constexpr int NTT = NT + (REM1 ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    float* eo = sm_w;                                            // [LHP][NTT][64]
    const int SWO = Ho | 1;                                      // odd row stride: conflict-free transposition
    float* stg = eo + LHP * NTT * 64 + (threadIdx.x >> 6) * 32 * SWO;   // this wave's [32 columns][SWO]
    for (int i = threadIdx.x; i < LHP * NTT * 64; i += 256) eo[i] = EO[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, ri = lane >> 5;
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    const float inv_ho = 1.0f / (float)Ho;
    auto tile_ptr = [&](long tile, int& m, long& n0) -> const float* {
        const long tl = tile < ntiles ? tile : ntiles - 1;       // past the end: reload the last tile (never stored)
        m = (int)(tl / tiles_n);
        n0 = (tl - (long)m * tiles_n) * 32;
        return T + dft_t_off(n0 + j, ri * M + m, 2 * M, Lh);
    };
    auto load_tile = [&](const float* tp, float (&v)[LHP]) {
#pragma unroll
        for (int fx = 0; fx < LHP; ++fx)      // LHP == 64: generic instance, padding rows meet zero weights; else LHP == Lh
            v[fx] = tp[(LHP == 64 ? (fx < Lh ? fx : Lh - 1) : fx) * 128];
    };
    // one tile: 49 MFMAs on two alternating accumulators (no back-to-back dependence), then the transposing store
    auto compute = [&](long tile, int m, long n0, const float (&v)[LHP]) {
        f32x16 acc[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][0][r] = acc[t][1][r] = 0.f;
        float racc = 0.f;
#pragma unroll
        for (int fx = 0; fx < LHP; ++fx) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t][fx & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(eo[(fx * NTT + t) * 64 + lane], v[fx], acc[t][fx & 1], 0, 0, 0);
            if (REM1) racc = __fmaf_rn(eo[(fx * NTT + NT) * 64 + lane], v[fx], racc);
        }
        if (tile >= ntiles) return;
        const int c = m / R, r_ = m - c * R;
        const float bv = bias ? bias[c] : 0.f;
        auto fin = [&](float x) {
            x += bv;
            if (act == ACT_LRELU) x = x > 0.f ? x : x * slope;
            else if (act == ACT_TANH) x = tanhf(x);
            return x;
        };
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * ri;
                if (32 * t + 31 < Ho || w < Ho) stg[j * SWO + w] = fin(acc[t][0][r] + acc[t][1][r]);
            }
        if (REM1) {
            const float tot = racc + __shfl_xor(racc, 32, 64);
            if (ri == 0) stg[j * SWO + 32 * NT] = fin(tot);
        }
        __builtin_amdgcn_wave_barrier();
        // the 32 output rows of the tile are 32*Ho consecutive floats of out, plus (R-1)*P for every image
        // boundary before the row (divisions through the float reciprocal: exact for these ranges)
        const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
        float* obase = out + (((long)c * B + b0) * R + r_) * P + (long)h0 * Ho;
        const int jump = (R - 1) * P;
        const int tmax = (int)(NB - n0 < 32 ? NB - n0 : 32);
        const int cnt = tmax * Ho;
        for (int e = lane; e < cnt; e += 64) {
            const int t = (int)(((float)e + 0.5f) * inv_ho), w = e - t * Ho;
            const int k = (int)(((float)(h0 + t) + 0.5f) * inv_ho);
            const float sv = stg[t * SWO + w];
            amx = fmaxf(amx, fabsf(sv));
            obase[e + k * jump] = sv;
        }
        h3_tile_flush_rd(amx, amax + c, lane);              // max |out| of channel c (h3 scales of the encoder tail)
        __builtin_amdgcn_wave_barrier();
    };
    // two register buffers, the loop walks two tiles per trip: the loads of one buffer are in flight while the other
    // is on the matrix pipe (no register copies: a copy would wait for the loads it moves)
    float va[LHP], vb[LHP];
    int ma, mb;
    long na, nb;
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    load_tile(tile_ptr(tile, ma, na), va);
    for (int it = 0; it < iters; it += 2) {
        load_tile(tile_ptr(tile + stride, mb, nb), vb);
        compute(tile, ma, na, va);
        load_tile(tile_ptr(tile + 2 * stride, ma, na), va);
        compute(tile + stride, mb, nb, vb);
        tile += 2 * stride;
    }
}

// S'[(fx,ri)][(m,n)] = sum_w E'[(fx,ri)][w] dY[(m,n)][w]; dY is [c][b][r][h][w].  Same tile walk as above.  The 32 x Ho
// values of a tile are (mostly) one contiguous run: coalesced loads into registers one tile ahead, transposed through
// this wave's LDS patch to the B-operand layout (column n = lane & 31, w = 2s + (lane >> 5)).
// LH2 = 2*Lh when known at compile time (row validity and store addresses fold), 0 = run-time check;
// AREG: the constant operand (NS*NRT values per lane) lives in registers for the whole kernel, else it is read from LDS.
template <int NS, int NRT, int LH2, bool AREG>
static __global__ __launch_bounds__(256, 2) void dft_dy_mf_kernel(const float* __restrict__ dY, const float* __restrict__ ED,
                                                           float* __restrict__ Sp, int M, int R, int B, int Ho, int Lh,
                                                           long NBpad, int iters, float* __restrict__ amax) {
    float amx = 0.f;                                             // max |S'| per filter row -> amax[M] (h3 scales of the GEMM that reads S')
    constexpr int NL = NS;                                       // 64-element slices of the 32 x Ho tile: ceil(32*Ho/64) <= NS
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    float* ed = sm_w;                                            // [NS][NRT][64] (used when !AREG)
    const int SWD = (2 * NS) | 1;                                // odd row stride >= 2*NS
    float* stg = ed + NS * NRT * 64 + (threadIdx.x >> 6) * (32 * SWD + 64);   // + 64 dump slots per wave
    if (!AREG)
        for (int i = threadIdx.x; i < NS * NRT * 64; i += 256) ed[i] = ED[i];
    for (int i = threadIdx.x; i < 4 * (32 * SWD + 64); i += 256) ed[NS * NRT * 64 + i] = 0.f;   // columns w >= Ho stay zero
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5;
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    const float inv_ho = 1.0f / (float)Ho;
    float areg[AREG ? NS : 1][AREG ? NRT : 1];
    if (AREG) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) areg[AREG ? s_ : 0][AREG ? rt : 0] = ED[(s_ * NRT + rt) * 64 + lane];
    }
    // A tile is 32*Ho consecutive floats of dY, plus (R-1)*P for every image boundary before the column.  Element
    // e = 64 i + lane is (column t = e / Ho, w = e % Ho) (exact through the float reciprocal for e < 4096).  All loads are
    // unconditional: columns past the end of the batch re-read the last valid column (finite values; those columns
    // of S' only ever meet the zero columns of A^T).
    auto load_tile = [&](long tile, float (&v)[NL]) {
        const long tl = tile < ntiles ? tile : ntiles - 1;
        const int m = (int)(tl / tiles_n);
        long n0 = (tl - (long)m * tiles_n) * 32;
        if (n0 >= NB) n0 = 0;
        const int c = m / R, r_ = m - c * R;
        const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
        const float* base = dY + (((long)c * B + b0) * R + r_) * P + (long)h0 * Ho;
        const int jump = (R - 1) * P;
        const int tlast = (int)(NB - n0 < 32 ? NB - n0 : 32) - 1;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = i * 64 + lane;
            const int t = (int)(((float)e + 0.5f) * inv_ho);
            const int tc = t < tlast ? t : tlast;
            const int k = (int)(((float)(h0 + tc) + 0.5f) * inv_ho);
            v[i] = base[e + (tc - t) * Ho + k * jump];
        }
    };
    float v[NL];
    load_tile((long)blockIdx.x * 4 + wave, v);
    for (int it = 0; it < iters; ++it) {
        const long tile = ((long)it * gridDim.x + blockIdx.x) * 4 + wave;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = i * 64 + lane;
            const int t = (int)(((float)e + 0.5f) * inv_ho);
            stg[t < 32 ? t * SWD + (e - t * Ho) : 32 * SWD + lane] = v[i];
        }
        __builtin_amdgcn_wave_barrier();
        float bq[NS];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) bq[s_] = stg[j * SWD + 2 * s_ + kh];
        __builtin_amdgcn_wave_barrier();
        load_tile(tile + (long)gridDim.x * 4, v);                // next tile: in flight during the MFMAs below
        f32x16 acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const float av = AREG ? areg[AREG ? s_ : 0][AREG ? rt : 0] : ed[(s_ * NRT + rt) * 64 + lane];
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[s_], acc[rt], 0, 0, 0);
            }
        if (tile < ntiles) {
            const int m = (int)(tile / tiles_n);
            const long n0 = (tile - (long)m * tiles_n) * 32;
            // row kk = 2 fx + ri = 32 rt + (r & 3) + 8 (r >> 2) + 4 kh: ri is a compile-time property of (rt, r)
            float* p0 = Sp + dft_t_off(n0 + j, m, 2 * M, Lh) + 2 * kh * 128;          // real rows, fx = fx0 + 2 kh
            float* p1 = Sp + dft_t_off(n0 + j, M + m, 2 * M, Lh) + 2 * kh * 128;      // imaginary rows
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk0 = 32 * rt + (r & 3) + 8 * (r >> 2);                 // kh = 0; kh = 1 adds 4
                    float* p = (kk0 & 1) ? p1 : p0;
                    const int fx0 = kk0 >> 1;
                    amx = fmaxf(amx, fabsf(acc[rt][r]));          // (rows beyond Lh come from zero table rows: 0)
                    if (LH2 > 0) {
                        if (kk0 + 4 < LH2) p[fx0 * 128] = acc[rt][r];
                        else if (kk0 < LH2) { if (kh == 0) p[fx0 * 128] = acc[rt][r]; }
                    } else {
                        if (fx0 + 2 * kh < Lh) p[fx0 * 128] = acc[rt][r];
                    }
                }
            h3_tile_flush_rd(amx, amax + m, lane);          // max |S'| of filter row m (rows m and M + m of the GEMM operand)
        }
    }
}

// ==========================================================================================
// Ring forms of the two transforms along w for the frames of the reference configurations (28x28: L = 44, Ho = 17;
// 64x64: L = 96, Ho = 33; the 50x50 MNIST-U / MNIST-N geometry: L = 66, Ho = 39).
//
// What bounded the register-staged kernels above was not the memory system but the compiler's in-order vmcnt
// bookkeeping (round-3 ISA reading): its loads of the NEXT tile and its stores of the CURRENT one share one counter,
// and the wait it places in front of the next tile's first use (a loop-header merge of "16 younger operations")
// makes every wave wait, once per tile, until most of its own output stores have been ACKNOWLEDGED (dft_dy_mf), or
// drains the whole queue right before the MFMA chain (dft_out_mf: `s_waitcnt vmcnt(0)` in front of the first MFMA,
// the next tile's loads sunk behind the chain).  Here every streamed operand arrives by `global_load_lds` DMAs that
// the compiler does not see, into a per-wave ring in LDS, with hand-counted `s_waitcnt vmcnt(N)` (N = the exact number
// of younger DMAs and stores, all issued unconditionally so that the count is uniform); there is no ordinary vector
// load in the loops, so the compiler places no vmcnt wait of its own, and the MFMA B operand is read straight from
// the ring (lane-linear: conflict free).  The constant operand lives in registers.
// ==========================================================================================
// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane BYTE offset (global "saddr" addressing): the
// per-lane offsets are loop invariants kept in registers, so issuing a piece costs no vector-ALU work at all.
// (nt on the 16-byte form: T is read exactly once, by dft_out_ring_kernel -- 884 -> 860 us; the dword form reads dY, which the
// launch before it has just written: nontemporal loads and stores both measured slower there)
#ifdef TVAE_T_CACHED
#define TVAE_DFT_DMA_X4(dst, off, base) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(off), "s"(base) : "memory", "m0")
#else
#define TVAE_DFT_DMA_X4(dst, off, base) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(dst), "v"(off), "s"(base) : "memory", "m0")
#endif
#define TVAE_DFT_DMA_X1(dst, off, base) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(off), "s"(base) : "memory", "m0")
#define TVAE_DFT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// out[c][b][r][h][w] = act(bias[c] + sum_k E[w][k] T[k][(m,n)]) -- the contraction over fx.  One wave owns a tile
// (filter row m, 32 columns n): its 2 * LHP frequency rows x 32 columns arrive as ND = ceil(LHP / 4) 1-KB DMAs (lane ->
// (fx & 3, re | im, 4 columns); the LDS image is [fx][re | im][32 columns] = the B-operand order of the K = 2 MFMA).
// Three slots per wave: while tile `it` is on the matrix pipe, tile it+1 is in flight and the pieces of tile it+2 are
// issued between the MFMAs into the slot tile it-1 has left; the consumed slot doubles as the transposition patch of the
// epilogue ([32 columns][HO], HO odd: its linear order IS the order of the tile's outputs in memory).  One 4-wave
// workgroup per CU (3 x 13 KB per wave at the 96-wide frame).  Per tile: ND DMAs, then SN stores; at the top of
// iteration it >= 2 the operations younger than tile it's DMAs are stores(it-2), DMAs(it+1), stores(it-1).
// T16 (round 4, one-part bf16 mode only): T holds 2-byte bf16 elements in the same element layout.  A DMA piece is then 16
// rows x 32 columns (lane -> row 16 g + (lane >> 2) = 2 fx + (re | im), columns 8 (lane & 3) ..), the LDS image
// [fx][re | im][32 columns] of bf16, and a lane widens its operand (one ds_read_u16 + shift) before the fp32 MFMA: half the
// bytes to fetch, seven DMAs per tile instead of thirteen at the 96-wide frame.
template <int LHP, int NT, bool REM1, int HO, bool TANH, bool T16 = false>
static __global__ __launch_bounds__(256, 1) void dft_out_ring_kernel(const float* __restrict__ T, const float* __restrict__ EO,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int M, int R, int B, int Lh, int act, float slope,
                                                                     float* __restrict__ amax) {
    constexpr int NTT = NT + (REM1 ? 1 : 0), ND = T16 ? (2 * LHP + 15) / 16 : (LHP + 3) / 4, SLOTB = ND * 1024, SLOTF = SLOTB / 4;
    float amx = 0.f;                                     // max |out| of the current tile -> amax[channel] (h3 scales of the launches that read out)
    constexpr int SN = (32 * HO + 63) / 64, P = HO * HO;
    static_assert((HO & 1) == 1, "odd output width: the [column][HO] patch is conflict free and linear in memory order");
    static_assert(32 * HO * 4 <= SLOTB, "the consumed slot must hold the transposition patch");
    static_assert(ND + 2 * SN <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    const int lane = threadIdx.x & 63, j = lane & 31, ri = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* ring = sm_w + wave * (3 * SLOTF);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm_w + (unsigned)(wave * 3 * SLOTB);
    float eo[LHP][NTT];
#pragma unroll
    for (int fx = 0; fx < LHP; ++fx)
#pragma unroll
        for (int t = 0; t < NTT; ++t) eo[fx][t] = EO[(fx * NTT + t) * 64 + lane];
    const long NB = (long)B * HO;
    const int tiles_n = (int)((NB + 31) / 32);           // n-tiles with at least one real column
    // A workgroup walks a CONTIGUOUS range of groups of four consecutive tiles (its four waves: the four 32-column tiles of one
    // 128-column block, whose 512-byte rows of T they read together), so a wave stays on one filter row for many tiles.
    const long ntiles = (long)M * tiles_n, stride = 4;
    const long per = ((ntiles + 3) / 4 + gridDim.x - 1) / gridDim.x;                   // groups per workgroup
    const long first = (long)blockIdx.x * per * 4 + wave;
    const long tend = min(ntiles, ((long)blockIdx.x + 1) * per * 4);
    if (first >= tend) return;                           // (no workgroup barrier anywhere in this kernel)
    const int my = (int)((tend - first + stride - 1) / stride);
    // DMA piece g, lane -> (fx = 4 g + (lane >> 4), re | im = (lane >> 3) & 1, columns 4 (lane & 7) ..): byte offset of the
    // lane's 16 bytes from the tile's (wave-uniform) base; rows beyond Lh (last piece) re-read the last row, never used
    unsigned doff[ND];
    if (T16) {
        const int c8 = (lane & 3) * 8;
#pragma unroll
        for (int g = 0; g < ND; ++g) {
            const int rr = 16 * g + (lane >> 2);         // row of the LDS image: 2 fx + (re | im)
            doff[g] = (unsigned)(((long)(rr & 1) * M * Lh + min(rr >> 1, Lh - 1)) * 128 + c8) * 2u;
        }
    } else {
        const int fxl = lane >> 4, rid = (lane >> 3) & 1, c4 = (lane & 7) * 4;
#pragma unroll
        for (int g = 0; g < ND; ++g)
            doff[g] = (unsigned)(((long)rid * M * Lh + min(4 * g + fxl, Lh - 1)) * 128 + c4) * 4u;
    }
    auto tile_mn = [&](int it, int& m, int& n0) {
        const int tl = (int)(first + (long)(it < my ? it : my - 1) * stride);   // past the end: the last tile again
        m = tl / tiles_n;                                // (host: fewer than 2^31 tiles, B * HO < 2^31)
        n0 = (tl - m * tiles_n) * 32;
    };
    auto dma_base = [&](int it) -> const float* {        // wave uniform: T + dft_t_off(n0, m, 2M, Lh)
        int m, n0;
        tile_mn(it, m, n0);
        const long eo_ = (((long)(n0 >> 7) * (2 * M) + m) * Lh) * 128 + (n0 & 127);      // element offset
        return T16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(T) + eo_) : T + eo_;
    };
    auto dma_piece = [&](const float* tb, int slot, int g) {
        const unsigned dst = ring_lds + (unsigned)(slot * SLOTB + g * 1024);
        TVAE_DFT_DMA_X4(dst, doff[g], tb);
    };
    {
        const float* t0 = dma_base(0);
#pragma unroll
        for (int g = 0; g < ND; ++g) dma_piece(t0, 0, g);
        const float* t1 = dma_base(1);
#pragma unroll
        for (int g = 0; g < ND; ++g) dma_piece(t1, 1, g);
    }
    const float sl = act == ACT_LRELU ? slope : 1.f;     // ACT_NONE: slope 1
    const int jump = (R - 1) * P;
    int slot = 0;
    int c_prev = -1;                                     // channel whose maximum `amx` is collecting
    for (int it = 0; it < my; ++it) {
        if (it == 0) TVAE_DFT_VMCNT(ND);
        else if (it == 1) TVAE_DFT_VMCNT(ND + SN);
        else TVAE_DFT_VMCNT(ND + 2 * SN);
        const float* tbn = dma_base(it + 2);
        const int slot2 = slot == 0 ? 2 : slot - 1;      // (it + 2) % 3: the slot tile it-1 has left
        const float* vs = ring + slot * SLOTF + lane;
        f32x16 acc[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][0][r] = acc[t][1][r] = 0.f;
        float racc = 0.f;
        const unsigned short* vs16 = reinterpret_cast<const unsigned short*>(ring + slot * SLOTF) + ri * 32 + j;
#pragma unroll
        for (int fx = 0; fx < LHP; ++fx) {
            const float v = T16 ? __uint_as_float((unsigned)vs16[fx * 64] << 16) : vs[fx * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t][fx & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(eo[fx][t], v, acc[t][fx & 1], 0, 0, 0);
            if (REM1) racc = __fmaf_rn(eo[fx][NTT - 1], v, racc);
            if (T16) {
                if ((fx & 7) == 1) dma_piece(tbn, slot2, fx >> 3);           // tile it+2, one piece per eight MFMA steps
            } else {
                if ((fx & 3) == 1) dma_piece(tbn, slot2, fx >> 2);           // tile it+2, one piece per four MFMA steps
            }
        }
#pragma unroll
        for (int g = 0; g < ND; ++g)
            if ((T16 ? 8 * g + 1 : 4 * g + 1) >= LHP) dma_piece(tbn, slot2, g);          // pieces the unrolled loop did not reach
        int m, n0;
        tile_mn(it, m, n0);
        const int c = m / R, r_ = m - c * R;
        if (c != c_prev && c_prev >= 0) h3_tile_flush(amx, amax + c_prev, lane);      // (wave uniform; a few times per wave)
        c_prev = c;
        const float bv = bias ? bias[c] : 0.f;
        float* stg = ring + slot * SLOTF;                // the slot just consumed: [32 columns][HO]
        float rtot = 0.f;
        if (REM1) rtot = racc + __shfl_xor(racc, 32, 64) + bv;
        if (TANH) {                                      // compile time: a run-time test is if-converted into BOTH forms per element
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][0][r] = tanhf(acc[t][0][r] + acc[t][1][r] + bv);
            rtot = tanhf(rtot);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = acc[t][0][r] + acc[t][1][r] + bv;
                    acc[t][0][r] = x > 0.f ? x : x * sl;
                }
            rtot = rtot > 0.f ? rtot : rtot * sl;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * ri;
                if (32 * t + 31 < HO || w < HO) stg[j * HO + w] = acc[t][0][r];
            }
        if (REM1) {
            if (ri == 0) stg[j * HO + 32 * NT] = rtot;
        }
        __builtin_amdgcn_wave_barrier();
        // the 32 output rows of the tile are 32*HO consecutive floats of out, plus (R-1)*P for every image boundary
        // before the row: element e belongs to the next image iff e >= (HO - h0) * HO (a second boundary, HO rows on, only
        // when HO < 32).  Exactly SN store instructions per tile: lanes past the end repeat the last element.
        const int b0 = n0 / HO, h0 = n0 - b0 * HO;
        float* obase = out + ((((long)c * B + b0) * R + r_) * P + (long)h0 * HO);      // wave uniform
        const int cnt = (int)(NB - n0 < 32 ? NB - n0 : 32) * HO;
        const int thr = (HO - h0) * HO;
#pragma unroll
        for (int i = 0; i < SN; ++i) {
            const int e = min(i * 64 + lane, cnt - 1);
            int o = e + (e >= thr ? jump : 0);
            if (HO < 32) o += e >= thr + P ? jump : 0;   // narrow outputs: 32 rows can span three images
            const float sv = stg[e];
            amx = fmaxf(amx, fabsf(sv));
            __builtin_nontemporal_store(sv, obase + (unsigned)o);          // written once, read by later launches
        }
        __builtin_amdgcn_wave_barrier();
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (c_prev >= 0) h3_tile_flush(amx, amax + c_prev, lane);
    TVAE_DFT_VMCNT(0);                                   // the clamped tail DMAs still target this wave's ring
}

// number of store instructions dft_dy_ring_kernel issues per tile (rows 32 rt + (r & 3) + 8 (r >> 2) [+ 4] below LH2)
constexpr int dft_dy_ring_stores(int NRT, int LH2, bool NYQ) {
    int n = NYQ ? 1 : 0;
    for (int rt = 0; rt < NRT; ++rt)
        for (int r = 0; r < 16; ++r)
            if (32 * rt + (r & 3) + 8 * (r >> 2) < LH2) ++n;
    return n;
}

// S'[(fx,ri)][(m,n)] = sum_w E'[(fx,ri)][w] dY[(m,n)][w] -- the DFT over w of the output gradient.  The 32 x HO values
// of a tile are (mostly) one contiguous run of dY: NL = ceil(32 HO / 64) dword DMAs bring it into the wave's slot in
// exactly that order ([column][w], row pitch HO: odd, so the B-operand reads column (lane & 31), w = 2s + (lane >> 5)
// are conflict free) -- no register staging, no transposition.  Two slots per wave; the DMAs of tile it+1 are issued
// between the MFMAs of tile it, and the only operations younger than them at the next wait are this tile's ST stores.
// NYQ (2 Lh = 32 NRT + 2, even frame): the Nyquist row is an alternating sum on the vector ALU (its sine row is zero),
// one store for both; without it those two rows would cost a whole fourth tile of MFMAs (17 of 68 at the 96-wide frame).
// S16 (round 4, one-part bf16 mode only): S' is STORED as bf16 -- the same element layout, 2-byte elements, rounded (RNE) as
// it leaves the accumulators; the weight-gradient GEMM then DMAs ready-made one-part cells (dense_wgrad_x6_dma_kernel<.., ABF>).
template <int NS, int NRT, int LH2, int HO, bool NYQ, bool S16 = false>
static __global__ __launch_bounds__(256, 2) void dft_dy_ring_kernel(const float* __restrict__ dY, const float* __restrict__ ED,
                                                                    float* __restrict__ Sp, int M, int R, int B, int Lh,
                                                                    long NBpad, float* __restrict__ amax) {
    constexpr int NL = (32 * HO + 63) / 64, SLOTB = NL * 256, SLOTF = NL * 64, P = HO * HO;
    constexpr int ST = dft_dy_ring_stores(NRT, LH2, NYQ);
    static_assert((HO & 1) == 1 && 2 * NS <= HO + 1 && 2 * NS >= HO, "odd output width, NS = ceil(HO / 2)");
    static_assert(ST <= 63, "vmcnt is a 6-bit counter");
    static_assert(!NYQ || LH2 == 32 * NRT + 2, "Nyquist rows on the vector ALU");
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    const int lane = threadIdx.x & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* ring = sm_w + wave * (2 * SLOTF);
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm_w + (unsigned)(wave * 2 * SLOTB);
    float areg[NS][NRT];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) areg[s_][rt] = ED[(s_ * NRT + rt) * 64 + lane];
    const int tiles_n = (int)(NBpad / 32);
    const long ntiles = (long)M * tiles_n, NB = (long)B * HO;
    // contiguous ranges of groups of four consecutive tiles per workgroup (see dft_out_ring_kernel)
    const long stride = 4;
    const long per = ((ntiles + 3) / 4 + gridDim.x - 1) / gridDim.x;
    const long first = (long)blockIdx.x * per * 4 + wave;
    const long tend = min(ntiles, ((long)blockIdx.x + 1) * per * 4);
    if (first >= tend) return;
    const int my = (int)((tend - first + stride - 1) / stride);
    const int jump = (R - 1) * P;
    auto tile_mn = [&](int it, int& m, int& n0) {
        const int tl = (int)(first + (long)(it < my ? it : my - 1) * stride);
        m = tl / tiles_n;
        n0 = (tl - m * tiles_n) * 32;
    };
    // The NL DMAs of a tile: element e = 64 i + lane is (column t = e / HO, w = e % HO) and sits e floats behind the
    // tile's first element, plus (R-1)*P once the column belongs to the next image (e >= (HO - h0) HO).  A ragged last
    // tile re-reads its last valid column for the columns past the end of the batch (finite values; those columns of S'
    // only ever meet zero columns of A^T); a tile of pure padding columns reads tile 0 of its row.
    struct Src { const float* base; int tlast, thr; };
    auto dma_src = [&](int it) {
        int m, n0;
        tile_mn(it, m, n0);
        if (n0 >= NB) n0 = 0;
        const int c = m / R, r_ = m - c * R;
        const int b0 = n0 / HO, h0 = n0 - b0 * HO;
        Src q;
        q.base = dY + ((((long)c * B + b0) * R + r_) * P + (long)h0 * HO);              // wave uniform
        q.tlast = (int)(NB - n0 < 32 ? NB - n0 : 32) - 1;
        q.thr = (HO - h0) * HO;
        return q;
    };
    auto dma_piece = [&](const Src& q, int slot, int i) {
        int e = i * 64 + lane;
        // the last piece overhangs the tile when 32 HO is not a multiple of 64; element 32 HO IS consumed (column 31 at
        // the padded step w = HO, against a zero of the constant operand), so it must be finite and inside dY: the
        // overhang re-reads the last valid column as well (found by the NaN guard bands of
        // tests/test_hip_modules.py::test_step_does_not_read_out_of_bounds[M28]: the last tile of the last filter read
        // 128 bytes past the end of dY)
        if (q.tlast < 31 || (i == NL - 1 && (32 * HO) % 64 != 0)) {      // first: wave uniform, rare; second: static
            const int t = e / HO;
            if (t > q.tlast) e -= (t - q.tlast) * HO;
        }
        int o = e + (e >= q.thr ? jump : 0);
        if (HO < 32) o += e >= q.thr + P ? jump : 0;     // narrow outputs: 32 columns can span three images
        const unsigned off = (unsigned)o * 4u;
        const unsigned dst = ring_lds + (unsigned)(slot * SLOTB + i * 256);
        TVAE_DFT_DMA_X1(dst, off, q.base);
    };
    {
        const Src q0 = dma_src(0);
#pragma unroll
        for (int i = 0; i < NL; ++i) dma_piece(q0, 0, i);
    }
    // store offsets (floats from Sp + dft_t_off(n0, m, 2M, Lh), wave uniform): real rows at j + 2 kh 128, imaginary rows
    // M * Lh * 128 further on
    const unsigned lre = (unsigned)(j + 2 * kh * 128), lim = lre + (unsigned)((long)M * Lh * 128);
    float mx = 0.f;                                      // h3 arithmetic of the GEMM that follows: max |S'| of the tile's filter row -> amax[M]
    int m_prev = -1;                                     // filter row whose maximum `mx` is collecting
    for (int it = 0; it < my; ++it) {
        const int slot = it & 1;
        if (it == 0) TVAE_DFT_VMCNT(0);
        else TVAE_DFT_VMCNT(ST);
        const Src qn = dma_src(it + 1);
        const float* bs = ring + slot * SLOTF + j * HO + kh;
        f32x16 acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        float racc = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float bq = bs[2 * s_];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s_][rt], bq, acc[rt], 0, 0, 0);
            if (NYQ) {                                   // (-1)^w, w = 2 s + kh; the last step's kh = 1 half is w = HO: outside
                if (2 * s_ + 1 < HO) racc += kh ? -bq : bq;
                else racc += kh ? 0.f : bq;
            }
            if (s_ < NL) dma_piece(qn, slot ^ 1, s_);    // tile it+1, one piece per k-step
        }
#pragma unroll
        for (int i = NS; i < NL; ++i) dma_piece(qn, slot ^ 1, i);
        int m, n0;
        tile_mn(it, m, n0);
        if (m != m_prev && m_prev >= 0) h3_tile_flush(mx, amax + m_prev, lane);       // (wave uniform; a few times per wave)
        m_prev = m;
        const long sbo = (((long)(n0 >> 7) * (2 * M) + m) * Lh) * 128 + (n0 & 127);     // element offset (wave uniform)
        float* sb = Sp + sbo;
        unsigned short* sb16 = reinterpret_cast<unsigned short*>(Sp) + sbo;
        auto put = [&](unsigned o, float v) {            // one store instruction either way (the waits count them)
            if (S16) sb16[o] = __builtin_bit_cast(unsigned short, (__bf16)v);
            else sb[o] = v;
        };
        // row kk = 2 fx + ri = 32 rt + (r & 3) + 8 (r >> 2) + 4 kh: ri is a compile-time property of (rt, r)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk0 = 32 * rt + (r & 3) + 8 * (r >> 2);             // kh = 0; kh = 1 adds 4
                const unsigned o = ((kk0 & 1) ? lim : lre) + (unsigned)((kk0 >> 1) * 128);
                if (kk0 + 4 < LH2) put(o, acc[rt][r]);
                else if (kk0 < LH2) { if (kh == 0) put(o, acc[rt][r]); }
                if (kk0 < LH2) mx = fmaxf(mx, fabsf(acc[rt][r]));      // (a few values beyond LH2 in the kh = 1 half: zero table rows)
            }
        if (NYQ) {                                       // fx = L/2: cosine row = the alternating sum, sine row = 0
            const float tot = racc + __shfl_xor(racc, 32, 64);
            const unsigned o = (kh ? lim : lre) - (unsigned)(2 * kh * 128) + (unsigned)((LH2 / 2 - 1) * 128);
            put(o, kh ? 0.f : tot);
            mx = fmaxf(mx, fabsf(tot));
        }
    }
    if (m_prev >= 0) h3_tile_flush(mx, amax + m_prev, lane);
    TVAE_DFT_VMCNT(0);
}

// ------------------------------------------------------------------------------------------
// Generic forms of the two transforms along w for frames the specialised instances above do not cover (Lh > 64 or
// Ho > 64: the 192-wide frame of the galaxy configuration has Lh = 97, Ho = 129).  Same tile walk (one wave owns 32
// columns of one filter row), same tables (dft_wtab_kernel with NT = ceil(Ho/32) whole 32-row tiles, zero rows beyond
// Ho), but nothing is sized at compile time except the number of output tiles: the constant operand is read from
// global memory (a 256-byte line per MFMA, L1 / L2 resident: one load per 64-cycle matrix instruction), the values
// arrive in chunks of eight frequencies, and the output tile goes through a 32 x 33 LDS patch per 32 rows of w.
// ------------------------------------------------------------------------------------------
template <int NT>
static __global__ __launch_bounds__(256) void dft_out_gen_kernel(const float* __restrict__ T, const float* __restrict__ EO,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int M, int R, int B, int Ho, int Lh, long NBpad, int act,
                                                                 float slope, float* __restrict__ amax) {
    float amx = 0.f;
    __shared__ float stg_all[4 * 32 * 33];
    float* stg = stg_all + (threadIdx.x >> 6) * 32 * 33;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, ri = lane >> 5;
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        if (n0 >= NB) continue;                          // a tile of pure padding columns
        const float* tp = T + dft_t_off(n0 + j, ri * M + m, 2 * M, Lh);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float vn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vn[e] = tp[(long)min(e, Lh - 1) * 128];
        for (int fx0 = 0; fx0 < Lh; fx0 += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vn[e];
#pragma unroll
            for (int e = 0; e < 8; ++e) vn[e] = tp[(long)min(fx0 + 8 + e, Lh - 1) * 128];     // next chunk in flight
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int fx = fx0 + e;
                if (fx < Lh) {                           // wave-uniform
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(EO[((long)fx * NT + t) * 64 + lane], v[e], acc[t], 0, 0, 0);
                }
            }
        }
        const int c = m / R, r_ = m - c * R;
        const float bv = bias ? bias[c] : 0.f;
        const int tmax = (int)(NB - n0 < 32 ? NB - n0 : 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // rows w = 32 t + (r & 3) + 8 (r >> 2) + 4 ri of column j -> stg[j][w - 32 t]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[t][r] + bv;
                if (act == ACT_LRELU) x = x > 0.f ? x : x * slope;
                else if (act == ACT_TANH) x = tanhf(x);
                stg[j * 33 + (r & 3) + 8 * (r >> 2) + 4 * ri] = x;
            }
            __builtin_amdgcn_wave_barrier();
            const int wn = min(32, Ho - 32 * t);         // valid rows of this w-tile
            for (int e = lane; e < tmax * 32; e += 64) {
                const int col = e >> 5, w = e & 31;
                if (w < wn) {
                    const long nn = n0 + col;
                    const int b = (int)(nn / Ho), h = (int)(nn - (long)b * Ho);
                    const float sv = stg[col * 33 + w];
                    amx = fmaxf(amx, fabsf(sv));
                    out[(((long)c * B + b) * R + r_) * P + (long)h * Ho + 32 * t + w] = sv;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        h3_tile_flush_rd(amx, amax + c, lane);              // max |out| of channel c
    }
}

// S'[(fx,ri)][(m,n)] = sum_w E'[(fx,ri)][w] dY[(m,n)][w], generic frame: the 32 x Ho values of a tile go through this
// wave's LDS patch ([32 columns][2*NS | 1]) into the B-operand layout; one 32-row tile of (fx, ri) at a time.
static __global__ __launch_bounds__(256) void dft_dy_gen_kernel(const float* __restrict__ dY, const float* __restrict__ ED,
                                                                float* __restrict__ Sp, int M, int R, int B, int Ho, int Lh,
                                                                long NBpad, int NS, int NRT, float* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    float amx = 0.f;
    const int SWD = (2 * NS) | 1;
    float* stg = sm_w + (threadIdx.x >> 6) * (32 * SWD);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5;
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const int c = m / R, r_ = m - c * R;
        // stage the tile: column t (n = n0 + t) is Ho consecutive floats of dY; columns past the batch are zero
        for (int e = lane; e < 32 * SWD; e += 64) {
            const int t = e / SWD, w = e - t * SWD;
            const long nn = n0 + t;
            float v = 0.f;
            if (nn < NB && w < Ho) {
                const int b = (int)(nn / Ho), h = (int)(nn - (long)b * Ho);
                v = dY[(((long)c * B + b) * R + r_) * P + (long)h * Ho + w];
            }
            stg[e] = v;
        }
        __builtin_amdgcn_wave_barrier();
        float* p0 = Sp + dft_t_off(n0 + j, m, 2 * M, Lh);
        float* p1 = Sp + dft_t_off(n0 + j, M + m, 2 * M, Lh);
        for (int rt = 0; rt < NRT; ++rt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int s_ = 0; s_ < NS; ++s_)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ED[((long)s_ * NRT + rt) * 64 + lane], stg[j * SWD + 2 * s_ + kh],
                                                           acc, 0, 0, 0);
            // row kk = 2 fx + ri = 32 rt + (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int fx = kk >> 1;
                if (fx < Lh) ((kk & 1) ? p1 : p0)[(long)fx * 128] = acc[r];
                amx = fmaxf(amx, fabsf(acc[r]));
            }
        }
        h3_tile_flush_rd(amx, amax + m, lane);              // max |S'| of filter row m
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Round 5: "wide" forms of the two generic transforms for LARGE frames (galaxy shape: L = 160, Lh = 81, Ho = 129), where the
// kernels above are latency bound (0.6 / 1.0 TB/s: a dependent global load of the constant operand per matrix instruction,
// integer divisions in a serial staging loop, one wave per tile).  Here a WORKGROUP owns a tile (filter row m, 32 columns) and
// its waves split the OUTPUT rows: wave q holds the constant operand of ITS 32 output rows in registers for the whole kernel
// (Ho / 2 resp. Lh values per lane) and all waves read the tile's streamed values from one shared LDS image, staged by all
// threads together with coalesced loads one tile ahead (two slots, one barrier per tile).  Plain loads and __syncthreads:
// nothing is hand counted.  Workgroups walk contiguous tile ranges (a filter row stays with a workgroup for many tiles).
// ------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS only: __syncthreads() also drains the vector-memory counter, i.e. every tile would wait for
// the acknowledgement of its own output stores
__device__ __forceinline__ void dft_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int DFT_WIDE_NS = 80;        // k-steps (pairs of w) the dY form keeps in registers: Ho <= 160
constexpr int DFT_WIDE_LH = 96;        // frequencies the out form keeps in registers: L <= 190

// S'[(fx,ri)][(m,n)] = sum_w E'[(fx,ri)][w] dY[(m,n)][w]; blockDim = 64 * NRT (wave rt = 32-row tile of (fx, ri) pairs).
// LDS: two slots of 32 * PW + 1 floats, PW = Ho | 1 (odd pitch: conflict-free operand reads).  NSR >= NS: register array size.
template <int NSR, int NLD, int WPE>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void dft_dy_wide_kernel(const float* __restrict__ dY, const float* __restrict__ ED,
                                                                 float* __restrict__ Sp, int M, int R, int B, int Ho, int Lh,
                                                                 long NBpad, int NS, int NRT, float* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    const int lane = threadIdx.x & 63, j = lane & 31, kh = lane >> 5;
    const int rt = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (slot: the tile's 32 columns at pitch PW, then 2 NSR zeros: the chain below runs NSR steps unconditionally -- steps beyond
    //  NS meet a zero of the constant operand and read finite padding; a run-time guard per matrix instruction would put every
    //  one of them, with its LDS read, into a basic block of its own and serialise the read latencies)
    const int PW = Ho | 1, SLOT = 32 * PW + 2 * NSR + 2, nthr = blockDim.x;
    float areg[NSR];
#pragma unroll
    for (int s_ = 0; s_ < NSR; ++s_) areg[s_] = s_ < NS ? ED[((long)s_ * NRT + rt) * 64 + lane] : 0.f;
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    const long jump = (long)(R - 1) * P;
    const long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long t_beg = (long)blockIdx.x * per, t_end = min(ntiles, t_beg + per);
    if (t_beg >= t_end) return;
    // NLD staged values per thread and tile (host: 32 Ho <= NLD * 64 NRT)
    const int nel = 32 * Ho;
    const int nld = (nel + nthr - 1) / nthr;             // ... of which this many are real (wave uniform)
    float stage[NLD];
    // element e = i * nthr + tid of a tile is the same (column, w) in every tile: its destination offset is computed ONCE (an
    // integer division per element and tile in the staging code cost as many issue cycles as the tile's matrix instructions)
    int dsto[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = i * nthr + (int)threadIdx.x;
        dsto[i] = e + (e / Ho) * (PW - Ho);
    }
    auto tile_load = [&](long tile) {                    // the tile's 32 * Ho values into registers (zeros past the batch)
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const int c = m / R, r_ = m - c * R;
        const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
        const float* base = dY + ((((long)c * B + b0) * R + r_) * P + (long)h0 * Ho);
        const long cnt = n0 < NB ? (NB - n0 < 32 ? NB - n0 : 32) * Ho : 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (i < nld) {
                const int e = i * nthr + (int)threadIdx.x;
                float v = 0.f;
                if (e < nel && e < cnt) {
                    const int x = h0 * Ho + e;           // image boundaries before this element: at most one when Ho >= 32
                    const int nb = Ho >= 32 ? (x >= P ? 1 : 0) : x / P;
                    v = base[e + nb * jump];
                }
                stage[i] = v;
            }
        }
    };
    auto tile_put = [&](int slot) {
        float* dst = sm_w + slot * SLOT;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (i < nld) {
                const int e = i * nthr + (int)threadIdx.x;
                if (e < nel) dst[dsto[i]] = stage[i];
            }
        }
    };                                                   // (index 32 PW and the pad element of every column are never written)
    // (the pad element of each column row -- w = Ho when Ho is even -- only ever meets a zero of the constant operand, but it
    //  must be finite: zero the slots once)
    for (int i = threadIdx.x; i < 2 * SLOT; i += nthr) sm_w[i] = 0.f;
    __syncthreads();
    tile_load(t_beg);
    tile_put(0);
    __syncthreads();
    float mx = 0.f;
    int m_prev = -1;
    for (long tile = t_beg; tile < t_end; ++tile) {
        const int slot = (int)((tile - t_beg) & 1);
        if (tile + 1 < t_end) tile_load(tile + 1);       // in flight under this tile's matrix instructions
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const float* bs = sm_w + slot * SLOT + j * PW + kh;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < NSR; s_ += 2) {            // two independent chains, no branches
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s_], bs[2 * s_], acc0, 0, 0, 0);
            if (s_ + 1 < NSR) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s_ + 1 < NSR ? s_ + 1 : s_], bs[2 * s_ + 2], acc1, 0, 0, 0);
        }
        if (m != m_prev && m_prev >= 0) h3_tile_flush_rd(mx, amax + m_prev, lane);
        m_prev = m;
        // the next tile into its slot BEFORE this tile's stores are issued: the wait for its loads then covers only operations
        // older than them (slot ^ 1 was last read before the previous barrier)
        if (tile + 1 < t_end) tile_put(slot ^ 1);
        float* p0 = Sp + dft_t_off(n0 + j, m, 2 * M, Lh);
        float* p1 = Sp + dft_t_off(n0 + j, M + m, 2 * M, Lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {                   // row kk = 2 fx + ri = 32 rt + (r & 3) + 8 (r >> 2) + 4 kh
            const int kk = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int fx = kk >> 1;
            const float v = acc0[r] + acc1[r];
            if (fx < Lh) {
                ((kk & 1) ? p1 : p0)[(long)fx * 128] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        }
        dft_lds_barrier();
    }
    if (m_prev >= 0) h3_tile_flush_rd(mx, amax + m_prev, lane);
}

// out[c][b][r][h][w] = act(bias[c] + sum_k E[w][k] T[k][(m,n)]); blockDim = 64 * NTW: wave t = 32 output columns w on the
// matrix pipe; with XROW (Ho = 32 NTW + 1: 129 at the galaxy shape) the single last column is a dot product on the vector ALU
// of wave 0, its table row taken from tile NTW of EO (NT = NTW + 1 tiles were tabulated).
// LDS: two slots of Lh * 64 floats ([fx][re | im][32 columns]: the B-operand order), one 32 x 33 patch per wave, Lh * 2 floats.
template <int LHR, int NLD>
static __global__ __launch_bounds__(512) void dft_out_wide_kernel(const float* __restrict__ T, const float* __restrict__ EO,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int M, int R, int B, int Ho, int Lh, long NBpad, int NT,
                                                                  int NTW, int act, float slope, float* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    const int lane = threadIdx.x & 63, j = lane & 31, ri = lane >> 5;
    const int wt = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int SLOT = LHR * 64, nthr = blockDim.x;        // (rows Lh .. LHR - 1 stay zero: the chain runs LHR steps unconditionally)
    const bool xrow = NT > NTW;                          // one extra output column w = 32 NTW
    float* patch = sm_w + 2 * SLOT + wt * (32 * 33);
    float* ex = sm_w + 2 * SLOT + NTW * (32 * 33);       // [fx][ri]: E[w = 32 NTW][(fx, ri)]
    float eo[LHR];
#pragma unroll
    for (int fx = 0; fx < LHR; ++fx) eo[fx] = fx < Lh ? EO[((long)fx * NT + wt) * 64 + lane] : 0.f;
    if (xrow)
        for (int i = threadIdx.x; i < 2 * Lh; i += nthr) ex[i] = EO[((long)(i >> 1) * NT + NTW) * 64 + (i & 1) * 32];
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    const long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long t_beg = (long)blockIdx.x * per, t_end = min(ntiles, t_beg + per);
    if (t_beg >= t_end) return;
    // NLD float4 pieces per thread and tile (host: 16 Lh <= NLD * 64 NTW; a compile-time count keeps `stage` in registers)
    const int npc = Lh * 16;                             // pieces of a tile: (fx, ri) rows x 8 float4
    float4 stage[NLD];
    auto tile_load = [&](long tile) __attribute__((always_inline)) {
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const float* t0 = T + dft_t_off(n0, m, 2 * M, Lh);
        const float* t1 = T + dft_t_off(n0, M + m, 2 * M, Lh);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int pc = min(i * nthr + (int)threadIdx.x, npc - 1);        // (past the end: the last piece again)
            const int q4 = pc & 7, rr = pc >> 3, fx = rr >> 1;
            stage[i] = *reinterpret_cast<const float4*>(((rr & 1) ? t1 : t0) + (long)fx * 128 + 4 * q4);
        }
    };
    auto tile_put = [&](int slot) __attribute__((always_inline)) {
        float4* dst = reinterpret_cast<float4*>(sm_w + slot * SLOT);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int pc = i * nthr + (int)threadIdx.x;
            if (pc < npc) dst[pc] = stage[i];            // piece pc = ((fx * 2 + ri) * 8 + q4): [fx][ri][32] floats, linear
        }
    };
    for (int i = threadIdx.x; i < 2 * SLOT; i += nthr) sm_w[i] = 0.f;
    __syncthreads();
    tile_load(t_beg);
    tile_put(0);
    __syncthreads();
    float amx = 0.f;
    int c_prev = -1;
    for (long tile = t_beg; tile < t_end; ++tile) {
        const int slot = (int)((tile - t_beg) & 1);
        if (tile + 1 < t_end) tile_load(tile + 1);
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const float* vs = sm_w + slot * SLOT + lane;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int fx = 0; fx < LHR; fx += 2) {            // two independent chains, no branches
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(eo[fx], vs[fx * 64], acc0, 0, 0, 0);
            if (fx + 1 < LHR) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(eo[fx + 1 < LHR ? fx + 1 : fx], vs[(fx + 1) * 64], acc1, 0, 0, 0);
        }
        float racc = 0.f;
        if (xrow && wt == 0) {                           // the single last column on the vector ALU (wave uniform)
            for (int fx = 0; fx < Lh; ++fx) racc = __fmaf_rn(ex[2 * fx + ri], vs[fx * 64], racc);
            racc += __shfl_xor(racc, 32, 64);
        }
        const int c = m / R, r_ = m - c * R;
        if (c != c_prev && c_prev >= 0) h3_tile_flush_rd(amx, amax + c_prev, lane);
        c_prev = c;
        if (tile + 1 < t_end) tile_put(slot ^ 1);        // before this tile's output stores (see dft_dy_wide_kernel)
        if (n0 < NB) {                                   // (a tile of pure padding columns has no outputs)
            const float bv = bias ? bias[c] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc0[r] + acc1[r] + bv;
                if (act == ACT_LRELU) x = x > 0.f ? x : x * slope;
                else if (act == ACT_TANH) x = tanhf(x);
                patch[j * 33 + (r & 3) + 8 * (r >> 2) + 4 * ri] = x;
            }
            __builtin_amdgcn_wave_barrier();
            const int tmax = (int)(NB - n0 < 32 ? NB - n0 : 32);
            const int wn = min(32, Ho - 32 * wt);        // valid output columns of this wave's tile
            const int w = lane & 31;
            const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
            float* obase = out + (((long)c * B + b0) * R + r_) * P + (long)h0 * Ho + 32 * wt + w;
            const long jump = (long)(R - 1) * P;
#pragma unroll 4
            for (int col = lane >> 5; col < tmax; col += 2) {
                if (w < wn) {
                    const int nb = Ho >= 32 ? (h0 + col >= Ho ? 1 : 0) : (h0 + col) / Ho;      // image boundaries before this column
                    const float sv = patch[col * 33 + w];
                    amx = fmaxf(amx, fabsf(sv));
                    obase[(long)col * Ho + nb * jump] = sv;
                }
            }
            if (xrow && wt == 0 && ri == 0 && j < tmax) {
                float x = racc + bv;
                if (act == ACT_LRELU) x = x > 0.f ? x : x * slope;
                else if (act == ACT_TANH) x = tanhf(x);
                const int nb = Ho >= 32 ? (h0 + j >= Ho ? 1 : 0) : (h0 + j) / Ho;
                amx = fmaxf(amx, fabsf(x));
                out[(((long)c * B + b0) * R + r_) * P + (long)h0 * Ho + (long)j * Ho + nb * jump + 32 * NTW] = x;
            }
            __builtin_amdgcn_wave_barrier();
        }
        dft_lds_barrier();
    }
    if (c_prev >= 0) h3_tile_flush_rd(amx, amax + c_prev, lane);
}

// Bias gradient of the lifting convolution for free: the fx = 0 real row of S' is sum_w dY[m][n][w], so
// db[c] = sum_{r, n} S'[c*R + r][fx = 0][n]  (reads M*NB floats instead of a pass over dY).  Two stages: one workgroup per
// filter row m (1 024 of them: the single-stage version had 128 workgroups walking 51 MB-strided runs), then R sums.
static __global__ void dft_dbias_rows_kernel(const float* __restrict__ Sp, float* __restrict__ part, int Lh, long NB, int M,
                                             int s16) {
    __shared__ float sm[16];
    const int m = blockIdx.x;
    float acc[1] = {0.f};
    if (s16) {           // S' stored as bf16 (one-part mode)
        const unsigned short* S16p = reinterpret_cast<const unsigned short*>(Sp);
        for (long n = threadIdx.x; n < NB; n += blockDim.x)
            acc[0] += __uint_as_float((unsigned)S16p[dft_t_off(n, m, 2 * M, Lh)] << 16);
    } else
    for (long n = threadIdx.x; n < NB; n += blockDim.x) acc[0] += Sp[dft_t_off(n, m, 2 * M, Lh)];
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) part[m] = acc[0];
}
static __global__ void dft_dbias_kernel(const float* __restrict__ part, float* __restrict__ db, int R, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += part[c * R + r];
    db[c] = s;
}

// ------------------------------------------------------------------------------------------
// One workgroup per (filter m, input channel ci):  dKh'[fy][fx] from the four real blocks of G[fx][m | M+m][k],
//   Re = G[m][(ci,0,fy)] + G[M+m][(ci,1,fy)],  Im = G[m][(ci,1,fy)] - G[M+m][(ci,0,fy)],
// then dbank[m][ci][u][v] = 1/L^2 sum_fx c_fx Re( e^{2 pi i fx v/L} sum_fy dKh'[fy][fx] e^{2 pi i fy u/L} ),  u, v < ksz,
// accumulated over blocks of FXB frequencies (the plane spectrum of a 192-wide frame does not fit LDS at once).
// G arrives as `nsl` split-K slabs (slab stride `gs` floats) of the weight-gradient GEMM; they are summed, in slab order,
// while the filter's rows are loaded -- the spectral gradient is never finalised into a tensor of its own.
// LDS: Kh L*FXB complex, Z ksz*FXB complex, tw L complex, out ksz*ksz floats.
// ------------------------------------------------------------------------------------------
static __global__ void dft_dbank_kernel(const float* __restrict__ G, int nsl, long gs, float* __restrict__ dbank, int ksz,
                                        int L, int Lh, int M, int Cin, int FXB) {
    extern __shared__ float sm_dft[];
    float2* Kh = reinterpret_cast<float2*>(sm_dft);
    float2* Z = Kh + L * FXB;
    float2* tw = Z + ksz * FXB;
    float* outp = reinterpret_cast<float*>(tw + L);
    const int m = blockIdx.x / Cin, ci = blockIdx.x - m * Cin;
    fill_twiddles(tw, L);
    for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) outp[i] = 0.f;
    const long rowlen = 2L * L * Cin;
    for (int fx0 = 0; fx0 < Lh; fx0 += FXB) {
        const int nfx = min(FXB, Lh - fx0);
        __syncthreads();
        for (int i = threadIdx.x; i < nfx * L; i += blockDim.x) {
            const int fy = i % L, f = i / L;
            const float* r0 = G + ((long)(fx0 + f) * 2 * M + m) * rowlen + (long)(2 * ci) * L;
            const float* r1 = G + ((long)(fx0 + f) * 2 * M + M + m) * rowlen + (long)(2 * ci) * L;
            float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
            for (int sl = 0; sl < nsl; ++sl) {
                a += r0[sl * gs + fy];
                b += r1[sl * gs + L + fy];
                c += r0[sl * gs + L + fy];
                d += r1[sl * gs + fy];
            }
            Kh[fy * FXB + f] = make_float2(a + b, c - d);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ksz * nfx; i += blockDim.x) {
            const int u = i / nfx, f = i - u * nfx;
            float2 acc = make_float2(0.f, 0.f);
            int ph = 0;
#pragma unroll 4
            for (int fy = 0; fy < L; ++fy) {
                const float2 p = cmul(Kh[fy * FXB + f], tw[ph]);
                acc.x += p.x;
                acc.y += p.y;
                ph += u;
                if (ph >= L) ph -= L;
            }
            Z[u * FXB + f] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) {
            const int u = i / ksz, v = i - u * ksz;
            float acc = 0.f;
            int ph = (int)(((long)fx0 * v) % L);
            for (int f = 0; f < nfx; ++f) {
                const int fx = fx0 + f;
                const float cf = ((fx == 0) || (2 * fx == L)) ? 1.f : 2.f;
                const float2 z = Z[u * FXB + f];
                acc += cf * (z.x * tw[ph].x - z.y * tw[ph].y);
                ph += v;
                if (ph >= L) ph -= L;
            }
            outp[i] += acc;                              // thread i owns element i in every block: no race
        }
    }
    const float inv = 1.f / ((float)L * (float)L);
    for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) dbank[((long)m * Cin + ci) * ksz * ksz + i] = outp[i] * inv;
}

// ------------------------------------------------------------------------------------------
// The same inverse transform with both contractions on the fp32 matrix pipe (round 4; ksz <= 64, >= 17 frequencies per
// block).  The direct sums above spend ~18 000 vector instructions per thread on 1.6 M complex multiply-adds per filter;
// as two small real GEMMs with twiddle operands they are 242 v_mfma_f32_32x32x2_f32 per wave:
//   stage 1   [Zr; Zi][u][f] = sum_fy [c, -s; s, c](fy u) [Kr; Ki][fy][f]       rows u (32 per tile), columns f, k = fy pairs
//   stage 2   out[u][v]     += sum_fx c_fx ( Zr[u][fx] cos(fx v) - Zi[u][fx] sin(fx v) )   rows u, columns v, k = fx pairs
// Four waves, one 32 x 32 output tile each ((u tile, f tile) resp. (u tile, v tile)).  The twiddle operand of a lane is
// tw[(fy u) mod L] with a running phase (u resp. v fixed per lane, +2u / +2v per k-step): one 8-byte LDS read per step and
// operand, next to one of the data (Kh[fy][f] = (Kr, Ki), Z[u][fx] = (Zr, Zi)).  Indices past the ends are clamped and
// meet a zero on the other side.  Same LDS footprint, slab summation and accumulation over frequency blocks as above.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void dft_dbank_mf_kernel(const float* __restrict__ G, int nsl, long gs,
                                                                  float* __restrict__ dbank, int ksz, int L, int Lh, int M,
                                                                  int Cin, int FXB) {
    extern __shared__ float sm_dft[];
    float2* Kh = reinterpret_cast<float2*>(sm_dft);
    float2* Z = Kh + L * FXB;
    float2* tw = Z + ksz * FXB;
    float* outp = reinterpret_cast<float*>(tw + L);
    const int m = blockIdx.x / Cin, ci = blockIdx.x - m * Cin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kh = lane >> 5;
    const int ut = wave & 1, xt = wave >> 1;              // this wave's tile: rows 32 ut .., columns 32 xt .. (f resp. v)
    fill_twiddles(tw, L);
    for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) outp[i] = 0.f;
    const long rowlen = 2L * L * Cin;
    const int u = 32 * ut + li;                           // A row of stage 1 and 2 (twiddle resp. data row)
    for (int fx0 = 0; fx0 < Lh; fx0 += FXB) {
        const int nfx = min(FXB, Lh - fx0);
        __syncthreads();
        for (int i = threadIdx.x; i < nfx * L; i += blockDim.x) {
            const int fy = i % L, f = i / L;
            const float* r0 = G + ((long)(fx0 + f) * 2 * M + m) * rowlen + (long)(2 * ci) * L;
            const float* r1 = G + ((long)(fx0 + f) * 2 * M + M + m) * rowlen + (long)(2 * ci) * L;
            float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
            for (int sl = 0; sl < nsl; ++sl) {
                a += r0[sl * gs + fy];
                b += r1[sl * gs + L + fy];
                c += r0[sl * gs + L + fy];
                d += r1[sl * gs + fy];
            }
            Kh[fy * FXB + f] = make_float2(a + b, c - d);
        }
        __syncthreads();
        if (32 * ut < ksz && 32 * xt < nfx) {             // ---- stage 1: Z[u][f], this wave's (u tile, f tile)
            f32x16 zr, zi;
#pragma unroll
            for (int r = 0; r < 16; ++r) zr[r] = zi[r] = 0.f;
            const int f = min(32 * xt + li, nfx - 1);     // B column (clamped: columns >= nfx are not stored)
            int ph = (kh * u) % L;                        // (fy u) mod L for fy = kh
            const int dph = (2 * u) % L;
            for (int t = 0; 2 * t < L; ++t) {
                const int fy = 2 * t + kh;
                const float2 w = tw[ph];
                const float2 kv = Kh[min(fy, L - 1) * FXB + f];
                const float ok = fy < L ? 1.f : 0.f;      // odd frames: the last pair's second row does not exist
                const float c = w.x * ok, s_ = w.y * ok;
                zr = __builtin_amdgcn_mfma_f32_32x32x2f32(c, kv.x, zr, 0, 0, 0);
                zr = __builtin_amdgcn_mfma_f32_32x32x2f32(-s_, kv.y, zr, 0, 0, 0);
                zi = __builtin_amdgcn_mfma_f32_32x32x2f32(s_, kv.x, zi, 0, 0, 0);
                zi = __builtin_amdgcn_mfma_f32_32x32x2f32(c, kv.y, zi, 0, 0, 0);
                ph += dph;
                if (ph >= L) ph -= L;
            }
            const int fc = 32 * xt + li;                  // D layout: lane = column f, registers = rows u
            if (fc < nfx) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ur = 32 * ut + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (ur < ksz) Z[ur * FXB + fc] = make_float2(zr[r], zi[r]);
                }
            }
        }
        __syncthreads();
        if (32 * ut < ksz && 32 * xt < ksz) {             // ---- stage 2: out[u][v] += ..., this wave's (u tile, v tile)
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const int v = 32 * xt + li;                   // B column
            const int ua = min(u, ksz - 1);               // A row (clamped: rows >= ksz are not stored)
            int ph = (int)(((long)(fx0 + kh) * v) % L);   // (fx v) mod L for fx = fx0 + kh
            const int dph = (2 * v) % L;
            for (int t = 0; 2 * t < nfx; ++t) {
                const int fl = 2 * t + kh, fx = fx0 + fl;
                const float2 z = Z[ua * FXB + min(fl, nfx - 1)];
                const float cf = fl < nfx ? (((fx == 0) || (2 * fx == L)) ? 1.f : 2.f) : 0.f;
                const float2 w = tw[ph];
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(z.x, cf * w.x, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(z.y, -cf * w.y, o, 0, 0, 0);
                ph += dph;
                if (ph >= L) ph -= L;
            }
            if (v < ksz) {                                // D layout: lane = column v, registers = rows u
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ur = 32 * ut + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (ur < ksz) outp[ur * ksz + v] += o[r];      // (one owner per element in every block: no race)
                }
            }
        }
    }
    __syncthreads();
    const float inv = 1.f / ((float)L * (float)L);
    for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) dbank[((long)m * Cin + ci) * ksz * ksz + i] = outp[i] * inv;
}

// ------------------------------------------------------------------------------------------
// Round 5, mixed form (dft_spectra_x_kernel): the spectral weight gradient is G[fx][m | M+m][(ci, ri, u)] with the tap row u
// already spatial, so only the x stage of the inverse transform is left:
//   D[u][fx] = ( G[m][(0,u)] + G[M+m][(1,u)],  G[m][(1,u)] - G[M+m][(0,u)] )
//   dbank[m][ci][u][v] = 1/L sum_fx c_fx ( Dr cos(2 pi fx v / L) - Di sin(2 pi fx v / L) ),  u, v < ksz
// on the fp32 matrix pipe (rows u, columns v, k = pairs of fx), slabs summed in order while they are loaded.  One workgroup
// per (filter, channel); wave w owns the 32 x 32 output tiles (ut, vt) = w, w + 4, ..  LDS: D ksz*Lh complex + twiddles.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void dft_dbank_x_kernel(const float* __restrict__ G, int nsl, long gs,
                                                                 float* __restrict__ dbank, int ksz, int L, int Lh, int M,
                                                                 int Cin) {
    extern __shared__ float sm_dft[];
    float2* D = reinterpret_cast<float2*>(sm_dft);       // [u][fx]  (pitch Lh)
    float2* tw = D + ksz * Lh;
    const int m = blockIdx.x / Cin, ci = blockIdx.x - m * Cin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kh = lane >> 5;
    fill_twiddles(tw, L);
    const long rowlen = 2L * ksz * Cin;
    for (int i = threadIdx.x; i < Lh * ksz; i += blockDim.x) {
        const int u = i % ksz, f = i / ksz;
        const float* r0 = G + ((long)f * 2 * M + m) * rowlen + (long)(2 * ci) * ksz;
        const float* r1 = G + ((long)f * 2 * M + M + m) * rowlen + (long)(2 * ci) * ksz;
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
        for (int sl = 0; sl < nsl; ++sl) {
            a += r0[sl * gs + u];
            b += r1[sl * gs + ksz + u];
            c += r0[sl * gs + ksz + u];
            d += r1[sl * gs + u];
        }
        D[u * Lh + f] = make_float2(a + b, c - d);
    }
    __syncthreads();
    const int nt = (ksz + 31) / 32;
    const float inv = 1.f / (float)L;
    for (int tile = wave; tile < nt * nt; tile += 4) {
        const int ut = tile / nt, vt = tile - ut * nt;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        const int v = 32 * vt + li;                      // B column
        const int ua = min(32 * ut + li, ksz - 1);       // A row (clamped: rows >= ksz are not stored)
        int ph = (int)(((long)kh * v) % L);              // (fx v) mod L for fx = kh
        const int dph = (2 * v) % L;
        for (int t = 0; 2 * t < Lh; ++t) {
            const int fx = 2 * t + kh;
            const float2 z = D[ua * Lh + min(fx, Lh - 1)];
            const float cf = fx < Lh ? (((fx == 0) || (2 * fx == L)) ? 1.f : 2.f) : 0.f;
            const float2 w = tw[ph];
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(z.x, cf * w.x, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(z.y, -cf * w.y, o, 0, 0, 0);
            ph += dph;
            if (ph >= L) ph -= L;
        }
        if (v < ksz) {                                   // D layout: lane = column v, registers = rows u
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ur = 32 * ut + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (ur < ksz) dbank[((long)m * Cin + ci) * ksz * ksz + ur * ksz + v] = o[r] * inv;
            }
        }
    }
}

}  // namespace tvae
