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
//   dft_image_kernel   y -> A^T[fx][(re/im, fy)][(b,h)]           (one workgroup per image, DFT by direct sums in LDS)
//   dft_bank_kernel    bank -> W[fx][m | M+m][(re/im, fy)]         (one workgroup per filter)
//   dft_out_kernel     T -> out (+bias, activation)                (contraction over fx on the vector ALU)
//   dft_dy_kernel      dY -> S'[fx][m | M+m][(b,h)]
//   dft_dbank_kernel   G[fx][m | M+m][(re/im, fy)] -> dbank        (inverse DFT per filter, cropped to ksz x ksz)
// Single input channel (Cin = 1: MNIST / particle configurations).  Twiddles: sincospi of exactly reduced angles.
#pragma once
#include <hip/hip_runtime.h>
#include "small_kernels.hpp"
#include "conv_x6_kernels.hpp"

namespace tvae {

constexpr int DFT_WMAX = 40;           // output width / height handled by the register accumulators

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
// One workgroup per image:  Yh = DFT2(padded image) restricted to fx < Lh, then
// AT[fx][fy][b*Ho + h] = Re(Yh[fy][fx] e^{+2 pi i fy h / L}),  AT[fx][L + fy][..] = Im(..).
// LDS: image n*n floats, R n*Lh complex, Yh L*Lh complex, tw L complex.
// ------------------------------------------------------------------------------------------
__global__ void dft_image_kernel(const float* __restrict__ y, float* __restrict__ AT, int n, int pad, int L, int Lh,
                                 int Ho, long NBpad) {
    extern __shared__ float sm_dft[];
    float* img = sm_dft;
    float2* R = reinterpret_cast<float2*>(img + n * n);
    float2* Yh = R + n * Lh;
    float2* tw = Yh + L * Lh;
    const int b = blockIdx.x;
    fill_twiddles(tw, L);
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) img[i] = y[(long)b * n * n + i];
    __syncthreads();
    // R[yy][fx] = sum_x img[yy][x] e^{-2 pi i fx (x+pad) / L}
    for (int i = threadIdx.x; i < n * Lh; i += blockDim.x) {
        const int yy = i / Lh, fx = i - yy * Lh;
        float re = 0.f, im = 0.f;
        int ph = (fx * pad) % L;
        for (int x = 0; x < n; ++x) {
            const float v = img[yy * n + x];
            const float2 t = tw[ph];
            re += v * t.x;
            im -= v * t.y;
            ph += fx;
            if (ph >= L) ph -= L;
        }
        R[i] = make_float2(re, im);
    }
    __syncthreads();
    // Yh[fy][fx] = sum_y R[y][fx] e^{-2 pi i fy (y+pad) / L}
    for (int i = threadIdx.x; i < L * Lh; i += blockDim.x) {
        const int fy = i / Lh, fx = i - fy * Lh;
        float2 acc = make_float2(0.f, 0.f);
        int ph = (fy * pad) % L;
        for (int yy = 0; yy < n; ++yy) {
            const float2 r = R[yy * Lh + fx];
            const float2 t = make_float2(tw[ph].x, -tw[ph].y);
            const float2 p = cmul(r, t);
            acc.x += p.x;
            acc.y += p.y;
            ph += fy;
            if (ph >= L) ph -= L;
        }
        Yh[i] = acc;
    }
    __syncthreads();
    // AT[fx][ri*L + fy][b*Ho + h]
    const int total = Lh * L * Ho;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int h = i % Ho;
        const int t2 = i / Ho;
        const int fy = t2 % L, fx = t2 / L;
        const float2 v = cmul(Yh[fy * Lh + fx], tw[(fy * h) % L]);
        float* dst = AT + ((long)fx * 2 * L + fy) * NBpad + (long)b * Ho + h;
        dst[0] = v.x;
        dst[(long)L * NBpad] = v.y;
    }
}

// ------------------------------------------------------------------------------------------
// One workgroup per filter m:  Kh = DFT2(filter at the origin of the L x L frame), fx < Lh, then the real operand
// rows of the spectral GEMM:  W[fx][m][fy] = Kr, W[fx][m][L+fy] = Ki;  W[fx][M+m][fy] = -Ki, W[fx][M+m][L+fy] = Kr.
// ------------------------------------------------------------------------------------------
__global__ void dft_bank_kernel(const float* __restrict__ bank, float* __restrict__ W, int ksz, int L, int Lh, int M,
                                int Mb) {
    extern __shared__ float sm_dft[];
    float* ker = sm_dft;
    float2* Q = reinterpret_cast<float2*>(ker + ksz * ksz);
    float2* Kh = Q + ksz * Lh;
    float2* tw = Kh + L * Lh;
    const int m = blockIdx.x;
    fill_twiddles(tw, L);
    for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) ker[i] = bank[(long)m * ksz * ksz + i];
    __syncthreads();
    for (int i = threadIdx.x; i < ksz * Lh; i += blockDim.x) {
        const int u = i / Lh, fx = i - u * Lh;
        float re = 0.f, im = 0.f;
        int ph = 0;
        for (int v = 0; v < ksz; ++v) {
            const float k = ker[u * ksz + v];
            re += k * tw[ph].x;
            im -= k * tw[ph].y;
            ph += fx;
            if (ph >= L) ph -= L;
        }
        Q[i] = make_float2(re, im);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * Lh; i += blockDim.x) {
        const int fy = i / Lh, fx = i - fy * Lh;
        float2 acc = make_float2(0.f, 0.f);
        int ph = 0;
        for (int u = 0; u < ksz; ++u) {
            const float2 p = cmul(Q[u * Lh + fx], make_float2(tw[ph].x, -tw[ph].y));
            acc.x += p.x;
            acc.y += p.y;
            ph += fy;
            if (ph >= L) ph -= L;
        }
        Kh[i] = acc;
    }
    __syncthreads();
    const long rowlen = 2L * L;
    for (int i = threadIdx.x; i < Lh * L; i += blockDim.x) {
        const int fy = i % L, fx = i / L;
        const float2 k = Kh[fy * Lh + fx];
        float* r0 = W + ((long)fx * Mb + m) * rowlen;            // Mb >= 2M rows per fx (padding rows stay zero)
        float* r1 = W + ((long)fx * Mb + M + m) * rowlen;
        r0[fy] = k.x;
        r0[L + fy] = k.y;
        r1[fy] = -k.y;
        r1[L + fy] = k.x;
    }
}

// tables for the contraction over fx: cs[fx][w] = c_fx/L^2 cos(2 pi fx w / L), sn likewise (scaled: forward), and the
// unscaled pair cw / sw (backward);  4 * Lh * DFT_WMAX floats:  [cs | sn | cw | sw]
__global__ void dft_tables_kernel(float* __restrict__ tab, int L, int Lh) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Lh * DFT_WMAX; i += gridDim.x * blockDim.x) {
        const int fx = i / DFT_WMAX, w = i - fx * DFT_WMAX;
        float s, c;
        sincospif(2.0f * (float)((fx * w) % L) / (float)L, &s, &c);
        const float cf = ((fx == 0) || (2 * fx == L)) ? 1.f : 2.f;
        const float sc = cf / ((float)L * (float)L);
        tab[i] = c * sc;
        tab[Lh * DFT_WMAX + i] = s * sc;
        tab[2 * Lh * DFT_WMAX + 2 * i] = c;           // (cos, -sin) pairs for the dY transform
        tab[2 * Lh * DFT_WMAX + 2 * i + 1] = -s;
    }
}

// ------------------------------------------------------------------------------------------
// out[c][b][r][h][w] = act( bias[c] + sum_fx ( Tr[fx][m][n] cs[fx][w] - Ti[fx][m][n] sn[fx][w] ) ),  m = c*R + r,
// n = b*Ho + h.  One thread per (m, n); the 2*Lh values of T are coalesced along n, the tables are wave-uniform
// (scalar loads); results go through LDS so that the stores are contiguous runs.
// grid (ceil(NB/256), M), block 256.
// ------------------------------------------------------------------------------------------
typedef float f32x2p __attribute__((ext_vector_type(2)));

// WT = compile-time output width (33 and 17 are the reference configurations; DFT_WMAX is the generic instance):
// the 2*Lh*WT multiply-adds per (m, n) run as packed v_pk_fma_f32 on pairs of outputs.
template <int WT>
__global__ __launch_bounds__(256) void dft_out_kernel(const float* __restrict__ T, const float* __restrict__ tab,
                                                      const float* __restrict__ bias, float* __restrict__ out, int M,
                                                      int R, int B, int Ho, int Lh, long NBpad, int act, float slope) {
    constexpr int W2 = (WT + 1) / 2;
    __shared__ float st[256 * (2 * W2 + 1)];
    const int m = blockIdx.y;
    const long n0 = (long)blockIdx.x * 256;
    const long n = n0 + threadIdx.x;
    const long NB = (long)B * Ho;
    const f32x2p* cs = reinterpret_cast<const f32x2p*>(tab);                       // [Lh][DFT_WMAX/2] pairs
    const f32x2p* sn = reinterpret_cast<const f32x2p*>(tab + Lh * DFT_WMAX);
    f32x2p acc[W2];
#pragma unroll
    for (int q = 0; q < W2; ++q) acc[q] = (f32x2p){0.f, 0.f};
    const long col = n < NBpad ? n : NBpad - 1;
    const float* tr_p = T + dft_t_off(col, m, 2 * M, Lh);
    const float* ti_p = T + dft_t_off(col, M + m, 2 * M, Lh);
#pragma unroll 2
    for (int fx = 0; fx < Lh; ++fx) {
        const float tr = tr_p[fx * 128];
        const float ti = ti_p[fx * 128];
        const f32x2p tr2 = {tr, tr}, ti2 = {-ti, -ti};
#pragma unroll
        for (int q = 0; q < W2; ++q) {
            acc[q] = __builtin_elementwise_fma(tr2, cs[fx * (DFT_WMAX / 2) + q], acc[q]);
            acc[q] = __builtin_elementwise_fma(ti2, sn[fx * (DFT_WMAX / 2) + q], acc[q]);
        }
    }
    const int c = m / R, r = m - c * R;
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int q = 0; q < W2; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float v = acc[q][e] + bv;
            if (act == ACT_LRELU) v = v > 0.f ? v : v * slope;
            else if (act == ACT_TANH) v = tanhf(v);
            st[threadIdx.x * (2 * W2 + 1) + 2 * q + e] = v;
        }
    __syncthreads();
    const int P = Ho * Ho;
    const int cnt = 256 * Ho;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int t = e / Ho, w = e - t * Ho;
        const long nn = n0 + t;
        if (nn < NB) {
            const int b = (int)(nn / Ho), h = (int)(nn - (long)b * Ho);
            out[(((long)c * B + b) * R + r) * P + h * Ho + w] = st[t * (2 * W2 + 1) + w];
        }
    }
}

// ------------------------------------------------------------------------------------------
// S'[fx][m][n] = sum_w dY[m][n][w] cos(2 pi fx w / L),   S'[fx][M+m][n] = - sum_w dY[m][n][w] sin(2 pi fx w / L)
// (S = DFT over w of the output gradient).  dY is [c][b][r][h][w]; one thread per (m, n).
// grid (ceil(NBpad/256), M), block 256: columns n >= NB are written as zeros.
// ------------------------------------------------------------------------------------------
template <int WT>
__global__ __launch_bounds__(256) void dft_dy_kernel(const float* __restrict__ dY, const float* __restrict__ tab,
                                                     float* __restrict__ Sp, int M, int R, int B, int Ho, int Lh,
                                                     long NBpad) {
    __shared__ float st[256 * (WT + 1)];
    const int m = blockIdx.y;
    const long n0 = (long)blockIdx.x * 256;
    const long NB = (long)B * Ho;
    const int c = m / R, r = m - c * R;
    const int P = Ho * Ho;
    const int cnt = 256 * Ho;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int t = e / Ho, w = e - t * Ho;
        const long nn = n0 + t;
        float v = 0.f;
        if (nn < NB) {
            const int b = (int)(nn / Ho), h = (int)(nn - (long)b * Ho);
            v = dY[(((long)c * B + b) * R + r) * P + h * Ho + w];
        }
        st[t * (WT + 1) + w] = v;
    }
    __syncthreads();
    float d[WT];
#pragma unroll
    for (int w = 0; w < WT; ++w) d[w] = w < Ho ? st[threadIdx.x * (WT + 1) + w] : 0.f;
    // (cos, -sin) pairs: S = sum_w d[w] e^{-2 pi i fx w / L} as ONE packed FMA per w
    const f32x2p* cm = reinterpret_cast<const f32x2p*>(tab + 2 * Lh * DFT_WMAX);   // [Lh][DFT_WMAX] pairs
    const long n = n0 + threadIdx.x;
    if (n >= NBpad) return;
    float* sr_p = Sp + dft_t_off(n, m, 2 * M, Lh);
    float* si_p = Sp + dft_t_off(n, M + m, 2 * M, Lh);
#pragma unroll 2
    for (int fx = 0; fx < Lh; ++fx) {
        f32x2p s2 = {0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WT; ++w) s2 = __builtin_elementwise_fma((f32x2p){d[w], d[w]}, cm[fx * DFT_WMAX + w], s2);
        sr_p[fx * 128] = s2[0];
        si_p[fx * 128] = s2[1];
    }
}

// ==========================================================================================
// The two transforms along w (33-wide output rows <-> Lh frequencies) on the matrix pipe.  They are tiny GEMMs with
// a constant operand,  out[w][(m,n)] = sum_k E[w][k] T[k][(m,n)]  and  S'[k][(m,n)] = sum_w E'[k][w] dY[(m,n)][w],
// in the same exact-split arithmetic: the constant matrix is pre-split into cells once per call, every lane builds the
// B fragments of its own column in registers (8 strided / contiguous values, split3x8), no workgroup barrier in the
// loop.  The vector-ALU versions above needed 3 234 FMAs per (m,n); here the ALU only splits 98 (resp. 33) values.
// k order: k = ri*KH + fx with KH = Lh rounded up to 8 (ri = 0: real plane, 1: imaginary plane).
// ==========================================================================================
constexpr int DFT_WROWS = 64;          // padded output width of the forward transform (two 32-row MFMA tiles)

// E cells  [part][octet < 2*KH/8][row w < 64]:   E[w][ri*KH+fx] = ri ? -c_fx/L^2 sin(2 pi fx w/L) : c_fx/L^2 cos(..)
// E' cells [part][octet < WOCT][row k < 128]:    E'[ri*KH+fx][w] = ri ? -sin(2 pi fx w/L) : cos(2 pi fx w/L)
__global__ void dft_etab_kernel(uint4* __restrict__ E3, uint4* __restrict__ Ep3, int L, int Lh, int KH, int Ho, int WOCT) {
    const int KO = 2 * KH / 8;
    const int nE = KO * DFT_WROWS, nEp = WOCT * 128;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nE + nEp; i += gridDim.x * blockDim.x) {
        float r[8];
        uint4* dst;
        long stride;
        if (i < nE) {
            const int w = i % DFT_WROWS, o = i / DFT_WROWS;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * o + j, ri = k / KH, fx = k - ri * KH;
                float v = 0.f;
                if (w < Ho && fx < Lh) {
                    float sn, cs;
                    sincospif(2.0f * (float)((fx * w) % L) / (float)L, &sn, &cs);
                    const float cf = ((fx == 0) || (2 * fx == L)) ? 1.f : 2.f;
                    v = (ri ? -sn : cs) * cf / ((float)L * (float)L);
                }
                r[j] = v;
            }
            dst = E3 + i;
            stride = nE;
        } else {
            const int ii = i - nE;
            const int k = ii % 128, o = ii / 128;
            const int ri = k / KH, fx = k - ri * KH;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = 8 * o + j;
                float v = 0.f;
                if (w < Ho && fx < Lh && ri < 2) {
                    float sn, cs;
                    sincospif(2.0f * (float)((fx * w) % L) / (float)L, &sn, &cs);
                    v = ri ? -sn : cs;
                }
                r[j] = v;
            }
            dst = Ep3 + ii;
            stride = nEp;
        }
        Cell16 h, m, l;
        split3x8(r, h, m, l);
        dst[0] = h.u;
        dst[stride] = m.u;
        dst[2 * stride] = l.u;
    }
}

// out[c][b][r][h][w] = act(bias[c] + sum_k E[w][k] T[k][(m,n)]).  One wave per tile of 32 columns (m fixed, 32
// consecutive n); a workgroup of 4 waves walks tiles blockIdx.x*4 + wave, += gridDim.x*4.
__global__ __launch_bounds__(256) void dft_out_mfma_kernel(const float* __restrict__ T, const uint4* __restrict__ E3,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int M, int R, int B, int Ho, int Lh, int KH, long NBpad,
                                                           int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_o[];
    const int KO = 2 * KH / 8;                       // octets
    uint4* Es = reinterpret_cast<uint4*>(sm_o);      // [part][octet][64 rows]
    float* stg = reinterpret_cast<float*>(sm_o + (size_t)3 * KO * DFT_WROWS * 16);   // per wave [32 cols][Ho+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * KO * DFT_WROWS; i += 256) Es[i] = E3[i];
    __syncthreads();
    float* wst = stg + wave * 32 * (DFT_WROWS + 1);
    const int khalf = lane >> 5, j = lane & 31;
    const long tiles_n = NBpad / 32;
    const long ntiles = (long)M * tiles_n;
    const long NB = (long)B * Ho;
    const long plane = (long)M * Lh * 128;                        // offset of the imaginary rows
    const int P = Ho * Ho;
    const int nsteps = KO / 2;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float* tcol = T + dft_t_off(n0 + j, m, 2 * M, Lh);        // T is [n >> 7][ri*M + m][fx][n & 127]
        // all 8*nsteps values of this lane's column first (independent loads in flight), then split + MFMA
        constexpr int MAXS = 8;                         // nsteps <= 8 (2*KH <= 128)
        float v[MAXS][8];
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = 16 * s + 8 * khalf + q;
                const int ri = k / KH, fx = k - ri * KH;
                v[s][q] = (s < nsteps && fx < Lh && ri < 2) ? tcol[fx * 128 + ri * plane] : 0.f;
            }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            if (s < nsteps) {
                Cell16 bf[3];
                split3x8(v[s], bf[0], bf[1], bf[2]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Cell16 af[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[p].u = Es[(p * KO + 2 * s + khalf) * DFT_WROWS + i * 32 + j];
                    mfma6(acc[i], af, bf);
                }
            }
        }
        // stage [col][w] through LDS, then contiguous runs
        const int c = m / R, rr = m - c * R;
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                float v = acc[i][r] + bv;
                if (act == ACT_LRELU) v = v > 0.f ? v : v * slope;
                else if (act == ACT_TANH) v = tanhf(v);
                wst[j * (DFT_WROWS + 1) + w] = v;
            }
        // two columns per pass (lanes 0..31 / 32..63 take w = lane & 31 and, in a second sweep, w + 32)
        const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
        for (int t = khalf; t < 32; t += 2) {
            int h = h0 + t, b = b0;
            while (h >= Ho) { h -= Ho; ++b; }
            if (n0 + t < NB) {
                float* dst = out + (((long)c * B + b) * R + rr) * P + h * Ho;
                for (int w = j; w < Ho; w += 32) dst[w] = wst[t * (DFT_WROWS + 1) + w];
            }
        }
    }
}

// S'[fx][ri*M + m][n] = sum_w E'[ri*KH+fx][w] dY[(m,n)][w].  One wave per tile of 32 columns; dY rows are staged
// through LDS (per wave [32 cols][pitch]) so that every lane finds its 8-value cells 16-byte aligned.
__global__ __launch_bounds__(256) void dft_dy_mfma_kernel(const float* __restrict__ dY, const uint4* __restrict__ Ep3,
                                                          float* __restrict__ Sp, int M, int R, int B, int Ho, int Lh,
                                                          int KH, int WOCT, long NBpad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_o[];
    uint4* Es = reinterpret_cast<uint4*>(sm_o);      // [part][octet < WOCT][128 rows]
    const int pitch = 8 * WOCT + 4;                  // floats per staged column (16-byte multiple, odd multiple of 4)
    float* stg = reinterpret_cast<float*>(sm_o + (size_t)3 * WOCT * 128 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * WOCT * 128; i += 256) Es[i] = Ep3[i];
    __syncthreads();
    float* wst = stg + wave * 32 * pitch;
    const int khalf = lane >> 5, j = lane & 31;
    const long tiles_n = NBpad / 32;
    const long ntiles = (long)M * tiles_n;
    const long NB = (long)B * Ho;
    const long plane = (long)M * Lh * 128;                        // offset of the imaginary rows
    const int P = Ho * Ho;
    const int nsteps = (WOCT + 1) / 2;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const int c = m / R, rr = m - c * R;
        // stage the 32 columns (zeros beyond Ho and beyond NB): lanes 0..31 / 32..63 take alternate columns and
        // w = lane & 31 (+32); all loads of the tile are issued before the first LDS write
        {
            const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
            float sv[16][2];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int t = 2 * it + khalf;
                int h = h0 + t, b = b0;
                while (h >= Ho) { h -= Ho; ++b; }
                const bool ok = n0 + t < NB;
                const float* src = dY + (((long)c * B + (ok ? b : 0)) * R + rr) * P + h * Ho;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int w = j + 32 * q;
                    sv[it][q] = (ok && w < Ho) ? src[w] : 0.f;
                }
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int t = 2 * it + khalf;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int w = j + 32 * q;
                    if (w < pitch) wst[t * pitch + w] = sv[it][q];
                }
            }
        }
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int s = 0; s < nsteps; ++s) {
            const int o = 2 * s + khalf;
            float v[8];
            if (o < WOCT) {
                const float4 a = *reinterpret_cast<const float4*>(wst + j * pitch + 8 * o);
                const float4 bq = *reinterpret_cast<const float4*>(wst + j * pitch + 8 * o + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0.f;
            }
            Cell16 bf[3];
            split3x8(v, bf[0], bf[1], bf[2]);
            const int oa = o < WOCT ? o : 0;         // padding octet: B is zero, any valid A cell
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Cell16 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p].u = Es[(p * WOCT + oa) * 128 + i * 32 + j];
                mfma6(acc[i], af, bf);
            }
        }
        // direct stores: row k = (ri, fx), 32 consecutive n per row
        float* scol = Sp + dft_t_off(n0 + j, m, 2 * M, Lh);             // S' is [n >> 7][ri*M + m][fx][n & 127]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int ri = k / KH, fx = k - ri * KH;
                if (ri < 2 && fx < Lh) scol[fx * 128 + ri * plane] = acc[i][r];
            }
    }
}

// Bias gradient of the lifting convolution for free: the fx = 0 real row of S' is sum_w dY[m][n][w], so
// db[c] = sum_{r, n} S'[c*R + r][fx = 0][n]  (reads M*NB floats instead of a pass over dY).  One workgroup per channel.
__global__ void dft_dbias_kernel(const float* __restrict__ Sp, float* __restrict__ db, int R, int Lh, long NB, int M) {
    __shared__ float sm[16];
    const int c = blockIdx.x;
    float acc[1] = {0.f};
    for (int r = 0; r < R; ++r)
        for (long n = threadIdx.x; n < NB; n += blockDim.x) acc[0] += Sp[dft_t_off(n, c * R + r, 2 * M, Lh)];
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) db[c] = acc[0];
}

// ------------------------------------------------------------------------------------------
// One workgroup per filter m:  dKh'[fy][fx] from the four real blocks of G[fx][m | M+m][(re/im, fy)]
//   Re = G[m][fy] + G[M+m][L+fy],  Im = G[m][L+fy] - G[M+m][fy]
// then dbank[m][u][v] = 1/L^2 sum_fx c_fx Re( e^{2 pi i fx v/L} sum_fy dKh'[fy][fx] e^{2 pi i fy u/L} ),  u, v < ksz.
// ------------------------------------------------------------------------------------------
__global__ void dft_dbank_kernel(const float* __restrict__ G, float* __restrict__ dbank, int ksz, int L, int Lh, int M) {
    extern __shared__ float sm_dft[];
    float2* Kh = reinterpret_cast<float2*>(sm_dft);
    float2* Z = Kh + L * Lh;
    float2* tw = Z + ksz * Lh;
    const int m = blockIdx.x;
    fill_twiddles(tw, L);
    const long rowlen = 2L * L;
    for (int i = threadIdx.x; i < Lh * L; i += blockDim.x) {
        const int fy = i % L, fx = i / L;
        const float* r0 = G + ((long)fx * 2 * M + m) * rowlen;
        const float* r1 = G + ((long)fx * 2 * M + M + m) * rowlen;
        Kh[fy * Lh + fx] = make_float2(r0[fy] + r1[L + fy], r0[L + fy] - r1[fy]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ksz * Lh; i += blockDim.x) {
        const int u = i / Lh, fx = i - u * Lh;
        float2 acc = make_float2(0.f, 0.f);
        int ph = 0;
        for (int fy = 0; fy < L; ++fy) {
            const float2 p = cmul(Kh[fy * Lh + fx], tw[ph]);
            acc.x += p.x;
            acc.y += p.y;
            ph += u;
            if (ph >= L) ph -= L;
        }
        Z[i] = acc;
    }
    __syncthreads();
    const float inv = 1.f / ((float)L * (float)L);
    for (int i = threadIdx.x; i < ksz * ksz; i += blockDim.x) {
        const int u = i / ksz, v = i - u * ksz;
        float acc = 0.f;
        int ph = 0;
        for (int fx = 0; fx < Lh; ++fx) {
            const float cf = ((fx == 0) || (2 * fx == L)) ? 1.f : 2.f;
            const float2 z = Z[u * Lh + fx];
            acc += cf * (z.x * tw[ph].x - z.y * tw[ph].y);
            ph += v;
            if (ph >= L) ph -= L;
        }
        dbank[(long)m * ksz * ksz + i] = acc * inv;
    }
}

}  // namespace tvae
