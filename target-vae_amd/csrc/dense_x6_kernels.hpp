// Dense layers  Y[m][n] = sum_k A(m,k) X[k][n]  on the bf16 matrix pipe with fp32-equivalent results (the "x6"
// arithmetic of conv_x6_kernels.hpp: every operand split EXACTLY into three bf16 numbers, six partial products,
// fp32 accumulation).  Used for the 512x512 decoder layers: forward (A = W) and data gradient (A = W^T, X = dY).
//
//   * A (the weight, <= 1.5 MB) is split once per call by a tiny pre-pass into fragment-ready cells
//     [part][k-octet][row]; it stays L2 resident, and every wave loads the cells of its own 64 rows straight into
//     registers (one 16-byte load per fragment part, 64 consecutive cells per wave instruction), one step ahead.
//   * X is feature-major [k][n] (n contiguous), i.e. the 8 consecutive k an MFMA B fragment needs per lane are strided
//     in memory.  The workgroup transposes on the fly: thread (octet, n) loads its 8 k-values with 8 coalesced dword
//     loads (two steps ahead), splits them (split3x8) in the shadow of the MFMAs and writes the three 16-byte cells to
//     the [part][octet][n] LDS stage of the NEXT step; B fragments are conflict-free ds_read_b128.
//   * Tile 512 x 128, eight waves stacked along the rows (64 x 128 each: 2 x 4 MFMA tiles, 48 MFMAs per 16-k step),
//     two per SIMD; one barrier per step (the B stage is shared), double-buffered B stage, XCD-aware tile order.
//   * Direct epilogue from the accumulator layout (32 consecutive n = 128 contiguous bytes per row and instruction):
//     bias, residual, activation, activation-derivative mask.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "conv_x6_kernels.hpp"
#include "small_kernels.hpp"

// Timing ablations (-DTVAE_ABL=bits, never in the shipped build; results are then WRONG): 1 no sign bits, 2 no column dot,
// 4 no output stores, 8 no epilogue at all (generic forward epilogue / plain4 / weight gradient).  profiles/README.md round 4.
#ifndef TVAE_ABL
#define TVAE_ABL 0
#endif

namespace tvae {

// Pre-pass: W fp32 -> cells [part][octet][row < Rpad].
//   transpose == 0: A(row, k) = W[row*ldw + k]        (forward: rows = out features, k = in features)
//   transpose == 1: A(row, k) = W[k*ldw + row]        (data gradient: rows = in features, k = out features)
// Rows >= Rrows and k >= K are zero; K8pad octets (even).  scale (optional, [K]): A(row, k) is multiplied by scale[k]
// before the split (one fp32 rounding, as an elementwise fp32 product would have).
static __global__ void dense_split3_kernel(const float* __restrict__ W, long ldw, uint4* __restrict__ A3, int Rrows, int Rpad,
                                           int K, int K8pad, int transpose, const float* __restrict__ scale) {
    const long total = (long)K8pad * Rpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i % Rpad);
        const int o = (int)(i / Rpad);
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * o + j;
            r[j] = (row < Rrows && k < K) ? (transpose ? W[(long)k * ldw + row] : W[(long)row * ldw + k]) : 0.f;
            if (scale && k < K) r[j] *= scale[k];
        }
        Cell16 h, m, l;
        split3x8(r, h, m, l);
        A3[i] = h.u;
        A3[total + i] = m.u;
        A3[2 * total + i] = l.u;
    }
}

// The same pre-pass in the h3 arithmetic: cells [part < 2][octet][row] of fp16 parts of A(row, k) * s[row],
// s[row] = h3_scale(rowmax[row]), rowmax[row] = max_k |A(row, k)| (dense_rowmax_kernel): ONE POWER OF TWO PER ROW (round 4).
// A row of the operand is a row of the product, so the scale is undone per accumulator row in the GEMM's epilogue and a
// row that lies 2^20 below the rest of the matrix (a dead unit, a filter that has not started to train) is computed with
// the same relative accuracy as any other.  The row maxima live in the first Rpad words behind the two parts
// (A3[2 * total]: the buffer is sized for three parts), where the GEMM finds them again.
// Block = 64 rows x 16 k-slices, like dense_rowsum_kernel; rows >= Rrows (padding) get 0.
static __global__ __launch_bounds__(1024) void dense_rowmax_kernel(const float* __restrict__ W, long ldw, int Rrows, int Rpad,
                                                                   int K, int transpose, const float* __restrict__ scale,
                                                                   float* __restrict__ rowmax) {
    __shared__ float part[16][64];
    const int r = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + r;
    float mx = 0.f;
    if (row < Rrows)
        for (int k = sl; k < K; k += 16) {
            float v = transpose ? W[(long)k * ldw + row] : W[(long)row * ldw + k];
            if (scale) v *= scale[k];
            mx = fmaxf(mx, fabsf(v));
        }
    part[sl][r] = mx;
    __syncthreads();
    if (sl == 0 && row < Rpad) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t = fmaxf(t, part[q][r]);
        rowmax[row] = t;
    }
}
// The same for a row-major operand (transpose == 0), where the kernel above has every lane on a row of its own (64 rows x 4 B
// per load instruction): one WAVE per row, lanes along k (19 -> ~6 us for a 512 x 512 weight).  Block = 4 rows.
static __global__ __launch_bounds__(256) void dense_rowmax_rows_kernel(const float* __restrict__ W, long ldw, int Rrows, int Rpad,
                                                                      int K, const float* __restrict__ scale,
                                                                      float* __restrict__ rowmax) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Rpad) return;
    float mx = 0.f;
    if (row < Rrows)
        for (int k = lane; k < K; k += 64) {
            float v = W[(long)row * ldw + k];
            if (scale) v *= scale[k];
            mx = fmaxf(mx, fabsf(v));
        }
    mx = h3_wave_max(mx);
    if (lane == 0) rowmax[row] = mx;
}
// max |x| of a vector into ONE word (the gy operand of the two-valued weight gradient)
static __global__ void dense_absmax_kernel(const float* __restrict__ W, long ldw, int Rrows, int K, int transpose,
                                           const float* __restrict__ scale, float* __restrict__ amax) {
    const long total = (long)Rrows * K;
    float mx = 0.f;
    if (!scale && (transpose ? ldw == Rrows : ldw == K) && (reinterpret_cast<size_t>(W) & 15) == 0) {
        // one contiguous vector: 16-byte loads (18 -> ~6 us for the 1 M values of gy)
        const float4* w4 = reinterpret_cast<const float4*>(W);
        const long n4 = total / 4, gsz = (long)gridDim.x * blockDim.x;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gsz) {
            const float4 v = w4[i];
            mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gsz) mx = fmaxf(mx, fabsf(W[i]));
        h3_block_amax(mx, amax);
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // consecutive threads read consecutive addresses in either orientation
        const int k = transpose ? (int)(i / Rrows) : (int)(i % K);
        const int row = transpose ? (int)(i % Rrows) : (int)(i / K);
        float v = transpose ? W[(long)k * ldw + row] : W[(long)row * ldw + k];
        if (scale) v *= scale[k];
        mx = fmaxf(mx, fabsf(v));
    }
    h3_block_amax(mx, amax);
}
// max |x| over `rows` (grid.y) rows of n floats with row stride ld (16-byte aligned rows): float4 loads
static __global__ void h3_absmax_rows_kernel(const float* __restrict__ x, long ld, long n, float* __restrict__ amax) {
    const float4* r4 = reinterpret_cast<const float4*>(x + (long)blockIdx.y * ld);
    const long n4 = n / 4;
    float mx = 0.f;
    const long gsz = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * gsz < n4; i += 4 * gsz) {             // four independent 16-byte loads in flight per thread
        const float4 v0 = r4[i], v1 = r4[i + gsz];
        const float4 v2 = r4[i + 2 * gsz], v3 = r4[i + 3 * gsz];
        const float a0 = fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w)));
        const float a1 = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)));
        const float a2 = fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w)));
        const float a3 = fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)));
        mx = fmaxf(mx, fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    }
    for (; i < n4; i += gsz) {
        const float4 v = r4[i];
        mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        mx = fmaxf(mx, fabsf(x[(long)blockIdx.y * ld + i]));
    h3_block_amax(mx, amax);
}
static __global__ void h3_zero_slots_kernel(float* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.f;
}
// rowmax: one maximum per row of the operand (row = i % Rpad, the stacked row index of batched operands)
static __global__ void dense_split2h_kernel(const float* __restrict__ W, long ldw, uint4* __restrict__ A3, int Rrows, int Rpad,
                                            int K, int K8pad, int transpose, const float* __restrict__ scale,
                                            const float* __restrict__ rowmax) {
    const long total = (long)K8pad * Rpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // row-major operands (forward form): consecutive threads take consecutive octets of a ROW, i.e. consecutive 32-byte
        // pieces of memory -- with the row fastest every lane of a load touched its own 768-byte-strided line (the spectral
        // weight's 77 MB: 100 -> 60 us); the 16-byte cell stores of eight neighbouring rows still complete one line.
        // Transposed operands keep the row fastest (there it IS the contiguous index).
        const int row = transpose ? (int)(i % Rpad) : (int)(i / K8pad);
        const int o = transpose ? (int)(i / Rpad) : (int)(i % K8pad);
        const long ci = (long)o * Rpad + row;            // cell index
        const float s = h3_scale(rowmax[row]);
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * o + j;
            r[j] = (row < Rrows && k < K) ? (transpose ? W[(long)k * ldw + row] : W[(long)row * ldw + k]) : 0.f;
            if (scale && k < K) r[j] *= scale[k];
            r[j] *= s;
        }
        Cell16 h, l;
        split2hx8(r, h, l);
        A3[ci] = h.u;
        A3[total + ci] = l.u;
    }
}

// The same split for row-major operands (transpose == 0) with BOTH sides coalesced: a workgroup takes 32 rows; its threads read
// consecutive octets of a row (consecutive 32-byte pieces of memory), park the cells in LDS as [octet][row] and write each
// octet's 32 cells as one 512-byte run (the kernel above stores 16-byte cells 16 Rpad bytes apart: 62 us for the 38 MB of the
// spectral weight; this one ~25).  Dynamic LDS: 32 K8pad cells x 2 parts.
static __global__ __launch_bounds__(256) void dense_split2h_rows_kernel(const float* __restrict__ W, long ldw, uint4* __restrict__ A3,
                                                                        int Rrows, int Rpad, int K, int K8pad,
                                                                        const float* __restrict__ scale,
                                                                        const float* __restrict__ rowmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char split_lds[];
    uint4* Hs = reinterpret_cast<uint4*>(split_lds);
    uint4* Ls = Hs + 32 * K8pad;
    const long total = (long)K8pad * Rpad;
    const int r0 = blockIdx.x * 32, ncell = 32 * K8pad;
    const bool vec = (ldw & 3) == 0 && (reinterpret_cast<size_t>(W) & 15) == 0;
    for (int c = threadIdx.x; c < ncell; c += 256) {
        const int rl = c / K8pad, o = c - rl * K8pad, row = r0 + rl;
        const float s = h3_scale(rowmax[row]);
        float r[8];
        if (vec && row < Rrows && 8 * o + 8 <= K) {      // two 16-byte loads
            const float4 a = *reinterpret_cast<const float4*>(W + (long)row * ldw + 8 * o);
            const float4 b = *reinterpret_cast<const float4*>(W + (long)row * ldw + 8 * o + 4);
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * o + j;
                r[j] = (row < Rrows && k < K) ? W[(long)row * ldw + k] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * o + j;
            if (scale && k < K) r[j] *= scale[k];
            r[j] *= s;
        }
        Cell16 h, l;
        split2hx8(r, h, l);
        Hs[o * 32 + rl] = h.u;
        Ls[o * 32 + rl] = l.u;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncell; c += 256) {
        const int o = c >> 5, rl = c & 31;
        const long ci = (long)o * Rpad + r0 + rl;
        A3[ci] = Hs[c];
        A3[total + ci] = Ls[c];
    }
}

// h3 scale of the recomputed first-layer activation (VirtAct): slots[0] = max |xr|, slots[1] = max_k (|wc[k][0]| + |wc[k][1]|),
// slots[2] = max_{b,k} |bc[k] + lb[b][k]| (atomic maxima into zeroed slots); |act(pre)| <= |pre| <= slots[1] slots[0] + slots[2]
// for LeakyReLU (slope <= 1), tanh and the identity.  slots[3] is the caller's (max |gy| of the weight gradient) and
// slots[4 + k] = max_b |bc[k] + lb[b][k]|: with it a consumer whose operand ROW is feature k (the weight gradient) bounds that
// row alone, (|wc[k][0]| + |wc[k][1]|) slots[0] + slots[4 + k].
static __global__ void dec_l0_bound_kernel(const float* __restrict__ xr, long nxr, const float* __restrict__ wc,
                                           const float* __restrict__ bc, const float* __restrict__ lb, long nlb, int K,
                                           float* __restrict__ slots) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    const long n4 = (reinterpret_cast<size_t>(xr) & 15) == 0 ? nxr / 4 : 0;        // 16-byte loads, several in flight
    const float4* x4 = reinterpret_cast<const float4*>(xr);
#pragma unroll 4
    for (long i = gid; i < n4; i += gsz) {
        const float4 v = x4[i];
        m0 = fmaxf(fmaxf(m0, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (long i = 4 * n4 + gid; i < nxr; i += gsz) m0 = fmaxf(m0, fabsf(xr[i]));
    for (long i = gid; i < K; i += gsz) m1 = fmaxf(m1, fabsf(wc[2 * i]) + fabsf(wc[2 * i + 1]));
    if (lb) {
        // thread = (feature k, one of 16 image chunks): a local maximum over its images, then ONE atomic per thread (one atomic
        // per (image, feature) pair -- 256 per word at the bench shape -- made this loop the kernel's cost)
        const long nimg = nlb / K, per = (nimg + 15) / 16;
        for (long t = gid; t < 16L * K; t += gsz) {
            const int k = (int)(t % K);
            const long b0 = (t / K) * per, b1 = b0 + per < nimg ? b0 + per : nimg;
            float v = 0.f;
            for (long b = b0; b < b1; ++b) v = fmaxf(v, fabsf(bc[k] + lb[b * K + k]));
            if (b1 > b0) {
                m2 = fmaxf(m2, v);
                h3_atomic_amax(slots + 4 + k, v);
            }
        }
    } else {
        for (long i = gid; i < K; i += gsz) {
            m2 = fmaxf(m2, fabsf(bc[i]));
            slots[4 + i] = fabsf(bc[i]);
        }
    }
    h3_block_amax(m0, slots);
    h3_block_amax(m1, slots + 1);
    h3_block_amax(m2, slots + 2);
}

// rowsum[row] = sum_k A(row, k) of the (scaled) operand above (VirtGrad.csum).  Block = 64 rows x 16 k-slices; the slice
// sums are added in slice order (deterministic).
static __global__ __launch_bounds__(1024) void dense_rowsum_kernel(const float* __restrict__ W, long ldw, int Rrows, int K,
                                                                   int transpose, const float* __restrict__ scale,
                                                                   float* __restrict__ rowsum) {
    __shared__ float part[16][64];
    const int r = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + r;
    float s = 0.f;
    if (row < Rrows)
        for (int k = sl; k < K; k += 16) {
            float v = transpose ? W[(long)k * ldw + row] : W[(long)row * ldw + k];
            if (scale) v *= scale[k];
            s += v;
        }
    part[sl][r] = s;
    __syncthreads();
    if (sl == 0 && row < Rrows) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][r];
        rowsum[row] = t;
    }
}

// NP = bf16 parts per operand: 3 = the exact split (six partial products), 1 = plain bf16 operands (round to nearest, ONE
// product, fp32 accumulate): the throughput mode BASELINE.json configs 2 and 5 name -- NOT fp32-equivalent (2^-9 per
// operand), never the default, its own tolerance in the tests.
template <int NP>
__device__ __forceinline__ void mfma_np(f32x16& acc, const Cell16 (&a)[3], const Cell16 (&b)[3]) {
    if (NP == 3) mfma6(acc, a, b);
    else if (NP == 2) mfma3h(acc, a, b);                 // two fp16 parts, three products ("h3", conv_x6_kernels.hpp)
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
}
// one part of A against one cell of B in the arithmetic of NP (the exact 0 / 1 operands: 1.0 is 0x3f80 as bf16, 0x3c00 as fp16)
template <int NP>
__device__ __forceinline__ void mfma_part(f32x16& acc, const Cell16& a, const Cell16& b) {
    if (NP == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
template <int NP> struct OneBits { static constexpr unsigned lo = NP == 2 ? 0x3c00u : 0x3f80u, hi = lo << 16; };
// h3 epilogue: the inverse powers of two of the accumulator's ROW (operand A, one scale per row: ia[row of the tile]) and
// COLUMN (operand X: ix[column of the tile]; XS = false: the exact 0 / 1 operand has none), one after the other (their product
// may leave the fp32 range, each intermediate result does not unless the true value does).  ia / ix are LDS tables.
template <bool XS>
__device__ __forceinline__ void h3_unscale_rc(f32x16 (&acc)[2][4], const float* ia, const float* ix, int wave, int lane) {
    float ixv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ixv[j] = XS ? ix[j * 32 + (lane & 31)] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = ia[wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j][r] = XS ? (acc[i][j][r] * a) * ixv[j] : acc[i][j][r] * a;
        }
}
// rounds two values to bf16 (RNE), packed (x0 in the low half): the whole "split" of the one-part mode
__device__ __forceinline__ unsigned bf16_pair(float x0, float x1) {
    const f32x2v x = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v));
}

// h3 arithmetic (NP == 2) only: device words holding max |.| (or an upper bound) of the operands.  Round 4: the scale is a
// power of two PER ROW of each operand in the sense of the product -- per output row for A, per output column (forward form)
// or per output column = row of X (weight-gradient form) for X -- wherever the producer of the operand can say what the
// maximum of that row is; the epilogue undoes row and column factors separately.  What is left to one scale is the
// reduction index, where an element far below its row's maximum is also negligible in the sum.
struct H3Scale {
    const float* amax_a;   // a_rows == 0: one word (max |A|).  a_rows != 0: one word per row -- forward form: indexed like the
                           // stacked rows of A3 (row tile offset + row); weight-gradient form: [batch * a_bstride + (row % a_mod)]
    const float* amax_x;   // x_group == 0: one word (or the bound words of a recomputed operand, dec_l0_bound_kernel).
                           // x_group > 0, forward form: word [batch * x_bstride + min(column / x_group, x_bstride - 1)] (one per
                           // image of x_group columns); weight-gradient form: word [batch * x_bstride + row / x_group]
    int a_rows, a_bstride, a_mod;
    int x_group, x_bstride;
};
constexpr H3Scale H3_NONE = {nullptr, nullptr, 0, 0, 0, 0, 0};

constexpr int DX6_THREADS = 512;                   // 8 waves: two per SIMD
constexpr int DX6_ROWS = 512;                      // tile rows (8 waves x 64)

struct VirtGrad {          // the streamed gradient operand given implicitly (backward of the single-output last Linear,
    const float* wo;       // src/models.py:121-123, never materialised):  value(m, n) = wo[m] * gy[n] * act'(H[m][n]),
    const float* gy;       // with the saved activation H passed where the operand pointer is expected
    int act;
    float slope;
    // LeakyReLU only -- the TWO-VALUED form (dense_x6_kernel<3>): act' = slope + (1 - slope) * [H > 0], so
    //   sum_k W(m,k) wo[k] gy[n] act'(H[k][n]) = gy[n] * ( slope * csum[m] + (1 - slope) * sum_k W'(m,k) [H[k][n] > 0] )
    // with W' = W diag(wo) (split by tvae_dense_split3 with scale = wo) and csum[m] = sum_k W'(m,k).  The streamed
    // operand [H > 0] is 0 or 1: ONE exact bf16 part, so a product block is three MFMAs instead of six and the operand
    // needs no split arithmetic at all.  wo is then unused by the kernel.
    const float* csum;
    // weight gradient only (LRF = 2): the operand given as the sign bits a forward launch stored (ColDot.bits),
    // bits[m * (N/32) + n/32]; the saved activation itself is then not read.  (The data gradient keeps reading H: its
    // operand loads are one instruction per k-row either way, and the bit form measured 0.2 ms slower.)
    const unsigned* bits;
    // data gradient, dense_x6_kernel<5> (round 4): the 0 / 1 operand comes from these bits as well and H is not an argument at
    // all -- the forward launch then need not store it.  Of the two row sums of <4> only the first is formed (from bits and
    // gy); the second, dWo[m] = sum_n gy[n] H[m][n], follows from quantities the backward computes anyway:
    //   H = act(pre), act(p) = act'(p) p for LeakyReLU, pre[m][n] = sum_k W[m][k] X[k][n] + b[m]
    //   => dWo[m] = sum_n g'[m][n] pre[m][n] = sum_k W[m][k] G[m][k] + b[m] g0[m],   g'[m][n] = gy[n] act'(H[m][n]),
    // with G[m][k] = sum_n g'[m][n] X[k][n] (the layer's weight gradient before its row factor wo[m]: `raw` below) and
    // g0[m] = sum_n g'[m][n] (its bias gradient before wo[m]).  A 512 x 512 dot product per row instead of a 2.1 GB tensor
    // written by the forward and read back by the backward.
    // data gradient only (dense_x6_kernel<4>): the launch that streams H for the two-valued operand also produces, per
    // 128-column tile, the two row sums the backward of the single-output Linear needs of H (tvae_dec_out_bwd's whole job):
    //   rpart[(m * (N/128) + tile_n) * 2 + 0] = sum_{n in tile} gy[n] [H[m][n] > 0]   -> bias gradient of the layer producing H
    //   rpart[(m * (N/128) + tile_n) * 2 + 1] = sum_{n in tile} gy[n] H[m][n]         -> dWo[m]
    float* rpart;
    // weight gradient only (LRF): 1 = leave the row factor wo[m] off the slabs (the finalize applies it and forms
    // sum_k W[m][k] G[m][k] from the raw sums: wgrad_lrf_finalize_kernel)
    int raw;
};
__device__ __forceinline__ float virt_value(const VirtGrad& vg, float h, float wo, float g) {
    const float dv = vg.act == ACT_LRELU ? (h > 0.f ? 1.f : vg.slope) : (vg.act == ACT_TANH ? 1.f - h * h : 1.f);
    return wo * g * dv;
}

struct VirtAct {           // a streamed ACTIVATION operand given implicitly: the output of SpatialGenerator's first
    const float* xr;       // layer (no Fourier features), h0[f][n] = act(wc[f][0] x'_0[n] + wc[f][1] x'_1[n] + bc[f] +
    const float* wc;       // lb[n / Np][f]) (src/models.py:107-118), recomputed where it is consumed instead of being
    const float* bc;       // stored: xr [N][2], wc [F][2], bc [F], lb [B][F] or NULL.  Column tiles / chunks must not
    const float* lb;       // straddle images where noted.
    int Np;
    int act;
    float slope;
};

struct InTail {            // optional fused backward of SpatialGenerator's FIRST layer (h0 = act(Wc x' + ..), in_dim 2,
    const float* xr;       // src/models.py:107-118) on the output dX of a data-gradient launch (single row tile, panels
    const float* wc;       // inside one image):  gxr[n][j] = sum_f wc[f][j] dX[f][n];
    float* gxr;            // part[panel][f][0..2] = sum_{n in panel} dX[f][n] * (1, x'_0[n], x'_1[n])
    float* part;           // xr [N][2], wc [rows][2], gxr [N][2], part [N/128][rows][3]
    const float* bc;       // bc != NULL: the mask operand (this layer's own output h0) is recomputed from (xr, wc, bc, lb)
    const float* lb;       //   instead of being read from ep.aux
    int Np;
};

struct ColDot {            // optional fused skinny layer on the OUTPUT of this one (single row tile, one output):
    const float* w;        //   y[n] = b[0] + sum_m w[m] * Y[m][n]   (SpatialGenerator's last Linear, src/models.py:121-123)
    const float* b;
    float* y;
    // optional second output of a LeakyReLU layer: the SIGN of every output element, one bit each,
    // bits[m * (N / 32) + n / 32] bit (n & 31) = [Y[m][n] > 0].  The two-valued implicit gradient (VirtGrad.csum,
    // dense_wgrad_x6_dma_kernel LRF) needs nothing else of Y: its data / weight gradient then stream 1/32 of the bytes.
    unsigned* bits;
};

// Sum over the 32 lanes of each half wave, left in every lane of the half: four DPP adds inside the 16-lane rows (quad
// swaps, half-mirror, mirror: vector-ALU only) and ONE cross-row exchange, instead of five dependent ds_bpermute round
// trips per value (the fused first-layer backward reduces 96 values per lane and tile this way).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum(float x) {
    x = dpp_add<0xB1>(x);                                // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);                                // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);                               // row_half_mirror: the other quad of each 8
    x = dpp_add<0x140>(x);                               // row_mirror: the other 8 of each 16
    return x + __shfl_xor(x, 16, 64);                    // the other row of the half wave
}

template <int ACT, int MASK, bool RES, bool AV, bool MB>
__device__ __forceinline__ void dense_x6_epilogue(f32x16 (&acc)[2][4], const Epilogue& ep, const float* bsm, int m0,
                                                  int n0, int M, int wave, int lane, const float* wsm,
                                                  float (&ysum)[4], const InTail& it, const float* wc2,
                                                  float (&gsum)[4][2], int tile_n, const float* cbm,
                                                  const float (&gyv)[4], float oms, unsigned* sbits, long bitw,
                                                  float& vmax) {
    // direct from the accumulator layout: 32 consecutive n (128 contiguous bytes) per row and instruction.  The
    // optional mask / residual operands are fetched 32 at a time (8 rows x 4 column tiles) before any of them is
    // consumed: 4 global round trips per wave tile instead of one per row, within the 256-register budget of two
    // waves per SIMD.
    const long off = n0 + (lane & 31);
    const long coff = ep.ctile ? (long)tile_n * ep.ctile + (lane & 31) : off;     // column-tiled output (see Epilogue)
    float x0[4], x1[4];
    if (it.xr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x0[j] = it.xr[2 * (off + j * 32)];
            x1[j] = it.xr[2 * (off + j * 32) + 1];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            float av[8][4], rv[8][4];
            long mrow[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = rh * 8 + q;
                mrow[q] = (long)min(m0 + wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
            }
            if (MASK != ACT_NONE && !AV) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[q][j] = ep.aux[mrow[q] * ep.ldaux + off + j * 32];
            }
            if (RES) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) rv[q][j] = ep.res[mrow[q] * ep.ldres + off + j * 32];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = rh * 8 + q;
                const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int m = m0 + row;
                const float bv = bsm[row];
                float* crow = ep.C ? ep.C + mrow[q] * ep.ldc + coff : nullptr;
                float rs[3] = {0.f, 0.f, 0.f};
                unsigned sw[4] = {0u, 0u, 0u, 0u};               // sign words of this row's four 32-column groups
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // MB (two-valued implicit gradient, see VirtGrad): bv = slope * csum[m], oms = 1 - slope
                    float v = MB ? gyv[j] * __fmaf_rn(oms, acc[i][j][r], bv) : acc[i][j][r] + bv;
                    if (RES) v += rv[q][j];
                    if (ACT == ACT_LRELU) v = v > 0.f ? v : v * ep.slope;
                    else if (ACT == ACT_TANH) v = tanhf(v);
                    if (MASK != ACT_NONE) {
                        float a;
                        if (AV) {    // recompute the masked layer's output (see VirtAct) instead of loading it
                            a = dec_l0_pre(wc2[2 * row], wc2[2 * row + 1], cbm[row], cbm[DX6_ROWS + row], x0[j], x1[j]);
                            if (MASK == ACT_TANH) a = tanhf(a);             // LReLU: only the sign is used
                        } else {
                            a = av[q][j];
                        }
                        if (MASK == ACT_LRELU) v *= a > 0.f ? 1.f : ep.slope;
                        else v *= 1.f - a * a;
                    }
                    if (sbits && !(TVAE_ABL & 1)) {                                 // sign bits (ColDot.bits): lanes 0-31 hold one row, 32-63 the row + 4
                        const unsigned long long bl = __ballot(v > 0.f);
                        sw[j] = (unsigned)(lane < 32 ? bl : bl >> 32);
                    }
                    if (m < M) {
                        vmax = fmaxf(vmax, fabsf(v));            // (Epilogue.amax_out)
                        if (crow && !(TVAE_ABL & 4)) __builtin_nontemporal_store(v, crow + j * 32);     // written once, read by a later launch: no reuse in L2
                        if (wsm && !(TVAE_ABL & 2)) ysum[j] += wsm[row] * v;        // fused column dot (next, skinny layer)
                        if (it.xr) {                             // fused first-layer backward (see InTail)
                            rs[0] += v;
                            rs[1] += v * x0[j];
                            rs[2] += v * x1[j];
                            gsum[j][0] += wc2[2 * row] * v;
                            gsum[j][1] += wc2[2 * row + 1] * v;
                        }
                    }
                }
                if (sbits && !(TVAE_ABL & 1) && (lane & 31) == 0 && m < M)          // the tile's 128 columns of a row are ONE aligned 16-byte store
                    *reinterpret_cast<uint4*>(sbits + (long)m * bitw + (n0 >> 5)) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
                if (it.xr) {
                    // row sums over this panel's 128 columns: the 32 lanes of a half wave hold the same row
#pragma unroll
                    for (int k = 0; k < 3; ++k) rs[k] = half_wave_sum(rs[k]);
                    if ((lane & 31) == 0 && m < M) {
                        float* pp = it.part + ((long)tile_n * M + m) * 3;
                        pp[0] = rs[0];
                        pp[1] = rs[1];
                        pp[2] = rs[2];
                    }
                }
            }
        }
}

// Lean epilogue of the decoder's LAST hidden layer when its activation is not stored (round 4: VirtGrad / dec.no_h): bias +
// LeakyReLU, the sign bits and the fused column dot of a full 512-row tile -- nothing else, no per-element uniform branches,
// no addresses of an output that is not written.  The generic epilogue spent 0.62 ms of the 2.0 ms launch here with one
// workgroup per CU and the matrix pipe idle (ablations, profiles/README.md round 4: sign bits 0.16, column dot 0.16, the
// rest 0.30); this one reads its per-row constants four rows at a time (rows 8 q + 4 half + p, p < 4, are consecutive:
// one ds_read_b128 per table) and folds the h3 row factor into the bias FMA (products with powers of two are exact, so
// the result is bitwise that of unscale-then-add).  max(v, slope v) IS LeakyReLU for 0 < slope < 1, signed zeros and NaN
// included.  bsm / wsm / h3a: 16-byte aligned LDS tables indexed by the row of the tile.
template <int NP>
__device__ __forceinline__ void dense_x6_epilogue_lean(f32x16 (&acc)[2][4], const float* bsm, const float* wsm, const float* h3a,
                                                       const float* h3x, float slope, int wave, int lane, int m0, int n0,
                                                       float (&ysum)[4], unsigned* sbits, long bitw) {
    float ixv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ixv[j] = NP == 2 ? h3x[j * 32 + (lane & 31)] : 1.f;
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = wave * 64 + i * 32 + 8 * q + 4 * half;        // rows rb .. rb + 3 <-> registers r = 4 q + p
            const float4 b4 = *reinterpret_cast<const float4*>(bsm + rb);
            const float4 w4 = *reinterpret_cast<const float4*>(wsm + rb);
            float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (NP == 2) a4 = *reinterpret_cast<const float4*>(h3a + rb);
            const float bq[4] = {b4.x, b4.y, b4.z, b4.w}, wq[4] = {w4.x, w4.y, w4.z, w4.w}, aq[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = 4 * q + p;
                unsigned sw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = NP == 2 ? __fmaf_rn(acc[i][j][r] * ixv[j], aq[p], bq[p]) : acc[i][j][r] + bq[p];
                    v = fmaxf(v, v * slope);
                    ysum[j] = __fmaf_rn(wq[p], v, ysum[j]);
                    const unsigned long long bl = __ballot(v > 0.f);     // lanes 0-31 hold one row, 32-63 the row + 4
                    sw[j] = (unsigned)(half ? bl >> 32 : bl);
                }
                if ((lane & 31) == 0 && sbits)             // (sbits == nullptr: inference-mode forward, nothing is kept for a backward)
                    *reinterpret_cast<uint4*>(sbits + (long)(m0 + rb + p) * bitw + (n0 >> 5)) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
            }
            __builtin_amdgcn_sched_barrier(0);           // one row group at a time: no hoisting of later groups' reads over this one
        }
}

// Sum over the 32 lanes of each half wave WITHOUT the LDS crossbar: four DPP adds inside the 16-lane rows, then row_bcast15
// (lane 15 of rows 0 / 2 into rows 1 / 3).  The total is valid in lanes 16-31 (first half) and 48-63 (second half) only.
__device__ __forceinline__ float half_wave_sum_hi(float x) {
    x = dpp_add<0xB1>(x);
    x = dpp_add<0x4E>(x);
    x = dpp_add<0x141>(x);
    x = dpp_add<0x140>(x);
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xA, 0xF, false));
}

// Lean epilogue of the two-valued data gradient of the decoder's FIRST hidden layer with the coordinate layer's backward fused
// into it (InTail with the recomputed mask operand; nothing is stored but the per-panel row sums and the coordinate
// gradient): the hot shape of every reference decoder without Fourier features.  The generic epilogue took 0.83 ms of this
// 1.79 ms launch (ablation, profiles/README.md round 4) -- per-element uniform branches on options that are off, 64-bit
// addresses of an output that does not exist, one LDS read per row and table, five dependent crossbar permutes per row sum.
// Here: per-row constants four rows at a time (ds_read_b128), u = act'(h0) (oms acc + slope csum) without the column
// factor gy[n], which moves into the three column vectors (gy, gy x0, gy x1) of the row sums and onto the finished
// coordinate sums; row sums reduced by DPP only.  13 vector instructions per element instead of ~40.
//   part[(tile_n M + f) 3 + k] = sum_{n in panel} dX[f][n] (1, x'_0[n], x'_1[n]),   gsum[j][c] = sum_f wc[f][c] dX[f][n_j]
template <int NP>
__device__ __forceinline__ void dense_x6_epilogue_lean_in(f32x16 (&acc)[2][4], const float* bsm, const float* wc2,
                                                          const float* cbm, const float* h3a, float slope, float oms, int wave,
                                                          int lane, int m0, int M, int tile_n, const float (&gyv)[4],
                                                          const float (&x0)[4], const float (&x1)[4], float* part,
                                                          float (&gsum)[4][2]) {
    float gx0[4], gx1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gx0[j] = gyv[j] * x0[j];
        gx1[j] = gyv[j] * x1[j];
    }
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = wave * 64 + i * 32 + 8 * q + 4 * half;        // rows rb .. rb + 3 <-> registers r = 4 q + p
            const float4 b4 = *reinterpret_cast<const float4*>(bsm + rb);
            const float4 c4 = *reinterpret_cast<const float4*>(cbm + rb);
            const float4 l4 = *reinterpret_cast<const float4*>(cbm + DX6_ROWS + rb);
            const float4 wA = *reinterpret_cast<const float4*>(wc2 + 2 * rb), wB = *reinterpret_cast<const float4*>(wc2 + 2 * rb + 4);
            float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (NP == 2) a4 = *reinterpret_cast<const float4*>(h3a + rb);
            const float bq[4] = {b4.x, b4.y, b4.z, b4.w}, cq[4] = {c4.x, c4.y, c4.z, c4.w}, lq[4] = {l4.x, l4.y, l4.z, l4.w};
            const float aq[4] = {a4.x, a4.y, a4.z, a4.w};
            const float w0q[4] = {wA.x, wA.z, wB.x, wB.z}, w1q[4] = {wA.y, wA.w, wB.y, wB.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = 4 * q + p;
                float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = __fmaf_rn(oms, NP == 2 ? acc[i][j][r] * aq[p] : acc[i][j][r], bq[p]);
                    const float pre = dec_l0_pre(w0q[p], w1q[p], cq[p], lq[p], x0[j], x1[j]);
                    const float u = pre > 0.f ? t : t * slope;
                    rs0 = __fmaf_rn(u, gyv[j], rs0);
                    rs1 = __fmaf_rn(u, gx0[j], rs1);
                    rs2 = __fmaf_rn(u, gx1[j], rs2);
                    gsum[j][0] = __fmaf_rn(w0q[p], u, gsum[j][0]);
                    gsum[j][1] = __fmaf_rn(w1q[p], u, gsum[j][1]);
                }
                rs0 = half_wave_sum_hi(rs0);
                rs1 = half_wave_sum_hi(rs1);
                rs2 = half_wave_sum_hi(rs2);
                if ((lane & 31) == 16) {
                    float* pp = part + ((long)tile_n * M + m0 + rb + p) * 3;
                    pp[0] = rs0;
                    pp[1] = rs1;
                    pp[2] = rs2;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gsum[j][0] *= gyv[j];
        gsum[j][1] *= gyv[j];
    }
}

// Lean epilogue of a PLAIN hidden layer that stores its output (round 6): forward  Y = act(W X + b)  and data gradient
// dX = act'(aux) . (W^T dpre)  of every hidden decoder layer that is not one of the fused special cases above -- the layers of
// deep decoders (galaxy: four), of Fourier decoders, of residual-free stacks with n_out > 1.  The generic epilogue took 203 us of
// a 492 us launch at the galaxy shape (ablations -DTVAE_ABL=8 / =4: 89 us of vector work around 114 us of stores); here: row
// constants four at a time (ds_read_b128), the h3 factors folded into the bias FMA, wave-uniform row pointers + one 32-bit lane
// offset, no per-element option tests.  Full 512-row tiles only (M % 512 == 0).  MASKA: multiply by LeakyReLU'(aux) (the data
// gradient's saved activation); ACT: LeakyReLU on the result (forward).  vmax: max |stored value| (Epilogue.amax_out).
// MB: the two-valued implicit gradient (VirtGrad.csum; dense_x6_kernel<5>: the 0 / 1 operand has no column scale):
// value = gy[n] (oms acc + slope csum[m]), bsm = slope csum.
template <int NP, bool ACT, bool MASKA, bool MB = false>
__device__ __forceinline__ void dense_x6_epilogue_lean_store(f32x16 (&acc)[2][4], float* __restrict__ C, long ldc,
                                                             const float* __restrict__ aux, long ldaux, long n0, const float* bsm,
                                                             const float* h3a, const float* h3x, float slope, int wave, int lane,
                                                             int m0, float& vmax, const float* gyv = nullptr, float oms = 0.f) {
    float ixv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ixv[j] = (NP == 2 && !MB) ? h3x[j * 32 + (lane & 31)] : 1.f;
    const int half = lane >> 5;
    const unsigned loff = (unsigned)((4 * half * ldc + (lane & 31)) * 4);          // bytes (host: 8 ldc floats fit 2^31 bytes)
    const unsigned aoff = (unsigned)((4 * half * ldaux + (lane & 31)) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = wave * 64 + i * 32 + 8 * q;                   // rows rb + 4 half + p <-> registers r = 4 q + p
            const float4 b4 = *reinterpret_cast<const float4*>(bsm + rb + 4 * half);
            float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (NP == 2) a4 = *reinterpret_cast<const float4*>(h3a + rb + 4 * half);
            const float bq[4] = {b4.x, b4.y, b4.z, b4.w}, aq[4] = {a4.x, a4.y, a4.z, a4.w};
            float av[4][4];
            if (MASKA) {                                 // the 16 mask values of this row group before any of its stores
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const char* arow = reinterpret_cast<const char*>(aux + (long)(m0 + rb + p) * ldaux + n0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[p][j] = __builtin_nontemporal_load(reinterpret_cast<const float*>(arow + aoff) + j * 32);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                char* rowp = reinterpret_cast<char*>(C + (long)(m0 + rb + p) * ldc + n0);        // wave uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v;
                    if (MB) v = gyv[j] * __fmaf_rn(oms, NP == 2 ? acc[i][j][4 * q + p] * aq[p] : acc[i][j][4 * q + p], bq[p]);
                    else v = NP == 2 ? __fmaf_rn(acc[i][j][4 * q + p] * ixv[j], aq[p], bq[p]) : acc[i][j][4 * q + p] + bq[p];
                    if (ACT) v = fmaxf(v, v * slope);
                    if (MASKA) v *= av[p][j] > 0.f ? 1.f : slope;
                    vmax = fmaxf(vmax, fabsf(v));
                    __builtin_nontemporal_store(v, reinterpret_cast<float*>(rowp + loff) + j * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);           // one row group at a time
        }
}

// (defined with dense_x6_plain4_kernel below: scale + store of a column-tiled output, whole row tiles)
template <int NP>
__device__ __forceinline__ void dense_x6_epilogue_store(f32x16 (&acc)[2][4], float* C, long ldc, long coff, const float* h3a,
                                                        const float* h3x, int wave, int lane, int m0);

// Tile 512 x 128: eight waves stacked along the rows (64 x 128 each: 2 x 4 MFMA tiles, 48 MFMAs per 16-k step), two per
// SIMD, so one wave's split arithmetic, LDS traffic and load waits run under the other's MFMAs.  Every k-value of X is
// split once per 512 output rows (4 per thread per step).
struct DenseBatch {        // batched launch: row tile t belongs to problem t / tiles_per_batch (0 = not batched)
    int tiles_per_batch;
    long x_stride;         // floats between the X operands of consecutive problems
    long c_stride;         // floats between the outputs (and aux / residual operands) of consecutive problems
};

// XV: 0 = X is read from memory, 1 = implicit gradient operand (VirtGrad), 2 = implicit first-layer activation (VirtAct),
//     3 = two-valued implicit gradient (VirtGrad.csum: X is the saved activation H, the operand is [H > 0]),
//     5 = 3 with the operand taken from the stored sign bits (VirtGrad.bits; X is not read) + the row sums
//         sum_n gy[n] [H[m][n] > 0] per column tile (rpart[(m * (N/128) + tile) * 2], the second word is not written)
//     4 = 3 + the row sums of H against gy per column tile (VirtGrad.rpart): the threads that load H for the operand
//         also form gy[n] [H > 0] and gy[n] H, reduce them over their wave's 64 columns (wave_sum8: 8 values per step)
//         and leave them in LDS; a 2.1 GB pass of its own over H (tvae_dec_out_bwd) is then not needed
// EPI (compile time: the lean epilogues must not share a kernel with the generic one -- as run-time branches of ONE kernel
// they made the register allocator spill 152 registers per lane around the k-loop and the launch 10 % slower):
//   0 = generic epilogue (every option a uniform run-time branch)
//   1 = dense_x6_epilogue_lean: output not stored, fused column dot + sign bits, LeakyReLU, ONE full 512-row tile (M == 512)
//   2 = dense_x6_epilogue_lean_in: two-valued data gradient, fused first-layer backward with the recomputed mask, M == 512
//   4 = dense_x6_epilogue_store: the spectral contraction's lean store (no bias / activation, column-tiled output, whole row tiles)
//       for the problems that take THIS tile (reductions > 256: several input channels, wide frames) -- round 6
//   3 = dense_x6_epilogue_lean_store: plain stored output, bias + LeakyReLU | none (forward) or the LeakyReLU mask of a saved
//       activation (data gradient), whole 512-row tiles; Epilogue.amax_out supported
// The host picks the instance (abi_dense_x6.hip: launch_dense_x6) when the call has exactly that shape.
template <int XV, int NP, int EPI = 0>
static __global__ __launch_bounds__(DX6_THREADS, 2)
void dense_x6_kernel(const uint4* __restrict__ A3, const float* __restrict__ X, long ldx, Epilogue ep, int M, int Mpad,
                     int N, int K, int K8pad, TileMap tm, DenseBatch bt, ColDot cd, InTail it, VirtGrad vg, VirtAct va,
                     H3Scale hs) {
    constexpr bool VIRT = XV == 1, MASKB = XV == 3 || XV == 4 || XV == 5, RSUM = XV == 4 || XV == 5, FROMBITS = XV == 5;
    __shared__ __attribute__((aligned(16))) uint4 Bs[2 * 3 * 2 * 128];    // [stage][part][octet half][n]
    __shared__ __attribute__((aligned(16))) float bsm[DX6_ROWS];
    __shared__ __attribute__((aligned(16))) float wsm_[2 * DX6_ROWS];
    __shared__ float vwo_[XV == 1 ? 512 : (XV == 2 ? 2048 : 1)];   // tables of the implicit operand (K <= 512)
    __shared__ __attribute__((aligned(16))) float cbm_[2 * DX6_ROWS];     // (bc, lb) rows of the recomputed mask operand (InTail.bc)
    __shared__ float rsm_[RSUM ? 2 * 2 * DX6_ROWS + 8 : 1];   // [column half of the tile][row of H][2]: row sums (VirtGrad.rpart) + a dump slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n, split_unused;
    if (!tm.decode(blockIdx.x, tile_m, tile_n, split_unused)) return;
    // batched: the rows of all problems are stacked in A3 (Mpad = total padded rows, M = rows per problem)
    int batch = 0;
    if (bt.tiles_per_batch > 0) {
        batch = tile_m / bt.tiles_per_batch;
        X += batch * bt.x_stride;
        ep.C += batch * bt.c_stride;
        if (ep.aux) ep.aux += batch * bt.c_stride;
        if (ep.res) ep.res += batch * bt.c_stride;
    }
    const int m0g = tile_m * DX6_ROWS;                                   // row offset inside A3
    const int m0 = m0g - batch * bt.tiles_per_batch * DX6_ROWS, n0 = tile_n * 128;   // row offset inside the problem
    const int khalf = lane >> 5;
    const int nk = K8pad >> 1;
    // h3 arithmetic: scale of the streamed operand.  The recomputed first-layer activation (XV == 2) has no producer that
    // could have measured it: hs.amax_x then holds three maxima {max |x'|, max_k (|wc[k][0]| + |wc[k][1]|), max |bc + lb|}
    // (dec_l0_bound_kernel) whose combination bounds every |act(pre)| <= |pre|.  The 0 / 1 operand needs none.
    // Round 4: one scale per ROW of A (h3a_: their inverses for the epilogue) and, where the operand's producer measured it,
    // one per column group of X (an image; h3x_: the inverse of every column of this tile).
    __shared__ __attribute__((aligned(16))) float h3a_[NP == 2 ? DX6_ROWS : 4];
    __shared__ float h3x_[NP == 2 ? 128 : 1];
    float sx = 1.f;
    if (NP == 2) {
        h3a_[tid] = h3_inv(h3_scale(hs.amax_a[hs.a_rows ? m0g + tid : 0]));
        if (!MASKB) {
            if (XV == 2) sx = h3_scale(__fmaf_rn(hs.amax_x[1], hs.amax_x[0], hs.amax_x[2]));
            else if (hs.x_group > 0) sx = h3_scale(hs.amax_x[(long)batch * hs.x_bstride + min((n0 + (tid & 127)) / hs.x_group, hs.x_bstride - 1)]);
            else sx = h3_scale(hs.amax_x[0]);
            if (tid < 128) h3x_[tid] = h3_inv(sx);
        }
    }
    if (MASKB) bsm[tid] = (m0 + tid) < M ? vg.slope * vg.csum[m0 + tid] : 0.f;
    else bsm[tid] = (ep.bias && (m0 + tid) < M) ? ep.bias[(m0 + tid) >> ep.bias_shift] : 0.f;
    if (it.xr) {
        wsm_[2 * tid] = (m0 + tid) < M ? it.wc[2 * (m0 + tid)] : 0.f;
        wsm_[2 * tid + 1] = (m0 + tid) < M ? it.wc[2 * (m0 + tid) + 1] : 0.f;
    } else {
        wsm_[tid] = (cd.w && (m0 + tid) < M) ? cd.w[m0 + tid] : 0.f;
    }
    const float* wsm = (cd.w && !it.xr) ? wsm_ : nullptr;
    if (it.bc) {
        const int img_ = n0 / it.Np;
        cbm_[tid] = (m0 + tid) < M ? it.bc[m0 + tid] : 0.f;
        cbm_[DX6_ROWS + tid] = (it.lb && (m0 + tid) < M) ? it.lb[(long)img_ * M + m0 + tid] : 0.f;
    }

    // A cells of this lane: fragment i (rows 64*wave + 32*i + lane&31), part p, octet 2t + khalf
    const long part_cells = (long)K8pad * Mpad;
    const uint4* a_ptr = A3 + (long)khalf * Mpad + m0g + 64 * wave + (lane & 31);
    auto load_a = [&](int t, Cell16 (&a)[2][3]) {
        const uint4* q = a_ptr + (long)(2 * t) * Mpad;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[i][p].u = q[p * part_cells + i * 32];
    };
    // B build role: k-quad kq (4 consecutive k = half a cell), column nb
    const int kq = tid >> 7, nb = tid & 127;
    const float* x_col = X + n0 + nb;
    // VIRT (compile time, so that the plain instance keeps its straight-line load stream): the operand is the saved
    // activation and the gradient wo[k] * gy[n] * act'(h) is formed when the cells are built (store_b)
    const float vg_g = (VIRT || RSUM) ? vg.gy[n0 + nb] : 0.f;
    if (VIRT) {
        if (tid < K && tid < 512) vwo_[tid] = vg.wo[tid];
    }
    float va_x0 = 0.f, va_x1 = 0.f;
    if (XV == 2) {                                       // tables (w0, w1, bc, lb) of the recomputed activation, K <= 512
        const int img_ = n0 / va.Np;
        // (rows k >= K of the table are zero: their values meet all-zero weight cells and only have to be finite, so the
        // loop below needs no k < K test)
        vwo_[tid] = tid < K ? va.wc[2 * tid] : 0.f;
        vwo_[512 + tid] = tid < K ? va.wc[2 * tid + 1] : 0.f;
        vwo_[1024 + tid] = tid < K ? va.bc[tid] : 0.f;
        vwo_[1536 + tid] = (va.lb && tid < K) ? va.lb[(long)img_ * K + tid] : 0.f;
        va_x0 = va.xr[2 * (long)(n0 + nb)];
        va_x1 = va.xr[2 * (long)(n0 + nb) + 1];
    }
    const bool vwo_lds = K <= 512;
    const bool va_lrelu = XV == 2 && va.act == ACT_LRELU;
    const unsigned* bit_col = FROMBITS ? vg.bits + ((n0 + nb) >> 5) : nullptr;      // this thread's word of a row of bits
    const long bitw_ = (long)(N >> 5);
    auto load_x = [&](int t, float (&x)[4]) {
        if (XV == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = 0.f;
        } else if (FROMBITS) {
            // the word that holds bit (row, column nb): 32 lanes share it (one 4-byte request per row and half wave); the
            // value travels as the 0 / 1 it stands for (clamped rows meet all-zero weight cells)
            const int kb = 16 * t + 4 * kq;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = __uint_as_float(bit_col[(long)min(kb + j, K - 1) * bitw_]);
        } else {
            // unconditional loads from a CLAMPED row and no masking: rows k >= K meet all-zero weight cells, so any
            // finite value will do (the clamped row is real data).  A load under an exec-mask branch, or a select on its
            // result, makes the compiler wait for it on the spot -- these loads are issued two steps ahead on purpose
            const int kb = 16 * t + 4 * kq;
#pragma unroll
            for (int j = 0; j < 4; ++j)               // the saved activation of the two-valued forms is read exactly once
                x[j] = MASKB ? __builtin_nontemporal_load(x_col + (long)min(kb + j, K - 1) * ldx) : x_col[(long)min(kb + j, K - 1) * ldx];
        }
    };
    auto virt_x = [&](int t, float (&x)[4]) {
        if (VIRT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * t + 4 * kq + j;
                x[j] = (k < K) ? virt_value(vg, x[j], vwo_lds ? vwo_[k & 511] : vg.wo[k], vg_g) : 0.f;
            }
        }
        if (XV == 2) {
            // LeakyReLU (every reference configuration) straight-line: the generic form put an exec-mask branch, a scalar
            // switch on the activation and tanhf's own branches around EACH of the four values of a step -- inside the
            // k-loop, between the MFMAs (found in the ISA, round 3)
            if (va_lrelu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = (16 * t + 4 * kq + j) & 511;
                    const float pre = dec_l0_pre(vwo_[k], vwo_[512 + k], vwo_[1024 + k], vwo_[1536 + k], va_x0, va_x1);
                    x[j] = pre > 0.f ? pre : pre * va.slope;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = (16 * t + 4 * kq + j) & 511;
                    const float pre = dec_l0_pre(vwo_[k], vwo_[512 + k], vwo_[1024 + k], vwo_[1536 + k], va_x0, va_x1);
                    x[j] = act_apply(pre, va.act, va.slope);
                }
            }
        }
    };
    auto store_b = [&](int stage, const float (&x)[4]) {
        uint2* dst = reinterpret_cast<uint2*>(Bs + stage * 768 + (kq >> 1) * 128 + nb) + (kq & 1);
        if (MASKB) {                                     // [H > 0]: 1.0 = 0x3f80 (bf16) / 0x3c00 (fp16), one part (rows k >= K meet zero weights)
            bool p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = FROMBITS ? ((__float_as_uint(x[j]) >> (nb & 31)) & 1u) != 0u : x[j] > 0.f;
            dst[0] = make_uint2((p[0] ? OneBits<NP>::lo : 0u) | (p[1] ? OneBits<NP>::hi : 0u),
                                (p[2] ? OneBits<NP>::lo : 0u) | (p[3] ? OneBits<NP>::hi : 0u));
            return;
        }
        if (NP == 1) {
            dst[0] = make_uint2(bf16_pair(x[0], x[1]), bf16_pair(x[2], x[3]));
            return;
        }
        if (NP == 2) {
            unsigned hw[2], lw[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) split2h_pair(x[2 * q] * sx, x[2 * q + 1] * sx, hw[q], lw[q]);
            dst[0] = make_uint2(hw[0], hw[1]);
            dst[2 * 256] = make_uint2(lw[0], lw[1]);
            return;
        }
        unsigned hw[2], mw[2], lw[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) split3_pair(x[2 * q], x[2 * q + 1], hw[q], mw[q], lw[q]);
        dst[0] = make_uint2(hw[0], hw[1]);
        dst[2 * 256] = make_uint2(mw[0], mw[1]);
        dst[2 * 512] = make_uint2(lw[0], lw[1]);
    };

    // RSUM: this thread holds H[16 t + 4 kq + j][n0 + nb], j < 4, for step t; the two waves with the same kq cover the
    // tile's 128 columns.  Lane v < 8 of each wave stores the wave's total of value v = 2 j + (0: gy [H > 0], 1: gy H);
    // every (t, kq, j) row is visited once, so the slots are plain stores (rows >= K are never read back).
    auto row_sums = [&](int t, bool real, const float (&x)[4]) {
        if (!RSUM) return;
        // `real` = false: the register set holds a step that does not exist (the loop's clamped prefetch past the end
        // re-reads an EARLIER step): its sums go to the dump slot -- branch free, like the rest of the operand path
        const int slot = real ? ((wave & 1) * DX6_ROWS + ((16 * t + 4 * kq) & (DX6_ROWS - 1))) * 2 : 2 * 2 * DX6_ROWS;
        if (FROMBITS) {                                  // one value per row: gy [bit]; lane v < 4 stores row j = v
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ((__float_as_uint(x[j]) >> (nb & 31)) & 1u) ? vg_g : 0.f;
            const float tot = wave_sum4(a, lane);
            if (lane < 4) rsm_[slot + 2 * lane] = tot;
            return;
        }
        float a[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[2 * j] = x[j] > 0.f ? vg_g : 0.f;
            a[2 * j + 1] = vg_g * x[j];
        }
        const float tot = wave_sum8(a, lane);
        if (lane < 8) rsm_[slot + lane] = tot;
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register sets for the weight cells and two for the X values, alternating by step parity (the loop is unrolled by
    // two so that the set is a compile-time choice): a step issues the loads of the NEXT step's cells and of the X values
    // THREE steps ahead into the set it has just finished with, and nothing is copied -- a register copy at the end of a
    // step would make the wave wait there for loads it does not need yet.
    Cell16 afA[2][3], afB[2][3];
    float xA[4], xB[4];                                  // X values of the next even / odd step
    if (XV != 0) __syncthreads();                       // vwo_ is read by the prologue's virt_x
    {
        float x0[4];
        load_x(0, x0);
        load_a(0, afA);
        load_x(nk > 1 ? 1 : 0, xB);
        load_x(nk > 2 ? 2 : 0, xA);
        virt_x(0, x0);
        store_b(0, x0);
        row_sums(0, true, x0);
    }
    __syncthreads();
    auto step = [&](int t, Cell16 (&afc)[2][3], Cell16 (&afn)[2][3], float (&xn)[4]) {
        const int cur = t & 1;
        load_a(t + 1 < nk ? t + 1 : t, afn);            // weight cells of the next step
        const uint4* bs = Bs + cur * 768 + khalf * 128 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MASKB) {                                 // exact 0 / 1 operand: the three weight parts against ONE cell
                Cell16 b0;
                b0.u = bs[j * 32];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p) mfma_part<NP>(acc[i][j], afc[i][p], b0);
            } else {
                Cell16 bf[3];
#pragma unroll
                for (int p = 0; p < NP; ++p) bf[p].u = bs[p * 256 + j * 32];
                mfma_np<NP>(acc[0][j], afc[0], bf);
                mfma_np<NP>(acc[1][j], afc[1], bf);
            }
            if (j == 1) {                                // cells of step t+1, then its buffer takes the values of step t+3
                virt_x(t + 1 < nk ? t + 1 : t, xn);
                store_b(cur ^ 1, xn);
                row_sums(t + 1, t + 1 < nk, xn);
                load_x(t + 3 < nk ? t + 3 : t, xn);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    // both steps of a trip are unconditional: a step under `if` lets the compiler sink the loads issued for it into the
    // branch, i.e. to just before their use
    int tt = 0;
    for (; tt + 1 < nk; tt += 2) {
        step(tt, afA, afB, xB);
        step(tt + 1, afB, afA, xA);
    }
    if (nk & 1) step(nk - 1, afA, afB, xB);
    if (NP == 2 && EPI == 0) h3_unscale_rc<!MASKB>(acc, h3a_, h3x_, wave, lane);       // (tables written before the k-loop's barriers)
    if (RSUM) {                                          // (the loop's last barrier made every slot visible)
        if (tid < K) {
            float* rp = vg.rpart + ((long)tid * (N >> 7) + tile_n) * 2;   // [row][tile][2]: a row's partials are contiguous
            rp[0] = rsm_[tid * 2] + rsm_[(DX6_ROWS + tid) * 2];
            if (!FROMBITS) rp[1] = rsm_[tid * 2 + 1] + rsm_[(DX6_ROWS + tid) * 2 + 1];
        }
    }
    // epilogue specialised on (activation, mask, residual): no per-element branches
    const bool res = ep.res != nullptr;
    float ysum[4] = {0.f, 0.f, 0.f, 0.f};
    float gsum[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float gyv[4] = {0.f, 0.f, 0.f, 0.f};
    if (MASKB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gyv[j] = vg.gy[n0 + j * 32 + (lane & 31)];
    }
    const float oms = 1.f - vg.slope;
    float vmax = 0.f;
#define TVAE_DX6_EPI(A_, M_, R_, V_)                                                                                    \
    dense_x6_epilogue<A_, M_, R_, V_, MASKB>(acc, ep, bsm, m0, n0, M, wave, lane, wsm, ysum, it, wsm_, gsum, tile_n, cbm_, \
                                             gyv, oms, cd.bits, (long)(N >> 5), vmax)
#define TVAE_DX6_EPI_R(A_, M_, V_) \
    do { if (res) TVAE_DX6_EPI(A_, M_, true, V_); else TVAE_DX6_EPI(A_, M_, false, V_); } while (0)
    if (EPI == 4) {      // (M is a multiple of the tile: a tile is all real rows, or all rows that pad the stacked problems to 512)
        if (m0 < M) dense_x6_epilogue_store<NP>(acc, ep.C, ep.ldc, (long)tile_n * ep.ctile, h3a_, h3x_, wave, lane, m0);
    } else
    if (EPI == 3 && MASKB) {     // two-valued data gradient whose result is stored (Fourier decoders: no fused first-layer backward)
        dense_x6_epilogue_lean_store<NP, false, true, true>(acc, ep.C, ep.ldc, ep.aux, ep.ldaux, n0, bsm, h3a_, h3x_, ep.slope, wave, lane,
                                                            m0, vmax, gyv, oms);
    } else
    if (EPI == 3) {
        if (ep.mask == ACT_LRELU)
            dense_x6_epilogue_lean_store<NP, false, true>(acc, ep.C, ep.ldc, ep.aux, ep.ldaux, n0, bsm, h3a_, h3x_, ep.slope, wave, lane, m0, vmax);
        else if (ep.act == ACT_LRELU)
            dense_x6_epilogue_lean_store<NP, true, false>(acc, ep.C, ep.ldc, nullptr, 0, n0, bsm, h3a_, h3x_, ep.slope, wave, lane, m0, vmax);
        else
            dense_x6_epilogue_lean_store<NP, false, false>(acc, ep.C, ep.ldc, nullptr, 0, n0, bsm, h3a_, h3x_, ep.slope, wave, lane, m0, vmax);
    } else
    if (EPI == 2) {
        float x0_[4], x1_[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long nn = n0 + j * 32 + (lane & 31);
            x0_[j] = it.xr[2 * nn];
            x1_[j] = it.xr[2 * nn + 1];
        }
        dense_x6_epilogue_lean_in<NP>(acc, bsm, wsm_, cbm_, h3a_, ep.slope, oms, wave, lane, m0, M, tile_n, gyv, x0_, x1_, it.part,
                                      gsum);
    } else if (EPI == 1) {
        dense_x6_epilogue_lean<NP>(acc, bsm, wsm_, h3a_, h3x_, ep.slope, wave, lane, m0, n0, ysum, cd.bits, (long)(N >> 5));
    } else
    if (TVAE_ABL & 8) {                                  // ablation: no epilogue (keep the accumulators alive)
        float t_ = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t_ += acc[i][j][r];
        if (t_ == 123.456f && cd.y) cd.y[n0] = t_;
    } else
    if (ep.mask == ACT_NONE) {
        if (ep.act == ACT_LRELU) TVAE_DX6_EPI_R(ACT_LRELU, ACT_NONE, false);
        else if (ep.act == ACT_TANH) TVAE_DX6_EPI_R(ACT_TANH, ACT_NONE, false);
        else TVAE_DX6_EPI_R(ACT_NONE, ACT_NONE, false);
    } else if (ep.mask == ACT_LRELU) {
        if (it.bc) TVAE_DX6_EPI_R(ACT_NONE, ACT_LRELU, true); else TVAE_DX6_EPI_R(ACT_NONE, ACT_LRELU, false);
    } else {
        if (it.bc) TVAE_DX6_EPI_R(ACT_NONE, ACT_TANH, true); else TVAE_DX6_EPI_R(ACT_NONE, ACT_TANH, false);
    }
#undef TVAE_DX6_EPI_R
#undef TVAE_DX6_EPI
    if ((EPI == 0 || EPI == 3) && ep.amax_out) {         // one atomic per wave at most (most find a larger value already there)
        const float m_ = h3_wave_max(vmax);
        if (lane == 0) h3_atomic_amax(ep.amax_out, m_);
    }
    if (it.xr) {
        float* cds = reinterpret_cast<float*>(Bs);          // [wave][128][2]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float v = gsum[j][k] + __shfl_xor(gsum[j][k], 32, 64);
                if (lane < 32) cds[(wave * 128 + j * 32 + lane) * 2 + k] = v;
            }
        __syncthreads();
        if (tid < 256) {
            float g = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) g += cds[w * 256 + tid];
            it.gxr[2 * (long)n0 + tid] = g;
        }
    } else if (cd.w) {
        // the eight waves hold disjoint rows of the same 128 columns: lane halves first, then waves through LDS
        float* cds = reinterpret_cast<float*>(Bs);          // the B stages are free after the k-loop
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = ysum[j] + __shfl_xor(ysum[j], 32, 64);
            if (lane < 32) cds[wave * 128 + j * 32 + lane] = v;
        }
        __syncthreads();
        if (tid < 128) {
            float y = cd.b ? cd.b[0] : 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) y += cds[w * 128 + tid];
            cd.y[n0 + tid] = y;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Plain instance with a 256 x 128 tile and FOUR waves for SHORT reductions (the spectral contraction of the
// frequency-domain convolution: K = 2 L Cin = 192, twelve 16-k steps per tile).  With so few steps the tile's output
// (here 128 KB of T) takes as long to drain as its MFMAs take to run, and one 8-wave workgroup owns a whole CU: all eight
// waves store at the same time while the matrix pipe idles.  Two 4-wave workgroups per CU run out of phase -- one
// workgroup's stores drain under the other's k-loop.  Same arithmetic, cells, B stage and epilogue as dense_x6_kernel<0>;
// a thread builds two k-quads of the B stage per step instead of one.
// ------------------------------------------------------------------------------------------
constexpr int DX4_THREADS = 256;
constexpr int DX4_ROWS = 256;

// Lean epilogue of the spectral contraction (no bias, no activation, every row of the tile real): scale, store.  The generic
// epilogue's per-row work -- options tested per element, a 64-bit address per row and column group -- cost 0.54 ms of the
// 1.47 ms launch (ablation without any epilogue: 0.94 ms, profiles/README.md round 4).  Here a row's base address is wave
// uniform (SGPRs), the lane's byte offset one register for the whole tile, and the h3 factors fold into one product per
// element and column group ((acc ix) ia: exact powers of two).
template <int NP>
__device__ __forceinline__ void dense_x6_epilogue_store(f32x16 (&acc)[2][4], float* C, long ldc, long coff, const float* h3a,
                                                        const float* h3x, int wave, int lane, int m0) {
    float ixv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ixv[j] = NP == 2 ? h3x[j * 32 + (lane & 31)] : 1.f;
    const int half = lane >> 5;
    const unsigned loff = (unsigned)((4 * half * ldc + (lane & 31)) * 4);          // bytes (host: 8 ldc floats fit 2^31 bytes)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = wave * 64 + i * 32 + 8 * q;                   // rows rb + 4 half + p <-> registers r = 4 q + p
            float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (NP == 2) a4 = *reinterpret_cast<const float4*>(h3a + rb + 4 * half);
            const float aq[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                char* rowp = reinterpret_cast<char*>(C + (long)(m0 + rb + p) * ldc + coff);      // wave uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = NP == 2 ? (acc[i][j][4 * q + p] * ixv[j]) * aq[p] : acc[i][j][4 * q + p];
#ifdef TVAE_T_CACHED      // experiment (profiles/tools/batch_sweep_dft.sh): default cache policy for T -- does the Infinity Cache keep it?
                    reinterpret_cast<float*>(rowp + loff)[j * 32] = v;
#else
                    __builtin_nontemporal_store(v, reinterpret_cast<float*>(rowp + loff) + j * 32);
#endif
                }
            }
        }
}

// The same for the bf16 STORAGE of the one-part throughput mode (round 4; TVAE_GEMM=bf16 only, never the fp32-class
// arithmetics): the accumulators are rounded to bf16 (RNE) and T is written as 2-byte elements in the same element layout,
// half the bytes for the GEMM to write and for the contraction over fx (dft_out_ring_kernel<.., T16>) to read.
__device__ __forceinline__ void dense_x6_epilogue_store_bf16(f32x16 (&acc)[2][4], unsigned short* C, long ldc, long coff,
                                                             int wave, int lane, int m0) {
    const int half = lane >> 5;
    const unsigned loff = (unsigned)((4 * half * ldc + (lane & 31)) * 2);          // bytes
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = wave * 64 + i * 32 + 8 * q;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                char* rowp = reinterpret_cast<char*>(C + (long)(m0 + rb + p) * ldc + coff);      // wave uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][4 * q + p]);
                    __builtin_nontemporal_store(b, reinterpret_cast<unsigned short*>(rowp + loff) + j * 32);
                }
            }
        }
}

// EPI = 1: dense_x6_epilogue_store (host: no bias / activation, column-tiled output, rows per problem a multiple of 256)
// EPI = 2: dense_x6_epilogue_store_bf16 (the same shape, NP == 1, ep.C points to 2-byte elements)
template <int NP, int EPI = 0>
static __global__ __launch_bounds__(DX4_THREADS, 2)
void dense_x6_plain4_kernel(const uint4* __restrict__ A3, const float* __restrict__ X, long ldx, Epilogue ep, int M, int Mpad,
                            int N, int K, int K8pad, TileMap tm, DenseBatch bt, H3Scale hs) {
    __shared__ __attribute__((aligned(16))) uint4 Bs[2 * 3 * 2 * 128];    // [stage][part][octet half][n]
    __shared__ float bsm[DX4_ROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n, split_unused;
    if (!tm.decode(blockIdx.x, tile_m, tile_n, split_unused)) return;
    int batch = 0;
    if (bt.tiles_per_batch > 0) {
        batch = tile_m / bt.tiles_per_batch;
        X += batch * bt.x_stride;
        ep.C += batch * bt.c_stride;
    }
    const int m0g = tile_m * DX4_ROWS;
    const int m0 = m0g - batch * bt.tiles_per_batch * DX4_ROWS, n0 = tile_n * 128;
    const int khalf = lane >> 5;
    const int nk = K8pad >> 1;
    bsm[tid] = (ep.bias && (m0 + tid) < M) ? ep.bias[(m0 + tid) >> ep.bias_shift] : 0.f;
    // h3: power-of-two scales -- per row of A, per column group (image) of the streamed operand (see H3Scale)
    __shared__ __attribute__((aligned(16))) float h3a_[NP == 2 ? DX4_ROWS : 4];
    __shared__ float h3x_[NP == 2 ? 128 : 1];
    float sx = 1.f;
    if (NP == 2) {
        h3a_[tid] = h3_inv(h3_scale(hs.amax_a[hs.a_rows ? m0g + tid : 0]));
        if (hs.x_group > 0) sx = h3_scale(hs.amax_x[(long)batch * hs.x_bstride + min((n0 + (tid & 127)) / hs.x_group, hs.x_bstride - 1)]);
        else sx = h3_scale(hs.amax_x[0]);
        if (tid < 128) h3x_[tid] = h3_inv(sx);
    }

    const long part_cells = (long)K8pad * Mpad;
    const uint4* a_ptr = A3 + (long)khalf * Mpad + m0g + 64 * wave + (lane & 31);
    auto load_a = [&](int t, Cell16 (&a)[2][3]) {
        const uint4* q = a_ptr + (long)(2 * t) * Mpad;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[i][p].u = q[p * part_cells + i * 32];
    };
    // B build role: k-quads kq and kq + 2 (4 consecutive k each = half a cell), column nb
    const int kq = tid >> 7, nb = tid & 127;
    const float* x_col = X + n0 + nb;
    auto load_x = [&](int t, float (&x)[2][4]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kb = 16 * t + 4 * (kq + 2 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[h][j] = x_col[(long)min(kb + j, K - 1) * ldx];     // clamped rows meet zero weights
        }
    };
    auto store_b = [&](int stage, const float (&x)[2][4]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint2* dst = reinterpret_cast<uint2*>(Bs + stage * 768 + h * 128 + nb) + kq;
            if (NP == 1) {
                dst[0] = make_uint2(bf16_pair(x[h][0], x[h][1]), bf16_pair(x[h][2], x[h][3]));
                continue;
            }
            if (NP == 2) {
                unsigned hw[2], lw[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) split2h_pair(x[h][2 * q] * sx, x[h][2 * q + 1] * sx, hw[q], lw[q]);
                dst[0] = make_uint2(hw[0], hw[1]);
                dst[2 * 256] = make_uint2(lw[0], lw[1]);
                continue;
            }
            unsigned hw[2], mw[2], lw[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) split3_pair(x[h][2 * q], x[h][2 * q + 1], hw[q], mw[q], lw[q]);
            dst[0] = make_uint2(hw[0], hw[1]);
            dst[2 * 256] = make_uint2(mw[0], mw[1]);
            dst[2 * 512] = make_uint2(lw[0], lw[1]);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Cell16 afA[2][3], afB[2][3];
    float xA[2][4], xB[2][4];
    {
        float x0[2][4];
        load_x(0, x0);
        load_a(0, afA);
        load_x(nk > 1 ? 1 : 0, xB);
        load_x(nk > 2 ? 2 : 0, xA);
        store_b(0, x0);
    }
    __syncthreads();
    auto step = [&](int t, Cell16 (&afc)[2][3], Cell16 (&afn)[2][3], float (&xn)[2][4]) {
        const int cur = t & 1;
        load_a(t + 1 < nk ? t + 1 : t, afn);
        const uint4* bs = Bs + cur * 768 + khalf * 128 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Cell16 bf[3];
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[p].u = bs[p * 256 + j * 32];
            mfma_np<NP>(acc[0][j], afc[0], bf);
            mfma_np<NP>(acc[1][j], afc[1], bf);
            if (j == 1) {
                store_b(cur ^ 1, xn);
                load_x(t + 3 < nk ? t + 3 : t, xn);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    int tt = 0;
    for (; tt + 1 < nk; tt += 2) {
        step(tt, afA, afB, xB);
        step(tt + 1, afB, afA, xA);
    }
    if (nk & 1) step(nk - 1, afA, afB, xB);
    if (EPI == 1) {      // (M is a multiple of the tile: a tile is all real rows, or all rows that pad the stacked problems to 512)
        if (m0 < M) dense_x6_epilogue_store<NP>(acc, ep.C, ep.ldc, (long)tile_n * ep.ctile, h3a_, h3x_, wave, lane, m0);
        return;
    }
    if (EPI == 2) {      // (ep.C was advanced by the batch offset in 4-byte units above: undo half of it for 2-byte elements)
        unsigned short* c16 = reinterpret_cast<unsigned short*>(ep.C - (long)batch * bt.c_stride) + (long)batch * bt.c_stride;
        if (m0 < M) dense_x6_epilogue_store_bf16(acc, c16, ep.ldc, (long)tile_n * ep.ctile, wave, lane, m0);
        return;
    }
    if (NP == 2) h3_unscale_rc<true>(acc, h3a_, h3x_, wave, lane);
    float ysum[4] = {0.f, 0.f, 0.f, 0.f};
    float gsum[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const float gyv[4] = {0.f, 0.f, 0.f, 0.f};
    const InTail it{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1};
    float vmax_ = 0.f;                                   // (Epilogue.amax_out is not wired for this instance)
    if (TVAE_ABL & 8) {                                  // ablation: no epilogue (keep the accumulators alive)
        float t_ = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t_ += acc[i][j][r];
        if (t_ == 123.456f) ep.C[0] = t_;
    } else
    if (ep.act == ACT_LRELU)
        dense_x6_epilogue<ACT_LRELU, ACT_NONE, false, false, false>(acc, ep, bsm, m0, n0, M, wave, lane, nullptr, ysum, it, nullptr,
                                                                     gsum, tile_n, nullptr, gyv, 0.f, nullptr, 0, vmax_);
    else if (ep.act == ACT_TANH)
        dense_x6_epilogue<ACT_TANH, ACT_NONE, false, false, false>(acc, ep, bsm, m0, n0, M, wave, lane, nullptr, ysum, it, nullptr,
                                                                    gsum, tile_n, nullptr, gyv, 0.f, nullptr, 0, vmax_);
    else
        dense_x6_epilogue<ACT_NONE, ACT_NONE, false, false, false>(acc, ep, bsm, m0, n0, M, wave, lane, nullptr, ysum, it, nullptr,
                                                                    gsum, tile_n, nullptr, gyv, 0.f, nullptr, 0, vmax_);
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a dense layer in the same arithmetic:  dW[m][k] = sum_n dY[m][n] X[k][n].
// Both operands are row-major with the reduction index n contiguous, so a fragment cell (8 consecutive n of one row)
// is 32 contiguous bytes of its row.  Tile 512 (m) x 128 (k), eight waves stacked along m (64 x 128 each); the
// reduction is split over n-chunks (slab partials + the deterministic finalize of the fp32 core), 1-D XCD-aware grid
// (the k-tiles of one chunk share the dY panel in one L2).
//   * A (dY): every lane loads and splits the cells of its OWN fragments (rows 64w + 32i + lane&31, octet lane>>5)
//     straight into registers, one step ahead -- no LDS, no barrier dependence;
//   * B (X): the 128 x 16 tile of a step is split cooperatively (4 values per thread) into the [part][octet][row]
//     LDS stage of the next step, as in dense_x6_kernel.
// Requires n-chunks that are multiples of 16 and 16-byte aligned rows (checked on the host).
// ------------------------------------------------------------------------------------------
// VIRT: implicit gradient A operand (VirtGrad); XVA: implicit first-layer activation as the X operand (VirtAct)
struct ATile {             // column addressing of the streamed A operand: element (row, n) lives at
    int sh;                //   row * ldd + (n >> sh) * ts + (n & mask)
    int mask;              // plain row-major rows: {30, 0x3fffffff, 0}; the column-tiled T / S' layout of the
    long ts;               // frequency-domain convolution (conv_dft_kernels.hpp): {7, 127, tile stride}
};
constexpr ATile ATILE_PLAIN = {30, 0x3fffffff, 0};

// ------------------------------------------------------------------------------------------
// dense_wgrad_x6_kernel with EVERY streamed operand of the loop brought in by global_load_lds DMAs three steps deep
// instead of per-lane loads one step ahead.  A debug build that skipped the A loads ran 1.1-1.4 ms faster per launch
// (4.1 -> 2.8 ms): the per-lane loads were each wave's critical path (flight time of about half a step, 32 partially
// used cache lines per instruction).  Here
//   * every wave owns a ring of three slots; a slot holds what THIS wave needs for one step: its 64 rows x 16 n of the
//     A operand (4 KB, four DMA instructions: instruction g moves rows 16g..16g+15, four lanes per row = one full
//     64-byte line per row), the 16 x 16 n of the X operand its own threads split (1 KB: every lane DMAs exactly the
//     16 bytes it later reads back) -- or, for the recomputed first-layer operand, the 32 coordinates and this
//     thread's latent term -- and the 16 gy values of the implicit gradient;
//   * the 16-byte piece of A a lane fetches is XOR-swizzled with (row >> 2) & 3, so that the per-lane fragment reads
//     (row = lane & 31 (+32), two pieces) are conflict-free ds_read_b128 although the DMA destination is lane-linear;
//   * the DMAs are issued from inline asm and waited for with hand-written s_waitcnt (the compiler would put vmcnt(0)
//     in front of every LDS read behind a DMA).  There is NO ordinary vector-memory load in the loop: DMAs complete in
//     order among themselves, but a first version that mixed them with register loads and counted both in one vmcnt
//     read slots (and let late register loads land in reallocated registers) before they were complete.
//     Per step: [s_waitcnt vmcnt(NDMA): only the previous step's DMAs may be in flight] -> [DMAs of step t+3].
// Same tile, B stage, MFMA schedule and slab output as dense_wgrad_x6_kernel.
// ------------------------------------------------------------------------------------------
constexpr int WG_SLOT_BYTES = 4096 + 1024 + 256;      // A | X (or coordinates 256 + latent term 256) | gy
constexpr int WG_RING_BYTES = 8 * 3 * WG_SLOT_BYTES;

// LRF (implicit LeakyReLU gradient only): the two-valued form of VirtGrad.  act' = slope + (1 - slope) [H > 0], so
//   dW[m][k] = wo[m] * ( slope * s[k] + (1 - slope) * sum_n [H[m][n] > 0] * (gy[n] X[k][n]) ),   s[k] = sum_n gy[n] X[k][n]:
// gy[n] goes into the X values before they are split, wo[m] onto the finished accumulator rows, and the streamed A operand
// is the 0 / 1 matrix [H > 0] -- ONE exact bf16 part (1.0 = 0x3f80), built with one compare per element, and THREE MFMAs
// per product block (the single A part against the three X parts) instead of six.  s[k] is accumulated by the threads
// that build the X cells (each owns one feature row k of the tile) and joins the partial slab in the epilogue.
// LRF: 0 = off, 1 = two-valued form with [H > 0] taken from the saved activation (dY = H), 2 = from the sign bits a forward
// launch stored (VirtGrad.bits: one word per row and 32 columns, two dword DMAs per wave and step instead of four 1 KB ones)
// ABF (round 4; NP == 1, plain operands only): the A operand is STORED as bf16 (S' of the one-part mode, written by
// dft_dy_ring_kernel<.., S16>): 8 consecutive n of a row ARE a one-part cell, so the two 1 KB DMAs of a step (instruction g:
// rows 32 g .. 32 g + 31, lane -> (row lane & 31, 16-byte piece lane >> 5)) land lane-linear in exactly the order in which
// the fragments read them back -- no split, no vector-ALU work at all on this operand, half the bytes.
template <bool VIRT, bool XVA, int LRF, int NP, bool ABF = false>
static __global__ __launch_bounds__(DX6_THREADS, 2)
void dense_wgrad_x6_dma_kernel(const float* __restrict__ dY, long ldd, const float* __restrict__ X, long ldx, float* ws,
                               int M, int Kf, int N, int nchunk, TileMap tm, DenseBatch bt, long dy_stride, VirtGrad vg,
                               VirtAct va, ATile atile, H3Scale hs) {
    static_assert(!ABF || (NP == 1 && !VIRT && !XVA && LRF == 0), "bf16-stored A: one-part mode, plain operands");
    constexpr int NDMA = (ABF ? 2 : (LRF == 2 ? 2 : 4)) + (XVA ? 2 : 1) + (VIRT ? 1 : 0);   // DMA instructions per wave and step
    __shared__ __attribute__((aligned(16))) uint4 Bs[2 * 3 * 2 * 128];    // [stage][part][octet half][k row]
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_ring[];   // [wave][slot < 3][WG_SLOT_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_k, split;
    if (!tm.decode(blockIdx.x, tile_m, tile_k, split)) return;
    int batch = 0, nbatch = 1;
    if (bt.tiles_per_batch > 0) {
        batch = tile_m / bt.tiles_per_batch;
        nbatch = tm.tilesM / bt.tiles_per_batch;
        tile_m -= batch * bt.tiles_per_batch;
        dY += (ABF ? batch * dy_stride / 2 : batch * dy_stride);        // (dy_stride counts elements: 2-byte ones with ABF; even)
        X += batch * bt.x_stride;
    }
    const int m0 = tile_m * DX6_ROWS, k0 = tile_k * 128;
    const int nbeg = split * nchunk;
    const int nend = min(N, nbeg + nchunk);
    const int nk = (nend - nbeg) >> 4;
    const int khalf = lane >> 5;
    if (nk <= 0) {                                      // empty reduction slice: zero partial
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (m >= M) continue;
                float* wrow = ws + (((long)split * nbatch + batch) * M + m) * Kf + k0 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j * 32 + (lane & 31) < Kf) wrow[j * 32] = 0.f;
            }
        return;
    }

    // h3 arithmetic: power-of-two scales of the two streamed operands (the 0 / 1 operand of the two-valued form has none)
    // The X operand of the two-valued decoder form is gy[n] * (recomputed first-layer activation): hs.amax_x then holds the
    // three maxima of dec_l0_bound_kernel and max |gy| ({max |x'|, max (|w0| + |w1|), max |bc + lb|, max |gy|}).
    // Round 4: one scale per ROW of either operand (a row of dY is a row of dW, a row of X a column of dW; see H3Scale); the
    // recomputed operand of the two-valued decoder form bounds its own row k: ((|wc[k][0]| + |wc[k][1]|) max |x'| +
    // max_b |bc[k] + lb[b][k]|) max |gy|  (hs.amax_x[4 + k]: dec_l0_bound_kernel).
    auto a_scale_of = [&](int m) -> float {             // m: row of the problem, clamped by the caller
        return h3_scale(hs.amax_a[hs.a_rows ? (long)batch * hs.a_bstride + (hs.a_mod > 0 ? m % hs.a_mod : m) : 0]);
    };
    float sa[2] = {1.f, 1.f};
    if (NP == 2 && !LRF) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sa[i] = a_scale_of(min(m0 + 64 * wave + 32 * i + (lane & 31), M - 1));
    }
    // ---- DMA sources
    const float* d_ptr[4];                               // A: instruction g, row 16g + lane/4 of this wave, swizzled piece
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (ABF) {                                       // bf16 rows: instruction g < 2, row 32 g + (lane & 31), piece lane >> 5
            const int row = m0 + 64 * wave + 32 * (g & 1) + (lane & 31);
            d_ptr[g] = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(dY) +
                                                      (long)min(row, M - 1) * ldd + 8 * (lane >> 5));
        } else {
            const int row = m0 + 64 * wave + 16 * g + (lane >> 2);
            d_ptr[g] = dY + (long)min(row, M - 1) * ldd + 4 * ((lane & 3) ^ ((lane >> 4) & 3));
        }
    }
    const unsigned* b_ptr[2] = {nullptr, nullptr};       // LRF == 2: the bit row of this lane's fragment row i
    if (LRF == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            b_ptr[i] = vg.bits + (long)min(m0 + 64 * wave + 32 * i + (lane & 31), M - 1) * (N >> 5);
    }
    const int kr = tid >> 2, q4 = tid & 3;               // B build role: feature row kr, n-quad q4
    const int kx = min(k0 + kr, Kf - 1);
    const float b_ok = (k0 + kr) < Kf ? 1.f : 0.f;
    // X: this thread's own 16 bytes (4 consecutive n of its row); XVA: dword lane & 31 of the step's 32 coordinates
    const float* x_src = XVA ? va.xr + 2 * (long)nbeg + (lane & 31) : X + (long)kx * ldx + nbeg + 4 * q4;
    const float* lb_src = XVA ? (va.lb ? va.lb + kx : va.bc + kx) : nullptr;   // no latent term: a valid dummy, scaled by 0
    const float lb_on = (XVA && va.lb) ? 1.f : 0.f;
    const float* g_src = VIRT ? vg.gy + nbeg + (lane & 15) : nullptr;
    const float va_w0 = XVA ? va.wc[2 * kx] : 0.f, va_w1 = XVA ? va.wc[2 * kx + 1] : 0.f, va_bc = XVA ? va.bc[kx] : 0.f;
    float sx = 1.f;
    if (NP == 2) {
        if (XVA && LRF) sx = h3_scale(__fmaf_rn(fabsf(va_w0) + fabsf(va_w1), hs.amax_x[0], hs.amax_x[4 + kx]) * hs.amax_x[3]);
        else if (hs.x_group > 0) sx = h3_scale(hs.amax_x[(long)batch * hs.x_bstride + kx / hs.x_group]);
        else sx = h3_scale(hs.amax_x[0]);
    }
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)wg_ring +
                              (unsigned)(wave * 3 * WG_SLOT_BYTES);
    const unsigned char* ring = wg_ring + wave * 3 * WG_SLOT_BYTES;
    auto dma_step = [&](int slot, int t) {
        const int na = nbeg + 16 * t;                    // wave-uniform; a 16-wide step never straddles a column tile
        const long off = (long)(na >> atile.sh) * atile.ts + (na & atile.mask);
        const unsigned sl = ring_lds + (unsigned)(slot * WG_SLOT_BYTES);
        if (ABF) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const unsigned short* src = reinterpret_cast<const unsigned short*>(d_ptr[g]) + off;
                const unsigned dst = sl + (unsigned)(g * 1024);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "memory", "m0");
            }
        } else if (LRF == 2) {                           // each lane fetches the word that holds its row's 16 columns
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned* src = b_ptr[i] + (na >> 5);
                const unsigned dst = sl + (unsigned)(i * 256);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(dst), "v"(src) : "memory", "m0");
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* src = d_ptr[g] + off;
                const unsigned dst = sl + (unsigned)(g * 1024);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "memory", "m0");
            }
        }
        if (XVA) {
            const float* sx = x_src + 32 * t;
            const unsigned dx = sl + 4096u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(dx), "v"(sx) : "memory", "m0");
            const float* slb = lb_src + (long)(lb_on != 0.f ? na / va.Np : 0) * Kf;
            const unsigned dl = sl + 4096u + 256u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(dl), "v"(slb) : "memory", "m0");
        } else {
            const float* sx = x_src + 16 * t;
            const unsigned dx = sl + 4096u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dx), "v"(sx) : "memory", "m0");
        }
        if (VIRT) {
            const float* sg = g_src + 16 * t;
            const unsigned dg = sl + 4096u + 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(dg), "v"(sg) : "memory", "m0");
        }
    };
    // ---- reads from a landed slot
    int a_at[2][2];                                      // byte offsets of this lane's two pieces of fragment i
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            a_at[i][h] = (2 * i + ((lane >> 4) & 1)) * 1024 + (lane & 15) * 64 + (((2 * khalf + h) ^ ((lane >> 2) & 3)) * 16);
    auto read_a = [&](int slot, float4 (&r)[2][2], float4 (&gq)[2]) {
        const unsigned char* sl = ring + slot * WG_SLOT_BYTES;
        if (ABF) {                                       // the cell of fragment i: lane-linear, as the DMA left it
#pragma unroll
            for (int i = 0; i < 2; ++i) r[i][0] = *reinterpret_cast<const float4*>(sl + i * 1024 + 16 * lane);
            return;
        }
        if (LRF == 2) {                                  // the word this lane fetched for fragment i (bit-cast into r[i][0].x)
#pragma unroll
            for (int i = 0; i < 2; ++i) r[i][0].x = *reinterpret_cast<const float*>(sl + i * 256 + 4 * lane);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) r[i][h] = *reinterpret_cast<const float4*>(sl + a_at[i][h]);
        if (VIRT && !LRF) {
            gq[0] = *reinterpret_cast<const float4*>(sl + 4096 + 1024 + 32 * khalf);
            gq[1] = *reinterpret_cast<const float4*>(sl + 4096 + 1024 + 32 * khalf + 16);
        }
    };
    const bool xva_lrelu = XVA && va.act == ACT_LRELU;
    auto read_x = [&](int slot) -> float4 {
        const unsigned char* sl = ring + slot * WG_SLOT_BYTES + 4096;
        if (XVA) {
            const float4 c0 = *reinterpret_cast<const float4*>(sl + 32 * q4);
            const float4 c1 = *reinterpret_cast<const float4*>(sl + 32 * q4 + 16);
            const float lbv = *reinterpret_cast<const float*>(sl + 256 + 4 * lane) * lb_on;
            if (xva_lrelu) {                             // straight-line LeakyReLU (see dense_x6_kernel: no per-value branches in the loop)
                const float p0 = dec_l0_pre(va_w0, va_w1, va_bc, lbv, c0.x, c0.y), p1 = dec_l0_pre(va_w0, va_w1, va_bc, lbv, c0.z, c0.w);
                const float p2 = dec_l0_pre(va_w0, va_w1, va_bc, lbv, c1.x, c1.y), p3 = dec_l0_pre(va_w0, va_w1, va_bc, lbv, c1.z, c1.w);
                // max(p, slope p) IS LeakyReLU for 0 < slope < 1 (signed zeros and NaN included): two instructions, not three
                if (va.slope > 0.f && va.slope < 1.f)
                    return make_float4(fmaxf(p0, p0 * va.slope), fmaxf(p1, p1 * va.slope), fmaxf(p2, p2 * va.slope),
                                       fmaxf(p3, p3 * va.slope));
                return make_float4(p0 > 0.f ? p0 : p0 * va.slope, p1 > 0.f ? p1 : p1 * va.slope, p2 > 0.f ? p2 : p2 * va.slope,
                                   p3 > 0.f ? p3 : p3 * va.slope);
            }
            return make_float4(act_apply(dec_l0_pre(va_w0, va_w1, va_bc, lbv, c0.x, c0.y), va.act, va.slope),
                               act_apply(dec_l0_pre(va_w0, va_w1, va_bc, lbv, c0.z, c0.w), va.act, va.slope),
                               act_apply(dec_l0_pre(va_w0, va_w1, va_bc, lbv, c1.x, c1.y), va.act, va.slope),
                               act_apply(dec_l0_pre(va_w0, va_w1, va_bc, lbv, c1.z, c1.w), va.act, va.slope));
        }
        return *reinterpret_cast<const float4*>(sl + 16 * lane);
    };
    float a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 64 * wave + 32 * i + (lane & 31);
        a_ok[i] = m < M ? ((VIRT && !LRF) ? vg.wo[m] : 1.f) : 0.f;
    }
    auto virt_a = [&](float4 (&r)[2][2], const float4 (&gq)[2]) {
        if (VIRT && !LRF) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    r[i][h] = make_float4(virt_value(vg, r[i][h].x, 1.f, gq[h].x), virt_value(vg, r[i][h].y, 1.f, gq[h].y),
                                          virt_value(vg, r[i][h].z, 1.f, gq[h].z), virt_value(vg, r[i][h].w, 1.f, gq[h].w));
        }
    };
    auto split_a = [&](const float4 (&r)[2][2], Cell16 (&a)[2][3], int t) {
        if (ABF) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i][0].u = make_uint4(__float_as_uint(r[i][0].x), __float_as_uint(r[i][0].y), __float_as_uint(r[i][0].z),
                                       __float_as_uint(r[i][0].w));
            return;
        }
        if (LRF == 2) {                                  // eight sign bits of the word -> one cell of 0 / 1.0
            // y = b | b << 15 puts bit 2q at 2q and bit 2q + 1 at 16 + 2q: one mask and one 24-bit multiply per word (the
            // multiplier (lo >> 2q) is exact for both encodings of 1.0: 0x3c00 = 0xf << 10, 0x3f80 = 0x7f << 7)
            const int sh = ((nbeg + 16 * t) & 31) + 8 * khalf;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned b = (__float_as_uint(r[i][0].x) >> sh) & 0xffu;
                const unsigned y = b | (b << 15);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[i][0].w[q] = (y & (0x00010001u << (2 * q))) * (OneBits<NP>::lo >> (2 * q));
            }
            return;
        }
        if (LRF) {                                       // cells of [H > 0]: 1.0 = 0x3f80 (bf16) / 0x3c00 (fp16), a single part
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float v[8] = {r[i][0].x, r[i][0].y, r[i][0].z, r[i][0].w, r[i][1].x, r[i][1].y, r[i][1].z, r[i][1].w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    a[i][0].w[q] = (v[2 * q] > 0.f ? OneBits<NP>::lo : 0u) | (v[2 * q + 1] > 0.f ? OneBits<NP>::hi : 0u);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float f = NP == 2 ? a_ok[i] * sa[i] : a_ok[i];
            const float v[8] = {r[i][0].x * f, r[i][0].y * f, r[i][0].z * f, r[i][0].w * f,
                                r[i][1].x * f, r[i][1].y * f, r[i][1].z * f, r[i][1].w * f};
            if (NP == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a[i][0].w[q] = bf16_pair(v[2 * q], v[2 * q + 1]);
            } else if (NP == 2) {
                split2hx8(v, a[i][0], a[i][1]);
            } else {
                split3x8(v, a[i][0], a[i][1], a[i][2]);
            }
        }
    };
    float ssum = 0.f;
    // the row's validity and (h3) its power-of-two scale are ONE per-thread factor: the values are born scaled, s[k] is
    // accumulated scaled and unscaled once at the end -- exact, bitwise what scaling at the split gave
    const float bsx = NP == 2 ? b_ok * sx : b_ok;
    auto store_b = [&](int stage, const float4& x, int slot, bool real_step) {
        unsigned hw[2], mw[2], lw[2];
        float4 gm = make_float4(bsx, bsx, bsx, bsx);
        if (LRF) {                                       // gy of this thread's four columns joins the X values
            const float4 g4 = *reinterpret_cast<const float4*>(ring + slot * WG_SLOT_BYTES + 4096 + 1024 + 16 * q4);
            gm = make_float4(g4.x * bsx, g4.y * bsx, g4.z * bsx, g4.w * bsx);
        }
        const float v[4] = {x.x * gm.x, x.y * gm.y, x.z * gm.z, x.w * gm.w};
        // s[k] of the two-valued form: this thread's share of row kr (the clamped step past the end must not count)
        if (LRF && real_step) ssum += (v[0] + v[1]) + (v[2] + v[3]);
        uint2* dst = reinterpret_cast<uint2*>(Bs + stage * 768 + (q4 >> 1) * 128 + kr) + (q4 & 1);
        if (NP == 1) {
            dst[0] = make_uint2(bf16_pair(v[0], v[1]), bf16_pair(v[2], v[3]));
            return;
        }
        if (NP == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) split2h_pair(v[2 * q], v[2 * q + 1], hw[q], lw[q]);
            dst[0] = make_uint2(hw[0], hw[1]);
            dst[2 * 256] = make_uint2(lw[0], lw[1]);
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) split3_pair(v[2 * q], v[2 * q + 1], hw[q], mw[q], lw[q]);
        dst[0] = make_uint2(hw[0], hw[1]);
        dst[2 * 256] = make_uint2(mw[0], mw[1]);
        dst[2 * 512] = make_uint2(lw[0], lw[1]);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Cell16 af[2][3];
    const int tl = nk - 1;
    {   // prologue: three steps in flight, step 0 split as soon as its slot has landed
        dma_step(0, 0);
        dma_step(1, min(1, tl));
        dma_step(2, min(2, tl));
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
        float4 ar[2][2], g0[2];
        read_a(0, ar, g0);
        virt_a(ar, g0);
        split_a(ar, af, 0);
        store_b(0, read_x(0), 0, true);
    }
    __syncthreads();
    int s_next = 1, s_dma = 0;                           // slot of step t+1, slot the DMAs of step t+3 go to (= t % 3)
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");      // slot of step t+1 has landed
        dma_step(s_dma, min(t + 3, tl));                 // unconditional (clamped): uniform vmcnt bookkeeping
        Cell16 an[2][3];
        const uint4* bs = Bs + cur * 768 + khalf * 128 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Cell16 bf[3];
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[p].u = bs[p * 256 + j * 32];
            if (LRF) {                                   // exact 0 / 1 operand: ONE A part against the X parts
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p) mfma_part<NP>(acc[i][j], af[i][0], bf[p]);
            } else {
                mfma_np<NP>(acc[0][j], af[0], bf);
                mfma_np<NP>(acc[1][j], af[1], bf);
            }
            if (j == 0) {                                // A cells of step t+1 from the ring
                float4 ar[2][2], gn[2];
                read_a(s_next, ar, gn);
                virt_a(ar, gn);
                split_a(ar, an, min(t + 1, tl));
            }
            if (j == 1) store_b(cur ^ 1, read_x(s_next), s_next, t + 1 < nk);   // B cells of step t+1
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = an[i][p];
        s_next = s_next == 2 ? 0 : s_next + 1;
        s_dma = s_dma == 2 ? 0 : s_dma + 1;
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail DMAs still target this wave's ring
    float sk[4] = {0.f, 0.f, 0.f, 0.f};                  // slope * s[k] of this lane's four columns
    float ixv[4] = {1.f, 1.f, 1.f, 1.f};                 // h3: inverse scale of the X row behind each of them
    float* ssm = reinterpret_cast<float*>(Bs);           // the B stages are free after the loop's last barrier
    if (LRF) {
        ssum += __shfl_xor(ssum, 1, 64);                 // the four n-quads (tid & 3) of feature row kr
        ssum += __shfl_xor(ssum, 2, 64);
        if (q4 == 0) ssm[kr] = NP == 2 ? ssum * h3_inv(sx) : ssum;
    }
    if (NP == 2) {                                       // [128 ..): per tile column (= X row), [256 ..): per tile row of dY
        if (q4 == 0) ssm[128 + kr] = h3_inv(sx);
        if (!LRF) ssm[256 + tid] = h3_inv(a_scale_of(min(m0 + tid, M - 1)));
    }
    if (LRF || NP == 2) __syncthreads();
    if (LRF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sk[j] = vg.slope * ssm[j * 32 + (lane & 31)];
    }
    if (NP == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ixv[j] = ssm[128 + j * 32 + (lane & 31)];
    }
    const float oms = 1.f - vg.slope;
    if (TVAE_ABL & 8) {                                  // ablation: no epilogue (keep the accumulators alive)
        float t_ = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t_ += acc[i][j][r];
        if (t_ == 123.456f) ws[0] = t_;
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = m0 + rl;
            if (m >= M) continue;
            const float wm = (LRF && !vg.raw) ? vg.wo[m] : 1.f;       // the row factor of the factored implicit gradient
            const float ia = (NP == 2 && !LRF) ? ssm[256 + rl] : 1.f;
            float* wrow = ws + (((long)split * nbatch + batch) * M + m) * Kf + k0 + (lane & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j * 32 + (lane & 31) < Kf) {
                    const float a = NP == 2 ? (acc[i][j][r] * ia) * ixv[j] : acc[i][j][r];     // one factor after the other
                    wrow[j * 32] = LRF ? wm * __fmaf_rn(oms, a, sk[j]) : a;
                }
        }
}

// ------------------------------------------------------------------------------------------
// Round 4: the spectral forward GEMM with the STREAMED operand resident and its stores inside the k-loop.
// T[fx][m'][n] = sum_k W[fx][m'][k] A^T[fx][k][n] has a short reduction (k = 2 L Cin: 192 at the 64 x 64 frame, twelve 16-wide
// steps) and a huge output (3.4 GB).  dense_x6_plain4_kernel computes a tile and THEN stores it: a wave cannot overlap its own
// stores with its own products, and with two workgroups per CU nobody else does it reliably either (1.39-1.46 ms; its stores
// alone 0.6-0.8 ms, its matrix work alone 0.93).  Here:
//   * a workgroup owns ONE panel (problem fx, 128 columns): phase 1 builds the B cells of all k-steps once into LDS (98 KB in
//     the h3 arithmetic; plain4 rebuilt -- loaded, scaled, split -- the panel once per 256-row tile, with a barrier per step);
//   * phase 2 has NO barrier: each wave walks its own 64-row blocks of the problem (rows 64 wave + 256 i), streaming the weight
//     cells from L2 into EIGHT register buffers (seven k-steps of distance) against the static cells;
//   * FOUR waves per workgroup, one per SIMD, so that a wave owns 512 registers: two accumulator sets of 128 (one lives in
//     AGPRs) -- while block i + 1 is being multiplied, the 128 store instructions of block i are issued a few per k-step (the
//     loop is unrolled over NK), so the write stream is continuous.  Only the last block of a wave stores alone.
// History of the forms, each parity-green and measured (profiles/README.md, round 4): eight waves with 32-row blocks and
// 64-register accumulator sets: 1.37 ms -- every 1 KB B-fragment read from LDS feeds ONE group of MFMAs there, two here, and the
// weight-cell loads (three to four buffers were all the registers allowed) waited behind the wave's own stores on the one
// in-order vmcnt counter (57 % of wave cycles parked; without those loads 1.05 ms); the same with the stores after the k-loop
// 1.51; 64-column panels at four waves per SIMD 1.44; weight cells resident in registers with double-buffered panels 1.46-1.50.
// This form: 1.24-1.30 ms; buffers 3 / 6 / 8: 1 437 / 1 272 / 1 235 us on one box; all twelve steps as one stream across the
// blocks: no better.
// Host: lean epilogue shape of the spectral contraction (plain column-tiled fp32 output), N % 128 == 0, rows per problem a
// multiple of 512 (M; Mb >= M is their stride in the cell array), NK = K8pad / 2 = 12 steps, NP = 2 or 3.  Grid: groups (problem,
// quarter of the column tiles) dealt round-robin to the XCDs: a problem's weight cells (0.8 MB) stay in one L2.
// ------------------------------------------------------------------------------------------
template <int NP, int NK>
static __global__ __launch_bounds__(256, 1)
void dense_x6_xres_kernel(const uint4* __restrict__ A3, const float* __restrict__ X, long ldx, float* __restrict__ C, long ldc,
                           long ctile, int M, int Mb, int Mpad, int K, int nprob, int tilesN, int nch, long x_stride,
                           long c_stride, H3Scale hs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xres_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int K8pad = 2 * NK;
    uint4* Bs = reinterpret_cast<uint4*>(xres_lds);                        // [step][part][octet half][128 n]
    float* h3a_ = reinterpret_cast<float*>(xres_lds + (size_t)NK * NP * 256 * 16);     // [Mb] inverse row scales
    float* h3x_ = h3a_ + Mb;                                               // [128] inverse column scales
    const int cs = (tilesN + nch - 1) / nch;
    const int g = ((blockIdx.x >> 3) / cs) * 8 + (blockIdx.x & 7), r = (blockIdx.x >> 3) % cs;
    const int batch = g / nch, tile_n = (g - batch * nch) * cs + r;
    if (batch >= nprob || tile_n >= tilesN) return;
    X += batch * x_stride;
    C += batch * c_stride + (long)tile_n * ctile;
    const int n0 = tile_n * 128;
    const int khalf = lane >> 5;
    {   // phase 1: the panel's cells, once (256 threads: k-quads kq and kq + 2 of a step)
        const int kq = tid >> 7, nb = tid & 127;
        float sx = 1.f;
        if (NP == 2) {
            sx = hs.x_group > 0 ? h3_scale(hs.amax_x[(long)batch * hs.x_bstride + min((n0 + nb) / hs.x_group, hs.x_bstride - 1)])
                                : h3_scale(hs.amax_x[0]);
            if (tid < 128) h3x_[tid] = h3_inv(sx);
            for (int m = tid; m < Mb; m += 256) h3a_[m] = h3_inv(h3_scale(hs.amax_a[hs.a_rows ? batch * Mb + m : 0]));
        }
        const float* x_col = X + n0 + nb;
#pragma unroll
        for (int t0 = 0; t0 < NK; t0 += 3) {              // three steps (24 loads) in flight
            float x[3][2][4];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * (t0 + u < NK ? t0 + u : NK - 1) + 4 * (kq + 2 * h) + j;
                        x[u][h][j] = x_col[(long)min(k, K - 1) * ldx];
                    }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (t0 + u >= NK) break;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint2* dst = reinterpret_cast<uint2*>(Bs + (size_t)(t0 + u) * NP * 256 + h * 128 + nb) + kq;
                    if (NP == 2) {
                        unsigned hw[2], lw[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) split2h_pair(x[u][h][2 * q] * sx, x[u][h][2 * q + 1] * sx, hw[q], lw[q]);
                        dst[0] = make_uint2(hw[0], hw[1]);
                        dst[2 * 256] = make_uint2(lw[0], lw[1]);
                    } else {
                        unsigned hw[2], mw[2], lw[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) split3_pair(x[u][h][2 * q], x[u][h][2 * q + 1], hw[q], mw[q], lw[q]);
                        dst[0] = make_uint2(hw[0], hw[1]);
                        dst[2 * 256] = make_uint2(mw[0], mw[1]);
                        dst[2 * 512] = make_uint2(lw[0], lw[1]);
                    }
                }
            }
        }
    }
    __syncthreads();
    const long part_cells = (long)K8pad * Mpad;
    const uint4* bs0 = Bs + khalf * 128 + (lane & 31);
    const uint4* a_base = A3 + (long)khalf * Mpad + (long)batch * Mb + (lane & 31);
    float ixv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ixv[j] = NP == 2 ? h3x_[j * 32 + (lane & 31)] : 1.f;
    const unsigned loff = (unsigned)((4 * khalf * ldc + (lane & 31)) * 4);
    // store row s (0 .. 31) of a finished 64-row block: fragment i = s >> 4, row mp + 32 i + 8 q + p (+ 4 upper half wave)
    auto store_row = [&](const f32x16 (&acc)[2][4], int mp, int s_) {
        const int i = s_ >> 4, q = (s_ >> 2) & 3, p = s_ & 3;
        const int row = mp + 32 * i + 8 * q + p;
        float* rowp = reinterpret_cast<float*>(reinterpret_cast<char*>(C + (long)row * ldc) + loff);
        const float ia = NP == 2 ? h3a_[row + 4 * khalf] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = NP == 2 ? (acc[i][j][4 * q + p] * ixv[j]) * ia : acc[i][j][4 * q + p];
            __builtin_nontemporal_store(v, rowp + j * 32);
        }
    };
    auto block = [&](f32x16 (&accC)[2][4], int mc, const f32x16 (&accP)[2][4], int mp, bool have_prev) {
        const uint4* a_run[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) a_run[p] = a_base + mc + p * part_cells;
#ifndef TVAE_XRES2_NB
#define TVAE_XRES2_NB 8
#endif
        constexpr int NB = TVAE_XRES2_NB;                // operand buffers: the weight cells of steps t .. t + NB - 1 (measured:
        Cell16 af[NB][2][3];                             //  3: 1 437 us, 6: 1 272, 8: 1 235; all twelve as one stream across blocks: no better)
        auto load_next = [&](Cell16 (&a)[2][3]) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                a[0][p].u = a_run[p][0];
                a[1][p].u = a_run[p][32];
                a_run[p] += 2 * (long)Mpad;
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) accC[i][j][r_] = 0.f;
#pragma unroll
        for (int u = 0; u + 1 < NB; ++u)
            if (u < NK) load_next(af[u]);
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            if (t + NB - 1 < NK) load_next(af[(t + NB - 1) % NB]);
            const uint4* bs = bs0 + (size_t)t * NP * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Cell16 bf[3];
#pragma unroll
                for (int p = 0; p < NP; ++p) bf[p].u = bs[p * 256 + j * 32];
                mfma_np<NP>(accC[0][j], af[t % NB][0], bf);
                mfma_np<NP>(accC[1][j], af[t % NB][1], bf);
            }
            if (have_prev) {
#pragma unroll
                for (int s_ = (32 * t) / NK; s_ < (32 * (t + 1)) / NK; ++s_) store_row(accP, mp, s_);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x16 accA[2][4], accB[2][4];
    int mprev = 0;
    bool have = false;
    for (int m0 = 64 * wave; m0 < M; m0 += 512) {        // two blocks per trip (rows m0 and m0 + 256): static accumulator sets
        block(accA, m0, accB, mprev, have);
        block(accB, m0 + 256, accA, m0, true);
        mprev = m0 + 256;
        have = true;
    }
#pragma unroll
    for (int s_ = 0; s_ < 32; ++s_) store_row(accB, mprev, s_);
}

// ------------------------------------------------------------------------------------------
// Round 4: the spectral weight gradient on a tile that FITS it.  The frequency-domain convolution's weight gradient is a
// batched GEMM with 2 L Cin columns -- 192 at the 96-wide frame of the 64 x 64 configuration -- so the 512 x 128 tile of
// dense_wgrad_x6_dma_kernel spent a quarter of its matrix instructions on the zero half of a second column tile, split
// every value of the large operand S' twice (once per column tile) and read it twice.  Here the tile is 256 rows x 32 NJ
// columns (NJ = 6: all 192 columns at once): eight waves, each ONE 32-row fragment of dY against all column groups
// (acc[NJ]: 96 registers), the same per-wave LDS-DMA ring with hand-counted waits (A: two 1 KB pieces per step, 16 rows x
// one 64-byte line each, XOR swizzled; X: every lane its own 16 bytes of row tid / 4, waves 0-3 a second piece for the rows
// beyond 128 -- waves 4-7 issue a clamped duplicate so that the count stays uniform), the same cooperative B stage
// ([part][octet half][32 NJ rows] cells, two stages, one barrier per step).  Plain operands only (no implicit forms);
// NP = 3 / 2 / 1 as everywhere.  Host: M % 256 == 0; tm.tilesN column tiles of 32 NJ columns cover Kf.
// Measured (64 x 64 step, h3): 1.58 -> 1.28 ms.  Ablation builds (-DTVAE_WW_ABL=bits: 1 no matrix instructions, 2 no
// B-fragment reads, 4 no A split, 8 no B build; profiles/tools/build_variant.sh): the bare ring + barriers stream the 4.3 GB
// in 0.95 ms (4.5 TB/s); a ring of four slots, a contiguous S' layout, nine reduction slices (whole rounds of the 256 CUs)
// and waiting for step t+1 after the first matrix instructions of step t all changed nothing (profiles/README.md, round 4).
// ------------------------------------------------------------------------------------------
constexpr int WW_SLOT_BYTES = 2048 + 2048;            // A (32 rows x 16 n) | X (two pieces)
#ifndef TVAE_WW_SLOTS
#define TVAE_WW_SLOTS 3
#endif
constexpr int WW_SLOTS = TVAE_WW_SLOTS;               // ring depth: steps in flight per wave
constexpr int WW_RING_BYTES = 8 * WW_SLOTS * WW_SLOT_BYTES;
constexpr int WW_ROWS = 256;

template <int NP, int NJ>
static __global__ __launch_bounds__(DX6_THREADS, 2)
void dense_wgrad_x6_wide_kernel(const float* __restrict__ dY, long ldd, const float* __restrict__ X, long ldx, float* ws, int M,
                                int Kf, int N, int nchunk, TileMap tm, DenseBatch bt, long dy_stride, ATile atile, H3Scale hs) {
    constexpr int KR = 32 * NJ;                          // feature rows (= tile columns)
    constexpr int STG = NP * 2 * KR;                     // cells per B stage
    constexpr int NDMA = 2 + 2;                          // DMA instructions per wave and step (A x 2, X x 2)
    static_assert(NJ >= 5 && NJ <= 8, "one full pass of the 512 build threads plus a partial second one");
    __shared__ __attribute__((aligned(16))) uint4 Bs[2 * STG];
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_ring[];   // [wave][slot < WW_SLOTS][WW_SLOT_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (reduction slice, problem, row tile).  A group is the row tiles of one (slice, problem): they share the
    // X panel, so they run back to back on ONE XCD; groups are dealt round-robin to the 8 XCDs.  Slices are therefore NOT
    // pinned to XCDs as TileMap does it, and their number is free (9 slices of 49 problems x 4 tiles fill 7 whole rounds of
    // the 256 CUs where 8 run 6.125: measured the same, the host keeps 8).
    // (Wider problems -- 1 152 columns at the galaxy shape -- are several such column tiles; the tiles of a group are row
    //  tile fastest, so the row tiles that share an X panel stay adjacent.)
    const int tiles_b = bt.tiles_per_batch, nbatch = tm.tilesM / tiles_b, gsz = tiles_b * tm.tilesN;
    const int g = ((blockIdx.x >> 3) / gsz) * 8 + (blockIdx.x & 7), rr = (blockIdx.x >> 3) % gsz;
    if (g >= tm.splits * nbatch) return;
    const int tile_m = rr % tiles_b, k0 = (rr / tiles_b) * (32 * NJ), split = g / nbatch, batch = g - split * nbatch;
    dY += batch * dy_stride;
    X += batch * bt.x_stride;
    const int m0 = tile_m * WW_ROWS;
    const int nbeg = split * nchunk;
    const int nend = min(N, nbeg + nchunk);
    const int nk = (nend - nbeg) >> 4;
    const int khalf = lane >> 5;
    float* slab = ws + (((long)split * nbatch + batch) * M) * Kf;
    if (nk <= 0) {                                      // empty reduction slice: zero partial
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (k0 + j * 32 + (lane & 31) < Kf) slab[(long)m * Kf + k0 + j * 32 + (lane & 31)] = 0.f;
        }
        return;
    }
    // ---- h3 scales: one per row of either operand (H3Scale)
    auto a_scale_of = [&](int m) -> float {
        return h3_scale(hs.amax_a[hs.a_rows ? (long)batch * hs.a_bstride + (hs.a_mod > 0 ? m % hs.a_mod : m) : 0]);
    };
    const float sa = NP == 2 ? a_scale_of(min(m0 + 32 * wave + (lane & 31), M - 1)) : 1.f;
    // B build roles: (row kr, n-quad q4) and, for the first 4 (KR - 128) threads, (row 128 + kr, q4)
    const int kr = tid >> 2, q4 = tid & 3;
    const bool two = tid < 4 * (KR - 128);               // wave uniform: KR - 128 is a multiple of 16 rows = one wave
    const int kr2 = 128 + kr;
    const int kx0 = min(k0 + kr, Kf - 1), kx1 = min(k0 + (two ? kr2 : kr), Kf - 1);
    auto x_scale_of = [&](int kx) -> float {
        if (NP != 2) return 1.f;
        return h3_scale(hs.x_group > 0 ? hs.amax_x[(long)batch * hs.x_bstride + kx / hs.x_group] : hs.amax_x[0]);
    };
    const float b0s = (k0 + kr < Kf ? 1.f : 0.f) * x_scale_of(kx0);
    const float b1s = ((two && k0 + kr2 < Kf) ? 1.f : 0.f) * x_scale_of(kx1);
    // ---- DMA sources
    const float* d_ptr[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int row = m0 + 32 * wave + 16 * g + (lane >> 2);
        d_ptr[g] = dY + (long)min(row, M - 1) * ldd + 4 * ((lane & 3) ^ ((lane >> 4) & 3));
    }
    const float* x_src0 = X + (long)kx0 * ldx + nbeg + 4 * q4;
    const float* x_src1 = X + (long)kx1 * ldx + nbeg + 4 * q4;      // (waves beyond the partial pass: a duplicate of piece 0)
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)wg_ring +
                              (unsigned)(wave * WW_SLOTS * WW_SLOT_BYTES);
    const unsigned char* ring = wg_ring + wave * WW_SLOTS * WW_SLOT_BYTES;
    auto dma_step = [&](int slot, int t) {
        const int na = nbeg + 16 * t;                    // wave-uniform; a 16-wide step never straddles a column tile
        const long off = (long)(na >> atile.sh) * atile.ts + (na & atile.mask);
        const unsigned sl = ring_lds + (unsigned)(slot * WW_SLOT_BYTES);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
                const float* src = d_ptr[g] + off;
            const unsigned dst = sl + (unsigned)(g * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "memory", "m0");
        }
        {
            const float* s0 = x_src0 + 16 * t;
            const unsigned d0 = sl + 2048u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(d0), "v"(s0) : "memory", "m0");
            const float* s1 = x_src1 + 16 * t;
            const unsigned d1 = sl + 3072u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(d1), "v"(s1) : "memory", "m0");
        }
    };
    // ---- reads from a landed slot: this lane's two 16-byte pieces of the A fragment (rows lane & 31, n-half khalf)
    int a_at[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        a_at[h] = ((lane >> 4) & 1) * 1024 + (lane & 15) * 64 + (((2 * khalf + h) ^ ((lane >> 2) & 3)) * 16);
    const float a_ok = (m0 + 32 * wave + (lane & 31)) < M ? 1.f : 0.f;
    auto build_a = [&](int slot, Cell16 (&a)[3]) {
        const unsigned char* sl = ring + slot * WW_SLOT_BYTES;
        const float4 r0 = *reinterpret_cast<const float4*>(sl + a_at[0]);
        const float4 r1 = *reinterpret_cast<const float4*>(sl + a_at[1]);
        const float f = NP == 2 ? a_ok * sa : a_ok;
        const float v[8] = {r0.x * f, r0.y * f, r0.z * f, r0.w * f, r1.x * f, r1.y * f, r1.z * f, r1.w * f};
#if defined(TVAE_WW_ABL) && (TVAE_WW_ABL & 4)
        a[0].u = make_uint4(__float_as_uint(r0.x), __float_as_uint(r0.y), __float_as_uint(r0.z), __float_as_uint(r0.w));
        a[1].u = make_uint4(__float_as_uint(r1.x), __float_as_uint(r1.y), __float_as_uint(r1.z), __float_as_uint(r1.w));
        a[2].u = a[0].u;
        return;                                          // ablation: no split arithmetic of the A operand
#endif
        if (NP == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[0].w[q] = bf16_pair(v[2 * q], v[2 * q + 1]);
        } else if (NP == 2) {
            split2hx8(v, a[0], a[1]);
        } else {
            split3x8(v, a[0], a[1], a[2]);
        }
    };
    auto store_b = [&](int stage, int slot, int which) {
        const float4 x = *reinterpret_cast<const float4*>(ring + slot * WW_SLOT_BYTES + 2048 + 1024 * which + 16 * lane);
        const float f = which ? b1s : b0s;
        const float v[4] = {x.x * f, x.y * f, x.z * f, x.w * f};
        const int row = which ? kr2 : kr;
        uint2* dst = reinterpret_cast<uint2*>(Bs + stage * STG + (q4 >> 1) * KR + row) + (q4 & 1);
#if defined(TVAE_WW_ABL) && (TVAE_WW_ABL & 8)
        if (x.x == 1.2345f) dst[0] = make_uint2(0, 0);   // ablation: no B-stage build
        return;
#endif
        unsigned hw[2], mw[2], lw[2];
        if (NP == 1) {
            dst[0] = make_uint2(bf16_pair(v[0], v[1]), bf16_pair(v[2], v[3]));
            return;
        }
        if (NP == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) split2h_pair(v[2 * q], v[2 * q + 1], hw[q], lw[q]);
            dst[0] = make_uint2(hw[0], hw[1]);
            dst[2 * 2 * KR] = make_uint2(lw[0], lw[1]);
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) split3_pair(v[2 * q], v[2 * q + 1], hw[q], mw[q], lw[q]);
        dst[0] = make_uint2(hw[0], hw[1]);
        dst[2 * 2 * KR] = make_uint2(mw[0], mw[1]);
        dst[2 * 4 * KR] = make_uint2(lw[0], lw[1]);
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    Cell16 af[3];
    const int tl = nk - 1;
    {   // prologue: WW_SLOTS steps in flight, step 0 split as soon as its slot has landed
#pragma unroll
        for (int i = 0; i < WW_SLOTS; ++i) dma_step(i, min(i, tl));
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WW_SLOTS - 1) * NDMA) : "memory");
        build_a(0, af);
        store_b(0, 0, 0);
        if (two) store_b(0, 0, 1);
    }
    __syncthreads();
    int s_next = 1, s_dma = 0;                           // slot of step t+1, slot the DMAs of step t+WW_SLOTS go to (= t % WW_SLOTS)
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WW_SLOTS - 2) * NDMA) : "memory");      // slot of step t+1 has landed
        dma_step(s_dma, min(t + WW_SLOTS, tl));                 // unconditional (clamped): uniform vmcnt bookkeeping
        Cell16 an[3];
        const uint4* bs = Bs + cur * STG + khalf * KR + (lane & 31);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            Cell16 bf[3];
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[p].u = bs[p * 2 * KR + j * 32];
#if defined(TVAE_WW_ABL) && (TVAE_WW_ABL & 2)
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[p].u = af[p].u;        // ablation: no B-fragment LDS reads
#endif
#if !(defined(TVAE_WW_ABL) && (TVAE_WW_ABL & 1))
            mfma_np<NP>(acc[j], af, bf);
#else
            acc[j][0] += __uint_as_float(bf[0].w[0] ^ af[0].w[1]);  // ablation: no matrix instructions
#endif
            if (j == 0) build_a(s_next, an);             // A cells of step t+1 from the ring
            if (j == 1) store_b(cur ^ 1, s_next, 0);     // B cells of step t+1
            if (j == 2 && two) store_b(cur ^ 1, s_next, 1);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = an[p];
        s_next = s_next == WW_SLOTS - 1 ? 0 : s_next + 1;
        s_dma = s_dma == WW_SLOTS - 1 ? 0 : s_dma + 1;
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail DMAs still target this wave's ring
    float* ssm = reinterpret_cast<float*>(Bs);           // the B stages are free after the loop's last barrier
    float ixv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ixv[j] = 1.f;
    if (NP == 2) {                                       // [0 .. KR): inverse scale per tile column (= X row), [KR ..): per tile row
        if (q4 == 0) {
            ssm[kr] = h3_inv(x_scale_of(kx0));
            if (two) ssm[kr2] = h3_inv(x_scale_of(kx1));
        }
        if (tid < WW_ROWS) ssm[KR + tid] = h3_inv(a_scale_of(min(m0 + tid, M - 1)));
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) ixv[j] = ssm[j * 32 + (lane & 31)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = m0 + rl;
        if (m >= M) continue;
        const float ia = NP == 2 ? ssm[KR + rl] : 1.f;
        float* wrow = slab + (long)m * Kf + k0 + (lane & 31);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (k0 + j * 32 + (lane & 31) < Kf) wrow[j * 32] = NP == 2 ? (acc[j][r] * ia) * ixv[j] : acc[j][r];
    }
}

}  // namespace tvae
