// Encoder tail  conv2 (1x1x1, 128 -> 128) + {conv_a, conv_r, conv_z} (128 -> 3 + 2 z_dim <= 7 rows)  of the reference's
// InferenceNetwork_AttentionTranslation_AttentionRotation (src/models.py:347-358, 390-392) on the bf16 matrix pipe with
// exactly split operands (the "x6" arithmetic of dense_x6_kernels.hpp), fused so that every [128][B*R*Ho*Ho] tensor is
// streamed ONCE per kernel:
//
//   forward   H = act(W2 A1 + b2),  heads = Wh H + bh                  reads A1, writes H (+ 7 rows)
//   data grad dA1 = act'(A1) . W2^T (act'(H) . Wh^T dheads)            reads H, A1, dheads; writes dA1  (dH never stored)
//   weight    dW2 = dH A1^T, dWh = dheads H^T, db2 = rowsum dH         reads H, A1, dheads
//
// Shape of the problem: M = K = 128 and N = 2.2 M columns at the headline batch, i.e. HBM bound once the products run on
// the bf16 pipe (438 G bf16 FLOP against 2.3 - 3.4 GB per kernel).  The 512-row tile of dense_x6_kernel would waste 3/4
// of its rows here, so these kernels turn the tiling round: the WHOLE weight (3 parts x 16 k-octets x 128 rows of
// 16-byte cells = 96 KB) is stationary in LDS, one persistent workgroup per CU, and every wave streams its own chunks
// of 32 columns through all 128 output rows (wave tile 128 x 32: 4 MFMA tiles, 24 MFMAs per 16-k step, 8 steps):
//   * the streamed operand goes global -> registers directly in B-fragment order (lane = column, 8 k-rows per lane:
//     8 dword loads, each two full 128-byte lines per wave instruction), is split in registers, and never touches LDS;
//     a step's registers are refilled with the SAME step of the wave's next chunk as soon as they have been split, i.e.
//     loads run one whole chunk (8 steps, 16 KB per wave, 128 KB per CU) ahead: with two waves per SIMD the kernel is
//     bound by HBM latency x bytes in flight, and a two-step lead (the 128 x 64 tile's register budget) measured
//     3.6 TB/s where this pattern alone streams 5.9 TB/s (profiles/experiments/stream_patterns.hip);
//   * no barrier after the prologue: the waves of a workgroup only share the read-only weight cells;
//   * the epilogue works in the accumulator layout (lane = column, 32 consecutive columns = 128 contiguous bytes per
//     row and store instruction); the skinny head projection is 7 FMAs per element on the vector ALU against
//     broadcast LDS reads of Wh^T, summed over the two lane halves.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_x6_kernels.hpp"

namespace tvae {

constexpr int ET_C = 128;                 // channels of both layers (reference default kernels_num; other widths: unfused path)
#ifndef ET_WAVES
#define ET_WAVES 8
#endif
#ifndef ET_DEPTH
#define ET_DEPTH 8
#endif
constexpr int ET_THREADS = 64 * ET_WAVES; // waves per workgroup (one workgroup per CU)
constexpr int ET_D = ET_DEPTH;            // k-steps the operand loads run ahead (2, 4 or 8 = one whole chunk)
constexpr int ET_CHUNK = 32;              // columns per wave chunk
constexpr int ET_MAXH = 7;                // head rows (3 + 2 z_dim)
#ifndef ET_ABL
#define ET_ABL 0                          // ablation switches of profiles/experiments/enc_tail_ablate.hip (0 in the library)
#endif

template <int ACT>
__device__ __forceinline__ float et_act(float v, float slope) {
    if (ACT == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (ACT == ACT_TANH) return tanhf(v);
    return v;
}

// row of accumulator register r of row tile i in lane half kh
__device__ __forceinline__ int et_row(int i, int r, int kh) { return 32 * i + 8 * (r >> 2) + (r & 3) + 4 * kh; }

// split of the 8 k-values a lane holds for its column into the B-fragment cells of one 16-k step
template <int NP>
__device__ __forceinline__ void et_split(const float (&x)[8], Cell16 (&bf)[3]) {
    if (NP == 3) {
        split3x8(x, bf[0], bf[1], bf[2]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) bf[0].w[q] = bf16_pair(x[2 * q], x[2 * q + 1]);
    }
}

// 24 MFMAs of one step: the weight cells of row tile 0 of the NEXT step are fetched during the last row tile of this
// one, the other row tiles one ahead (two register sets; a scheduling fence per row tile keeps the scheduler from
// hoisting all twelve LDS reads, which would spill the accumulators)
template <int NP>
__device__ __forceinline__ void et_load_a(const uint4* __restrict__ Ws, int t, int i, int kh, int nl, Cell16 (&a)[3]) {
    if (ET_ABL & 16) return;
    const uint4* wp = Ws + (2 * t + kh) * ET_C + nl + 32 * i;
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p].u = wp[p * 16 * ET_C];
}
template <int NP>
__device__ __forceinline__ void et_step_mfma(f32x16 (&acc)[4], const uint4* __restrict__ Ws, int t, int kh, int nl,
                                             Cell16 (&a0)[3], Cell16 (&a1)[3], const Cell16 (&bf)[3]) {
    if (ET_ABL & 32) {                                   // weight reads without the MFMAs
        et_load_a<NP>(Ws, t, 1, kh, nl, a1);
        acc[0][0] += __uint_as_float(a0[0].w[0] ^ a0[NP - 1].w[3] ^ bf[0].w[1]);
        et_load_a<NP>(Ws, t, 2, kh, nl, a0);
        acc[1][0] += __uint_as_float(a1[0].w[0] ^ a1[NP - 1].w[3]);
        et_load_a<NP>(Ws, t, 3, kh, nl, a1);
        acc[2][0] += __uint_as_float(a0[0].w[0] ^ a0[NP - 1].w[3]);
        et_load_a<NP>(Ws, (t + 1) & 7, 0, kh, nl, a0);
        acc[3][0] += __uint_as_float(a1[0].w[0] ^ a1[NP - 1].w[3]);
        return;
    }
    et_load_a<NP>(Ws, t, 1, kh, nl, a1);
    mfma_np<NP>(acc[0], a0, bf);
    __builtin_amdgcn_sched_barrier(0);
    et_load_a<NP>(Ws, t, 2, kh, nl, a0);
    mfma_np<NP>(acc[1], a1, bf);
    __builtin_amdgcn_sched_barrier(0);
    et_load_a<NP>(Ws, t, 3, kh, nl, a1);
    mfma_np<NP>(acc[2], a0, bf);
    __builtin_amdgcn_sched_barrier(0);
    et_load_a<NP>(Ws, (t + 1) & 7, 0, kh, nl, a0);
    mfma_np<NP>(acc[3], a1, bf);
    __builtin_amdgcn_sched_barrier(0);
}

// FULL: the chunk lies inside [0, N) -- no per-element bounds (every chunk but possibly the last one)
template <int ACT, bool FULL>
__device__ __forceinline__ void et_fwd_epilogue(f32x16 (&acc)[4], const float* __restrict__ whs, float* __restrict__ H,
                                                long ldh, float* __restrict__ heads, long ldo, const float* __restrict__ bh,
                                                int nh, long n0, long N, int nl, int kh, float slope) {
    float hs[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) hs[o] = 0.f;
    const bool in0 = FULL || n0 + nl < N;
    // uniform row pointer + one 32-bit lane offset in BYTES: the stores take the (SGPR base, VGPR offset) form
    const unsigned loff = (unsigned)(4 * kh * ldh + nl) * 4u;
    const float* wlane = whs + 32 * kh;
    // Wh^T rows one pair of rows ahead, and a scheduling fence per pair: left alone the scheduler hoists all 128 LDS
    // reads above the first FMA and spills the accumulators to make room
    float4 wq[2][2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        wq[0][u][0] = *reinterpret_cast<const float4*>(wlane + et_row(0, u, 0) * 8);
        wq[0][u][1] = *reinterpret_cast<const float4*>(wlane + et_row(0, u, 0) * 8 + 4);
    }
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int cur = it & 1;
        if (it + 1 < 32) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rown = et_row((2 * it + 2 + u) >> 4, (2 * it + 2 + u) & 15, 0);
                wq[cur ^ 1][u][0] = *reinterpret_cast<const float4*>(wlane + rown * 8);
                wq[cur ^ 1][u][1] = *reinterpret_cast<const float4*>(wlane + rown * 8 + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = (2 * it + u) >> 4, r = (2 * it + u) & 15;
            const float4 w0 = wq[cur][u][0], w1 = wq[cur][u][1];           // w1.w = b2[row]
            char* hrow = reinterpret_cast<char*>(H + (long)et_row(i, r, 0) * ldh + n0);
            const float v = et_act<ACT>(acc[i][r] + w1.w, slope);
            if (in0 && !(ET_ABL & 1)) *reinterpret_cast<float*>(hrow + loff) = v;
            if (ET_ABL & 2) { hs[0] += v; continue; }
            hs[0] = __fmaf_rn(w0.x, v, hs[0]);
            hs[1] = __fmaf_rn(w0.y, v, hs[1]);
            hs[2] = __fmaf_rn(w0.z, v, hs[2]);
            hs[3] = __fmaf_rn(w0.w, v, hs[3]);
            hs[4] = __fmaf_rn(w1.x, v, hs[4]);
            hs[5] = __fmaf_rn(w1.y, v, hs[5]);
            hs[6] = __fmaf_rn(w1.z, v, hs[6]);
        }
        // pin the partial sums here: otherwise the compiler sinks six of the seven FMA chains below the loop and keeps
        // (spills) every operand alive until then
#pragma unroll
        for (int o = 0; o < ET_MAXH; ++o) asm volatile("" : "+v"(hs[o]));
        __builtin_amdgcn_sched_barrier(0);
    }
    // the two lane halves hold disjoint rows of the same 32 columns: half 0 stores the even head rows, half 1 the odd ones
    float* hp = heads + n0 + nl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float ve = hs[2 * q] + __shfl_xor(hs[2 * q], 32, 64);
        const float vo = (2 * q + 1 < ET_MAXH) ? hs[2 * q + 1] + __shfl_xor(hs[2 * q + 1], 32, 64) : 0.f;
        const int o = 2 * q + kh;
        if (o < nh && in0) hp[(long)o * ldo] = (kh ? vo : ve) + bh[o];
    }
}

// W3: cells of W2 as written by dense_split3_kernel (transpose = 0), [part][16 octets][Rpad rows]; only rows < 128 are read.
template <int NP>
static __global__ __launch_bounds__(ET_THREADS, ET_WAVES / 4)
void enc_tail_fwd_x6_kernel(const uint4* __restrict__ W3, int Rpad, const float* __restrict__ X, long ldx,
                            const float* __restrict__ b2, const float* __restrict__ Wh, const float* __restrict__ bh, int nh,
                            float* __restrict__ H, long ldh, float* __restrict__ heads, long ldo, long N, int act,
                            float slope) {
    extern __shared__ __attribute__((aligned(16))) uint4 Ws[];            // [NP][16][128]
    __shared__ __attribute__((aligned(16))) float whs[ET_C * 8];          // row m: Wh[0..6][m], b2[m]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kh = lane >> 5;
    for (int i = tid; i < NP * 16 * ET_C; i += ET_THREADS) {
        const int row = i & (ET_C - 1), po = i >> 7;                      // po = part * 16 + octet
        Ws[i] = W3[(long)po * Rpad + row];
    }
    for (int i = tid; i < ET_C * 8; i += ET_THREADS) {
        const int row = i >> 3, o = i & 7;
        whs[i] = o == 7 ? (b2 ? b2[row] : 0.f) : (o < nh ? Wh[o * ET_C + row] : 0.f);
    }
    __syncthreads();

    const long nchunks = (N + ET_CHUNK - 1) / ET_CHUNK;
    const long gw = (long)blockIdx.x * (ET_THREADS / 64) + wave, gstride = (long)gridDim.x * (ET_THREADS / 64);
    if (gw >= nchunks) return;
    const long my = (nchunks - 1 - gw) / gstride + 1;

    // uniform row pointer (SGPRs) + a 32-bit lane offset in BYTES (columns beyond N re-read column N - 1)
    const unsigned xlane = (unsigned)(8 * kh * ldx) * 4u;
    auto load_x = [&](long ci, int t, float (&x)[8]) {
        if (ci >= my) ci = my - 1;                                        // harmless reload of real data
        const long n0 = (gw + ci * gstride) * ET_CHUNK;
        const char* p = reinterpret_cast<const char*>(X + (long)(16 * t) * ldx + n0);
        const unsigned off = xlane + 4u * (unsigned)min((long)nl, N - 1 - n0);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float*>(p + (long)j * ldx * 4 + off);
    };

    f32x16 acc[4];
    float x[ET_D][8];                                                     // ring: x[t % ET_D] holds step t (ET_D steps ahead)
#pragma unroll
    for (int t = 0; t < ET_D; ++t) load_x(0, t, x[t]);
    Cell16 a0[3], a1[3];
    if (ET_ABL & 16) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a0[p].u = a1[p].u = make_uint4(0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    }
    et_load_a<NP>(Ws, 0, 0, kh, nl, a0);
    for (long ci = 0; ci < my; ++ci) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Cell16 bf[3];
            et_split<NP>(x[t % ET_D], bf);
            if (!(ET_ABL & 8)) load_x(ci + (t + ET_D) / 8, (t + ET_D) % 8, x[t % ET_D]);
            if (ET_ABL & 4) { acc[t & 3][0] += __uint_as_float(bf[0].w[0] ^ bf[NP - 1].w[3]); continue; }
            et_step_mfma<NP>(acc, Ws, t, kh, nl, a0, a1, bf);
        }
        const long n0 = (gw + ci * gstride) * ET_CHUNK;
#define TVAE_ET_EPI(A_)                                                                                    \
    do {                                                                                                  \
        if (n0 + ET_CHUNK <= N) et_fwd_epilogue<A_, true>(acc, whs, H, ldh, heads, ldo, bh, nh, n0, N, nl, kh, slope); \
        else et_fwd_epilogue<A_, false>(acc, whs, H, ldh, heads, ldo, bh, nh, n0, N, nl, kh, slope);      \
    } while (0)
        if (act == ACT_LRELU) TVAE_ET_EPI(ACT_LRELU);
        else if (act == ACT_TANH) TVAE_ET_EPI(ACT_TANH);
        else TVAE_ET_EPI(ACT_NONE);
#undef TVAE_ET_EPI
    }
}

}  // namespace tvae
