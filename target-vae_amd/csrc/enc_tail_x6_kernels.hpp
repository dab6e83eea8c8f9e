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
constexpr int ET_WAVES = 8;               // waves per workgroup (one workgroup per CU): two per SIMD
constexpr int ET_THREADS = 64 * ET_WAVES;
constexpr int ET_D = 4;                   // k-steps the operand loads run ahead (ring of register sets; 2, 4 or 8)
constexpr int ET_CHUNK = 32;              // columns per wave chunk
constexpr int ET_MAXH = 7;                // head rows (3 + 2 z_dim)

template <int ACT>
__device__ __forceinline__ float et_act(float v, float slope) {
    if (ACT == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (ACT == ACT_TANH) return tanhf(v);
    return v;
}

// row of accumulator register r of row tile i in lane half kh
__device__ __forceinline__ int et_row(int i, int r, int kh) { return 32 * i + 8 * (r >> 2) + (r & 3) + 4 * kh; }

// split of the 8 k-values a lane holds for its column into the B-fragment cells of one 16-k step
// (h3: the values scaled by the power of two s, two fp16 parts)
__device__ __forceinline__ void et_split2h(const float (&x)[8], float s, Cell16 (&bf)[3]) {
    const float y[8] = {x[0] * s, x[1] * s, x[2] * s, x[3] * s, x[4] * s, x[5] * s, x[6] * s, x[7] * s};
    split2hx8(y, bf[0], bf[1]);
}
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
    const uint4* wp = Ws + (2 * t + kh) * ET_C + nl + 32 * i;
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p].u = wp[p * 16 * ET_C];
}
template <int NP>
__device__ __forceinline__ void et_step_mfma(f32x16 (&acc)[4], const uint4* __restrict__ Ws, int t, int kh, int nl,
                                             Cell16 (&a0)[3], Cell16 (&a1)[3], const Cell16 (&bf)[3]) {
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
// SH: the activation H is stored (training: the backward reads it); false = inference-mode forward (H == nullptr: the 1.14 GB
// tensor is neither allocated nor written -- only the head rows leave the kernel)
template <int ACT, bool FULL, bool SH = true>
__device__ __forceinline__ void et_fwd_epilogue(f32x16 (&acc)[4], const float* __restrict__ whs, float* __restrict__ H,
                                                long ldh, float* __restrict__ heads, long ldo, const float* __restrict__ bh,
                                                int nh, long n0, long N, int nl, int kh, float slope,
                                                uint4* __restrict__ bitsH) {
    float hs[8];
    unsigned hb[4] = {0u, 0u, 0u, 0u};                   // sign bits of this lane's column, word i = rows 32 i .. 32 i + 31
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
            if (ACT == ACT_LRELU) hb[i] |= v > 0.f ? (1u << (8 * (r >> 2) + (r & 3))) : 0u;   // + 4 kh: shifted below
            if (SH && in0) __builtin_nontemporal_store(v, reinterpret_cast<float*>(hrow + loff));
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
    if (ACT == ACT_LRELU && bitsH) {                     // halves hold rows (.., +4): merge, one 16-byte store per column
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hb[i] <<= 4 * kh;
            hb[i] |= (unsigned)__shfl_xor((int)hb[i], 32, 64);
        }
        if (kh == 0 && in0) bitsH[n0 + nl] = make_uint4(hb[0], hb[1], hb[2], hb[3]);
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
                            float slope, uint4* __restrict__ bitsH, uint4* __restrict__ bitsA, H3Scale hs) {
    // NP == 2 (h3): W3 = tvae_dense_split2h cells; the streamed operand arrives over eight k-steps into the SAME accumulators,
    // so its scale is the tensor's: max |X| from the kernel that produced X (hs.amax_x; dft_out_ring_kernel's epilogue)
    extern __shared__ __attribute__((aligned(16))) uint4 Ws[];            // [NP][16][128]
    __shared__ __attribute__((aligned(16))) float whs[ET_C * 8];          // row m: Wh[0..6][m], b2[m]
    __shared__ float ia_sm[NP == 2 ? ET_C : 1];                           // h3: inverse scale of row m of W2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kh = lane >> 5;
    for (int i = tid; i < NP * 16 * ET_C; i += ET_THREADS) {
        const int row = i & (ET_C - 1), po = i >> 7;                      // po = part * 16 + octet
        Ws[i] = W3[(long)po * Rpad + row];
    }
    if (NP == 2 && tid < ET_C) ia_sm[tid] = h3_inv(h3_scale(hs.amax_a[tid]));
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
    et_load_a<NP>(Ws, 0, 0, kh, nl, a0);
    // Round 4: hs.amax_a = one maximum per row of W2 (tvae_dense_split2h), hs.amax_x = one per CHANNEL of the streamed operand
    // (its producer's epilogue).  The channels are this GEMM's reduction index, so the operand takes the largest of them;
    // the rows of W2 are rows of the product: each accumulator row gets its own inverse (ia_sm).
    float sx = 1.f;
    if (NP == 2) sx = h3_scale(h3_wave_max(fmaxf(hs.amax_x[lane], hs.amax_x[64 + lane])));
    const float ix = h3_inv(sx);
    for (long ci = 0; ci < my; ++ci) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        unsigned ab[4] = {0u, 0u, 0u, 0u};               // sign bits of the INPUT column (rows 16 t + 8 kh + j)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Cell16 bf[3];
            if (bitsA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ab[t >> 1] |= x[t % ET_D][j] > 0.f ? (1u << (16 * (t & 1) + j)) : 0u;
            }
            if (NP == 2) et_split2h(x[t % ET_D], sx, bf);
            else et_split<NP>(x[t % ET_D], bf);
            load_x(ci + (t + ET_D) / 8, (t + ET_D) % 8, x[t % ET_D]);
            et_step_mfma<NP>(acc, Ws, t, kh, nl, a0, a1, bf);
        }
        if (NP == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = (acc[i][r] * ia_sm[et_row(i, r, kh)]) * ix;
        }
        const long n0 = (gw + ci * gstride) * ET_CHUNK;
        if (bitsA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ab[i] <<= 8 * kh;
                ab[i] |= (unsigned)__shfl_xor((int)ab[i], 32, 64);
            }
            if (kh == 0 && n0 + nl < N) bitsA[n0 + nl] = make_uint4(ab[0], ab[1], ab[2], ab[3]);
        }
#define TVAE_ET_EPI(A_)                                                                                    \
    do {                                                                                                  \
        if (!H) et_fwd_epilogue<A_, false, false>(acc, whs, H, ldh, heads, ldo, bh, nh, n0, N, nl, kh, slope, bitsH); \
        else if (n0 + ET_CHUNK <= N) et_fwd_epilogue<A_, true>(acc, whs, H, ldh, heads, ldo, bh, nh, n0, N, nl, kh, slope, bitsH); \
        else et_fwd_epilogue<A_, false>(acc, whs, H, ldh, heads, ldo, bh, nh, n0, N, nl, kh, slope, bitsH); \
    } while (0)
        if (act == ACT_LRELU) TVAE_ET_EPI(ACT_LRELU);
        else if (act == ACT_TANH) TVAE_ET_EPI(ACT_TANH);
        else TVAE_ET_EPI(ACT_NONE);
#undef TVAE_ET_EPI
    }
}

// ------------------------------------------------------------------------------------------
// Data gradient through the tail (LeakyReLU):  dA1 = act'(A1) . W2^T dH,  dH = act'(H) . Wh^T dheads  -- dH is never
// stored.  Inputs per column: nh head gradients and the two 128-bit sign words the forward kernel wrote; the only large
// tensor touched is the output.  Two chained GEMMs per 32-column chunk with a register hand-off:
//   G = Wh^T dheads   (128 x 32, k = head row: 24 MFMAs) lands in the accumulator layout (lane = column, register r of
//   row tile i = row 32 i + 8 (r >> 2) + (r & 3) + 4 kh); registers 8 u' .. 8 u' + 7 of a tile, masked by the sign of H,
//   ARE the B fragment of a 16-k step over rows 16 u .. 16 u + 15 in the k order
//       slot (kh, j)  <->  row 16 u + 8 (j >> 2) + 4 kh + (j & 3),
//   so the host splits W2^T with its k columns permuted the same way (W3p) and nothing is transposed or exchanged.
// Per iteration: [steps of chunk c] [inputs of chunk c + 1 consumed, loads of c + 2 issued] [G of c + 1] [stores of c]:
// every load is consumed before the stores that follow it are issued (loads and stores share one in-order counter per
// kind, and a wait behind mixed traffic drains both).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float et_mask(unsigned w, int bit, float slope) { return (w >> bit) & 1u ? 1.f : slope; }

// NP == 2 (h3): the skinny first GEMM (one 16-k step) stays in the exact three-part split (Wh3: tvae_dense_split3); the
// 128 x 128 second one runs on two fp16 parts (W3p: tvae_dense_split2h, its maximum at amax_a).  Its streamed operand G is
// complete in this wave's registers before the first product, so its scale is EXACT and local: the power of two that brings
// max |G| of the wave's 128 x 32 chunk below 2^15, undone on that chunk's accumulators -- no tensor-wide maximum needed.
template <int NP>
static __global__ __launch_bounds__(ET_THREADS, ET_WAVES / 4)
void enc_tail_dgrad_x6_kernel(const uint4* __restrict__ W3p, int Rpad, const uint4* __restrict__ Wh3, int Rpadh,
                              const float* __restrict__ dheads, long ldd, int nh, const uint4* __restrict__ bitsH,
                              const uint4* __restrict__ bitsA, float* __restrict__ dA1, long lda, long N, float slope,
                              const float* __restrict__ amax_a) {
    constexpr int NP1 = NP == 2 ? 3 : NP;                // parts of the first (head-row) GEMM
    extern __shared__ __attribute__((aligned(16))) uint4 Ws[];            // [NP][16][128] W2^T (permuted k), then [NP1][2][128] Wh^T
    uint4* Whs = Ws + NP * 16 * ET_C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kh = lane >> 5;
    for (int i = tid; i < NP * 16 * ET_C; i += ET_THREADS) Ws[i] = W3p[(long)(i >> 7) * Rpad + (i & (ET_C - 1))];
    for (int i = tid; i < NP1 * 2 * ET_C; i += ET_THREADS) Whs[i] = Wh3[(long)(i >> 7) * Rpadh + (i & (ET_C - 1))];
    __shared__ float ia_sm[NP == 2 ? ET_C : 1];          // h3: inverse scale of row c of W2^T (one maximum per row: tvae_dense_split2h)
    if (NP == 2 && tid < ET_C) ia_sm[tid] = h3_inv(h3_scale(amax_a[tid]));
    __syncthreads();

    const long nchunks = (N + ET_CHUNK - 1) / ET_CHUNK;
    const long gw = (long)blockIdx.x * (ET_THREADS / 64) + wave, gstride = (long)gridDim.x * (ET_THREADS / 64);
    if (gw >= nchunks) return;
    const long my = (nchunks - 1 - gw) / gstride + 1;

    float dh[8];
    uint4 wHn, wAn;
    auto load_in = [&](long ci) {
        if (ci >= my) ci = my - 1;
        const long n = min((gw + ci * gstride) * ET_CHUNK + nl, N - 1);
#pragma unroll
        for (int o = 0; o < ET_MAXH; ++o) dh[o] = o < nh ? dheads[(long)o * ldd + n] : 0.f;
        dh[7] = 0.f;
        wHn = bitsH[n];
        wAn = bitsA[n];
    };
    Cell16 dcell[3];
    unsigned wHs[4], wAs[4], wAsn[4];
    auto consume = [&]() {                               // everything loaded is read here (before the next stores)
        et_split<NP1>(dh, dcell);
        if (kh) {                                        // k slots 8 .. 15 of the head-row reduction do not exist
#pragma unroll
            for (int p = 0; p < NP1; ++p) dcell[p].u = make_uint4(0u, 0u, 0u, 0u);
        }
        wHs[0] = wHn.x >> (4 * kh); wHs[1] = wHn.y >> (4 * kh); wHs[2] = wHn.z >> (4 * kh); wHs[3] = wHn.w >> (4 * kh);
        wAsn[0] = wAn.x >> (4 * kh); wAsn[1] = wAn.y >> (4 * kh); wAsn[2] = wAn.z >> (4 * kh); wAsn[3] = wAn.w >> (4 * kh);
    };
    f32x16 acc[4], G[4];
    float gs = 1.f;                                      // h3: scale of the chunk whose G is in registers
    auto g_phase = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Cell16 wc[3];
#pragma unroll
            for (int p = 0; p < NP1; ++p) wc[p].u = Whs[(p * 2 + kh) * ET_C + 32 * i + nl];
#pragma unroll
            for (int r = 0; r < 16; ++r) G[i][r] = 0.f;
            mfma_np<NP1>(G[i], wc, dcell);
        }
        if (NP == 2) {
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(G[i][r]));
            gs = h3_scale(h3_wave_max(mx));
        }
    };
    load_in(0);
    consume();
    load_in(1);
    g_phase();
    Cell16 a0[3], a1[3];
    et_load_a<NP>(Ws, 0, 0, kh, nl, a0);
    const unsigned loff = (unsigned)(4 * kh * lda + nl) * 4u;               // bytes: 32-bit register offset form
    for (long ci = 0; ci < my; ++ci) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wAs[i] = wAsn[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        const float gs_c = gs, inv_c = NP == 2 ? h3_inv(gs) : 1.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * (u & 1) + j;
                xv[j] = G[u >> 1][r] * et_mask(wHs[u >> 1], 8 * (r >> 2) + (r & 3), slope);
            }
            Cell16 bf[3];
            if (NP == 2) et_split2h(xv, gs_c, bf);
            else et_split<NP>(xv, bf);
            et_step_mfma<NP>(acc, Ws, u, kh, nl, a0, a1, bf);
        }
        consume();
        load_in(ci + 2);
        g_phase();
        const long n0 = (gw + ci * gstride) * ET_CHUNK;
        const bool in0 = n0 + nl < N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                char* drow = reinterpret_cast<char*>(dA1 + (long)et_row(i, r, 0) * lda + n0);
                const float v = (NP == 2 ? (acc[i][r] * ia_sm[et_row(i, r, kh)]) * inv_c : acc[i][r]) * et_mask(wAs[i], 8 * (r >> 2) + (r & 3), slope);
                if (in0) __builtin_nontemporal_store(v, reinterpret_cast<float*>(drow + loff));
            }
    }
}


// ------------------------------------------------------------------------------------------
// Weight gradient of conv2 in one pass, dH never stored:
//     dW2[c2][c] = sum_n dH[c2][n] A1[c][n],   dH[c2][n] = act'(H[c2][n]) * sum_h Wh[h][c2] dheads[h][n]
// (reference: autograd of nn.Conv3d(C, C, 1) behind the three 1x1x1 heads, src/models.py:347-358,390-392) from A1, the
// head gradients (7 rows) and the sign words of H that the forward launch stored -- 548 B per column instead of the
// 1 056 B (dH written by tvae_heads_bwd, then dH and A1 read by the fp32-MFMA GEMM) of the unfused pair.
//
// The reduction runs over the 2.2 M columns, so both MFMA operands are "row, 8 consecutive columns" fragments -- 32-byte
// pieces 8.9 MB apart if a lane fetched its own (the per-wave version of round 2 lost to exactly that).  Here the
// workgroup is cooperative: one persistent 8-wave workgroup per CU walks chunks of EW_NC = 32 columns;
//   * A1's 128 x 32 chunk arrives by 16 LDS-DMAs of 1 KB (8 rows x 128 contiguous bytes each: whole lines), the 7 head-
//     gradient rows and the 32 sign-word columns by one more each, one chunk ahead (double buffer, counted per wave);
//   * thread (row, k-octet) builds ONE cell of each operand per chunk: the 8 A1 values of its row from the raw stage,
//     and the 8 dH values of its row formed on the spot (7 FMAs per value against broadcast LDS reads of the head
//     gradients, one mask bit each), split exactly into three bf16 parts, stored as [part][octet][row] cells;
//   * wave (i, jj) owns the 32 x 64 block (row tile i of dH) x (column tiles 2 jj, 2 jj + 1 of A1^T) of dW2: per 16-column
//     step one A fragment and two B fragments (conflict-free 16-byte cell reads), twelve MFMAs;
//   * the 128 x 128 accumulators stay in registers for the workgroup's whole column range; the per-workgroup results go
//     to a slab and a second launch adds the slabs in workgroup order (deterministic, no atomics).
// Columns: N % 32 == 0 (else the unfused path).  LDS 132 KB (two DMA stages, two sets of cells).
// ------------------------------------------------------------------------------------------
constexpr int EW_NC = 32;                              // columns per chunk (two 16-column MFMA steps)
constexpr int EW_RAW = ET_C * EW_NC * 4;               // raw A1 stage of one chunk: [row][32 columns], 16 KB
constexpr int EW_AUX = 1024;                           // head gradients [<= 8 rows][32 columns] (896 B used)
constexpr int EW_BITS = 1024;                          // sign words [32 columns][4] (the DMA's 64 lanes fetch every column twice)
constexpr int EW_STAGE = EW_RAW + EW_AUX + EW_BITS;    // one DMA stage
constexpr int EW_CELLS = 3 * 4 * ET_C * 16;            // one operand of one chunk as cells [part][octet < 4][row]: 24 KB
constexpr int EW_LDS = 2 * EW_STAGE + 4 * EW_CELLS;    // two stages, two sets of (dH cells, A1 cells)

#define TVAE_EW_DMA_X4(dst, src) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory", "m0")

template <int NP>
static __global__ __launch_bounds__(ET_THREADS, 2) void enc_tail_wgrad_x6_kernel(
    const float* __restrict__ A1, long lda, const float* __restrict__ dheads, long ldd, int nh,
    const uint4* __restrict__ bitsH, const float* __restrict__ Wh, float* __restrict__ slabs, long N, float slope,
    const float* __restrict__ amax_a1, const float* __restrict__ amax_dh) {
    // NP == 2 (h3): both operands are split in here and accumulate over ALL chunks, so their scales are the tensors':
    // max |A1| from A1's producer, and for dH = act'(H) . Wh^T dheads the bound max |dheads| * max_row sum_h |Wh[h][row]|
    // (the second factor formed below from the Wh column every thread already holds)
    extern __shared__ __attribute__((aligned(16))) unsigned char ew_sm[];
    __shared__ float red_[NP == 2 ? 2 * ET_C : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ew_sm;
    uint4* cellsD = reinterpret_cast<uint4*>(ew_sm + 2 * EW_STAGE);              // dH cells [set][part][octet][row]: set stride 2 EW_CELLS
    uint4* cellsA = reinterpret_cast<uint4*>(ew_sm + 2 * EW_STAGE + EW_CELLS);   // A1 cells, likewise
    // this workgroup's chunks: a contiguous range (chunk = 32 columns)
    const long nchunks = N / EW_NC;
    const long per = (nchunks + gridDim.x - 1) / gridDim.x;
    const long c_beg = (long)blockIdx.x * per;
    const long c_end = c_beg + per < nchunks ? c_beg + per : nchunks;
    // cell build role: row (of A1 resp. dH) and k-octet of the chunk
    const int row = tid & 127, oct = tid >> 7;
    float wh[ET_MAXH];
#pragma unroll
    for (int h = 0; h < ET_MAXH; ++h) wh[h] = h < nh ? Wh[(long)h * ET_C + row] : 0.f;
    // Round 4: the rows of both operands are rows / columns of dW2, so each gets its own power of two: channel `row` of A1 from
    // its producer's per-channel maximum, row `row` of dH from the bound max |dheads| * sum_h |Wh[h][row]| of that row alone.
    float sA = 1.f, sD = 1.f;
    float* is_sm = reinterpret_cast<float*>(red_);       // [0 .. 128): 1 / sD[row], [128 .. 256): 1 / sA[row]
    if (NP == 2) {
        float rs = 0.f;
#pragma unroll
        for (int h = 0; h < ET_MAXH; ++h) rs += fabsf(wh[h]);
        sA = h3_scale(amax_a1[row]);
        sD = h3_scale(amax_dh[0] * rs);
        if (oct == 0) {
            is_sm[row] = h3_inv(sD);
            is_sm[ET_C + row] = h3_inv(sA);
        }
        __syncthreads();                                 // (a workgroup without chunks reaches the epilogue without another barrier)
    }
    // DMA role (per wave and chunk: two pieces of A1 + one auxiliary piece = 3 instructions, uniform for the counting):
    //   A1 piece g = 2 wave + q: rows 8 g .. 8 g + 7, lane -> (row 8 g + (lane >> 3), 16-byte piece (lane & 7) of the stage row)
    //   aux: wave 1: the sign words (16 bytes) of column (lane & 31); every other wave: head-gradient row (lane >> 3) < nh
    //        (rows beyond nh re-read row 0; waves 2..7 repeat wave 0's piece into the same place: harmless)
    const float* a_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {                        // 16-byte pieces of a row are stored XOR-swizzled by (row >> 1) & 7:
        const int r_ = 8 * (2 * wave + q) + (lane >> 3);     // the cell build's two 16-byte reads per row are then conflict free
        a_src[q] = A1 + (long)r_ * lda + 4 * ((lane & 7) ^ ((r_ >> 1) & 7));
    }
    const int hrow = (lane >> 3) < nh ? (lane >> 3) : 0;
    const float* h_src = dheads + (long)hrow * ldd + 4 * (lane & 7);
    auto dma_chunk = [&](long ch, int stage) {
        const long n0 = ch * EW_NC;
        const unsigned st = lds0 + (unsigned)(stage * EW_STAGE);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* src = a_src[q] + n0;
            const unsigned dst = st + (unsigned)((2 * wave + q) * 1024);
            TVAE_EW_DMA_X4(dst, src);
        }
        if (wave == 1) {                                 // wave-uniform branch: every wave issues exactly one aux DMA
            const uint4* src = bitsH + n0 + (lane & 31);
            const unsigned dst = st + (unsigned)(EW_RAW + EW_AUX);
            TVAE_EW_DMA_X4(dst, src);
        } else {
            const float* src = h_src + n0;
            const unsigned dst = st + (unsigned)EW_RAW;
            TVAE_EW_DMA_X4(dst, src);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int ti = wave >> 1, tj = 2 * (wave & 1), kh = lane >> 5, li = lane & 31;
    // cells of one chunk from its raw stage: the A1 cell and the dH cell of (row, oct)
    auto build = [&](int stage, int cb) {
        const unsigned char* sb = ew_sm + stage * EW_STAGE;
        uint4* cD = cellsD + cb * (2 * EW_CELLS / 16);
        uint4* cA = cellsA + cb * (2 * EW_CELLS / 16);
        {
            const float4* rp = reinterpret_cast<const float4*>(sb + row * (EW_NC * 4));
            const int sw = (row >> 1) & 7;
            const float4 v0 = rp[(2 * oct) ^ sw], v1 = rp[(2 * oct + 1) ^ sw];
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            Cell16 c3[3];
            if (NP == 2) et_split2h(x, sA, c3);
            else et_split<NP>(x, c3);
#pragma unroll
            for (int p = 0; p < NP; ++p) cA[(p * 4 + oct) * ET_C + row] = c3[p].u;
        }
        {   // G = Wh^T dheads (wave-uniform LDS addresses: broadcast reads), masked by the sign bit of H
            const float* hs = reinterpret_cast<const float*>(sb + EW_RAW) + oct * 8;
            const unsigned* bw = reinterpret_cast<const unsigned*>(sb + EW_RAW + EW_AUX) + (row >> 5);
            float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < ET_MAXH; ++h) {
                const float4 d0 = *reinterpret_cast<const float4*>(hs + h * EW_NC);
                const float4 d1 = *reinterpret_cast<const float4*>(hs + h * EW_NC + 4);
                g[0] = __fmaf_rn(wh[h], d0.x, g[0]); g[1] = __fmaf_rn(wh[h], d0.y, g[1]);
                g[2] = __fmaf_rn(wh[h], d0.z, g[2]); g[3] = __fmaf_rn(wh[h], d0.w, g[3]);
                g[4] = __fmaf_rn(wh[h], d1.x, g[4]); g[5] = __fmaf_rn(wh[h], d1.y, g[5]);
                g[6] = __fmaf_rn(wh[h], d1.z, g[6]); g[7] = __fmaf_rn(wh[h], d1.w, g[7]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w = bw[(oct * 8 + j) * 4];
                g[j] *= ((w >> (row & 31)) & 1u) ? 1.f : slope;
            }
            Cell16 c3[3];
            if (NP == 2) et_split2h(g, sD, c3);
            else et_split<NP>(g, c3);
#pragma unroll
            for (int p = 0; p < NP; ++p) cD[(p * 4 + oct) * ET_C + row] = c3[p].u;
        }
    };
    // Software pipeline, ONE barrier per chunk: iteration ch runs the MFMAs of chunk ch (cells built during iteration
    // ch - 1) while the same threads build the cells of chunk ch + 1 from the stage that landed meanwhile, and the DMAs of
    // chunk ch + 2 are in flight into the stage chunk ch has left.  Cells and stages are double buffered.
    if (c_beg < c_end) {
        dma_chunk(c_beg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_chunk(c_beg + 1 < c_end ? c_beg + 1 : c_beg, 1);
        build(0, 0);
    }
    for (long ch = c_beg; ch < c_end; ++ch) {
        const int cb = (int)((ch - c_beg) & 1);          // cells (and stage) of chunk ch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's pieces of chunk ch + 1 have landed
        __syncthreads();                                                 // cells of chunk ch complete; stage cb is free again
        dma_chunk(ch + 2 < c_end ? ch + 2 : c_end - 1, cb);              // (clamped: uniform bookkeeping)
        const uint4* cD = cellsD + cb * (2 * EW_CELLS / 16);
        const uint4* cA = cellsA + cb * (2 * EW_CELLS / 16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Cell16 af[3], bf[2][3];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                af[p].u = cD[(p * 4 + 2 * ks + kh) * ET_C + 32 * ti + li];
                bf[0][p].u = cA[(p * 4 + 2 * ks + kh) * ET_C + 32 * tj + li];
                bf[1][p].u = cA[(p * 4 + 2 * ks + kh) * ET_C + 32 * (tj + 1) + li];
            }
            mfma_np<NP>(acc[0], af, bf[0]);
            mfma_np<NP>(acc[1], af, bf[1]);
        }
        // (unconditional: in the last iteration it rebuilds the clamped re-fetch of the last chunk into the cell set nobody reads
        //  any more -- under `if` the build was a basic block of its own BEHIND the twelve back-to-back MFMAs, i.e. the wave
        //  first stalled at every dependent MFMA issue and only then started ~190 vector instructions with the matrix pipe idle;
        //  in one block the scheduler spreads them between the dependent matrix instructions)
        build(cb ^ 1, cb ^ 1);
        // ask for the interleaving explicitly: one matrix instruction, then a slice of the build's vector / LDS work
#pragma unroll
        for (int g_ = 0; g_ < 4 * NP; ++g_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);       // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);      // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // DS write
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail DMAs still target this workgroup's LDS
    // slab [workgroup][c2][c]: lane (c = 32 (tj + j) + li), register r -> row c2 = et_row(ti, r, kh)
    float* slab = slabs + (long)blockIdx.x * ET_C * ET_C;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            slab[(long)et_row(ti, r, kh) * ET_C + 32 * (tj + j) + li] =
                NP == 2 ? (acc[j][r] * is_sm[et_row(ti, r, kh)]) * is_sm[ET_C + 32 * (tj + j) + li] : acc[j][r];
}

// ------------------------------------------------------------------------------------------
// Round 6: the same cooperative reduction for TWO STORED operands -- dW[r][c] = sum_n D[r][n] A[c][n], D with <= 128 rows, A
// with 128 -- the two weight gradients of the encoder tail with many head rows (dW2 = dH A1^T, dWh = dheads H^T; galaxy
// configuration: 103 head rows, 2.1 M columns at 8 images), which ran as fp32-MFMA GEMMs (0.66 + 0.75 ms for 2 x 2.2 GB).
// h3 arithmetic only (two fp16 parts per operand: the exact three-part split would need 160 KB of cells and stages):
// D under one scale from the word amax_d (max |D|, left by the row-sum pass that reads D anyway), A under one scale per row
// from amax_a[128].  Per chunk and wave four LDS-DMAs (two 1 KB pieces of each operand).  Rows of D beyond rows_d re-read
// row rows_d - 1: they only reach rows of dW nobody reads.  LDS 128 KB.
// ------------------------------------------------------------------------------------------
constexpr int PW_STAGE = 2 * EW_RAW;                   // raw A chunk + raw D chunk
constexpr int PW_CELLS = 2 * 4 * ET_C * 16;            // one operand of one chunk as cells [part < 2][octet < 4][row]: 16 KB
constexpr int PW_LDS = 2 * PW_STAGE + 4 * PW_CELLS;    // two stages, two sets of (D cells, A cells)

static __global__ __launch_bounds__(ET_THREADS, 2) void enc_tail_wgrad_plain_kernel(
    const float* __restrict__ D, long ldd, int rows_d, const float* __restrict__ A, long lda, float* __restrict__ slabs, long N,
    const float* __restrict__ amax_d, const float* __restrict__ amax_a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ew_sm[];
    __shared__ float is_sm[2 * ET_C];                    // [0 .. 128): 1 / sD, [128 .. 256): 1 / sA[row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ew_sm;
    uint4* cellsD = reinterpret_cast<uint4*>(ew_sm + 2 * PW_STAGE);              // [set][D cells | A cells]
    uint4* cellsA = reinterpret_cast<uint4*>(ew_sm + 2 * PW_STAGE + PW_CELLS);
    const long nchunks = N / EW_NC;
    const long per = (nchunks + gridDim.x - 1) / gridDim.x;
    const long c_beg = (long)blockIdx.x * per;
    const long c_end = c_beg + per < nchunks ? c_beg + per : nchunks;
    const int row = tid & 127, oct = tid >> 7;
    const float sA = h3_scale(amax_a[row]);
    const float sD = h3_scale(amax_d[0]);
    if (oct == 0) {
        is_sm[row] = h3_inv(sD);
        is_sm[ET_C + row] = h3_inv(sA);
    }
    __syncthreads();
    // DMA role: piece g = 2 wave + q of each operand: rows 8 g .. 8 g + 7, lane -> (row, 16-byte piece (lane & 7)), pieces of a row
    // stored XOR-swizzled by (row >> 1) & 7 (conflict-free 16-byte reads in the cell build)
    const float* a_src[2];
    const float* d_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r_ = 8 * (2 * wave + q) + (lane >> 3);
        const int rd = r_ < rows_d ? r_ : rows_d - 1;
        a_src[q] = A + (long)r_ * lda + 4 * ((lane & 7) ^ ((r_ >> 1) & 7));
        d_src[q] = D + (long)rd * ldd + 4 * ((lane & 7) ^ ((r_ >> 1) & 7));
    }
    auto dma_chunk = [&](long ch, int stage) {
        const long n0 = ch * EW_NC;
        const unsigned st = lds0 + (unsigned)(stage * PW_STAGE);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* src = a_src[q] + n0;
            const unsigned dst = st + (unsigned)((2 * wave + q) * 1024);
            TVAE_EW_DMA_X4(dst, src);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* src = d_src[q] + n0;
            const unsigned dst = st + (unsigned)(EW_RAW + (2 * wave + q) * 1024);
            TVAE_EW_DMA_X4(dst, src);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int ti = wave >> 1, tj = 2 * (wave & 1), kh = lane >> 5, li = lane & 31;
    auto build = [&](int stage, int cb) {
        const unsigned char* sb = ew_sm + stage * PW_STAGE;
        uint4* cD = cellsD + cb * (2 * PW_CELLS / 16);
        uint4* cA = cellsA + cb * (2 * PW_CELLS / 16);
        const int sw = (row >> 1) & 7;
        {
            const float4* rp = reinterpret_cast<const float4*>(sb + row * (EW_NC * 4));
            const float4 v0 = rp[(2 * oct) ^ sw], v1 = rp[(2 * oct + 1) ^ sw];
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            Cell16 c3[3];
            et_split2h(x, sA, c3);
#pragma unroll
            for (int p = 0; p < 2; ++p) cA[(p * 4 + oct) * ET_C + row] = c3[p].u;
        }
        {
            const float4* rp = reinterpret_cast<const float4*>(sb + EW_RAW + row * (EW_NC * 4));
            const float4 v0 = rp[(2 * oct) ^ sw], v1 = rp[(2 * oct + 1) ^ sw];
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            Cell16 c3[3];
            et_split2h(x, sD, c3);
#pragma unroll
            for (int p = 0; p < 2; ++p) cD[(p * 4 + oct) * ET_C + row] = c3[p].u;
        }
    };
    // (the pipeline of enc_tail_wgrad_x6_kernel: one barrier per chunk, cells and stages double buffered)
    if (c_beg < c_end) {
        dma_chunk(c_beg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_chunk(c_beg + 1 < c_end ? c_beg + 1 : c_beg, 1);
        build(0, 0);
    }
    for (long ch = c_beg; ch < c_end; ++ch) {
        const int cb = (int)((ch - c_beg) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_chunk(ch + 2 < c_end ? ch + 2 : c_end - 1, cb);
        const uint4* cD = cellsD + cb * (2 * PW_CELLS / 16);
        const uint4* cA = cellsA + cb * (2 * PW_CELLS / 16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Cell16 af[3], bf[2][3];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                af[p].u = cD[(p * 4 + 2 * ks + kh) * ET_C + 32 * ti + li];
                bf[0][p].u = cA[(p * 4 + 2 * ks + kh) * ET_C + 32 * tj + li];
                bf[1][p].u = cA[(p * 4 + 2 * ks + kh) * ET_C + 32 * (tj + 1) + li];
            }
            mfma_np<2>(acc[0], af, bf[0]);
            mfma_np<2>(acc[1], af, bf[1]);
        }
        build(cb ^ 1, cb ^ 1);                           // (unconditional, as in enc_tail_wgrad_x6_kernel)
#pragma unroll
        for (int g_ = 0; g_ < 8; ++g_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);      // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // DS write
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* slab = slabs + (long)blockIdx.x * ET_C * ET_C;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            slab[(long)et_row(ti, r, kh) * ET_C + 32 * (tj + j) + li] =
                (acc[j][r] * is_sm[et_row(ti, r, kh)]) * is_sm[ET_C + 32 * (tj + j) + li];
}

// dW2[e] = sum over workgroups of slabs[g][e], in workgroup order
static __global__ void enc_tail_wgrad_total_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dW2) {
    // thread (element e, slab group q < 4): eight loads in flight per thread, fixed summation order
    __shared__ float sm[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int g0 = 8 * q; g0 < nslab; g0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (g0 + u < nslab) s[u] += slabs[(long)(g0 + u) * ET_C * ET_C + e];
    }
    sm[q][threadIdx.x & 63] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (q == 0) dW2[e] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

}  // namespace tvae
