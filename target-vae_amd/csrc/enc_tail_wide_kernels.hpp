// Encoder tail with MANY head rows (round 6): conv2 (1x1x1, 128 -> 128) + {conv_a, conv_r, conv_z} stacked to
// nh = 3 + 2 z_dim rows with 8 <= nh <= 128 -- the galaxy configuration's z_dim = 50 gives 103 (reference
// train_galaxy.py:412-420, src/models.py:347-358,390-392).  The kernels of enc_tail_x6_kernels.hpp do the head projection on
// the vector ALU (7 FMAs per element) and stop at 7 rows; beyond that the projection is a GEMM of the size of conv2 itself,
// and until round 6 such shapes ran five separate fp32-MFMA GEMMs (5.1 ms of the 21 ms galaxy step).
//
// Both directions are the SAME chain of two 128 x 128 GEMMs per 32-column chunk with a register hand-off, i.e. the structure
// of enc_tail_dgrad_x6_kernel with a streamed first operand:
//
//   forward        H = act(W2 A1 + b2)            -> stored (training), sign words of H and A1
//                  heads = Wh H + bh              -> rows < nh stored
//   data gradient  dH = act'(H) . (Wh^T dheads)   -> optionally stored (the weight gradients read it)
//                  dA1 = act'(A1) . (W2^T dH)     -> stored
//
//   * both weights are stationary in LDS as fragment cells [part][16 octets][128 rows] (2 x 64 KB in the two-part h3
//     arithmetic; the exact three-part split would need 192 KB: those modes keep the unfused path);
//   * GEMM 1 streams its operand global -> registers in B-fragment order exactly like enc_tail_fwd_x6_kernel (rows beyond
//     the operand's kx are zero fragments against zero weight cells);
//   * its accumulators, after the elementwise stage, ARE the B fragments of GEMM 2 in the k order
//         slot (kh, j) of step u  <->  row 16 u + 8 (j >> 2) + 4 kh + (j & 3)
//     so the host splits the second weight with its k columns permuted that way (tvae/ops.py: _enc_tail_perm);
//   * h3 scales: rows of both weights from their split (tvae_dense_split2h); the streamed operand of GEMM 1 from its
//     producer (per-channel maxima of A1; ONE word max |dheads| from the row-sum pass the backward runs anyway); the operand
//     of GEMM 2 is complete in the wave's registers before its first product: exact local power of two per 128 x 32 chunk.
#pragma once
#include <hip/hip_runtime.h>
#include "enc_tail_x6_kernels.hpp"

namespace tvae {

struct EtWide {
    const uint4* Wa3;      // cells of GEMM 1's weight [NP][K8a octets][RpadA rows]; rows < 128 and octets < 16 are read
    int RpadA, K8a;
    const uint4* Wb3;      // cells of GEMM 2's weight (k permuted), [NP][16][RpadB]
    int RpadB;
    const float* X;        // streamed operand [kx][N]
    long ldx;
    int kx;
    const float* b1;       // forward: b2 [128] (or NULL);  data gradient: unused
    const float* b2;       // forward: bh [m2];             data gradient: unused
    float* Y1;             // forward: H [128][N] or NULL (inference);  data gradient: dH [128][N] or NULL
    long ld1;
    float* Y2;             // forward: heads [m2][N];  data gradient: dA1 [128][N]
    long ld2;
    int m2;                // rows of Y2 that exist
    uint4* bitsH;          // forward: out (or NULL);  data gradient: in
    uint4* bitsA;
    long N;
    float slope;
    const float* amax_wa;  // h3: one maximum per row of Wa (>= 128 words)
    const float* amax_wb;  // h3: one maximum per row of Wb
    const float* amax_x;   // h3: nx words, the operand takes the largest
    int nx;
};

// ACT: the forward's activation (compile time: a run-time switch inside the unrolled elementwise stage would put a uniform
// branch around each of its 64 elements); the data gradient is LeakyReLU's (its mask comes from the sign words).
template <int NP, bool DG, int ACT>
static __global__ __launch_bounds__(ET_THREADS, ET_WAVES / 4) void enc_tail_wide_kernel(EtWide a) {
    extern __shared__ __attribute__((aligned(16))) uint4 Ws[];           // Wa [NP][16][128], then Wb [NP][16][128]
    uint4* Wbs = Ws + NP * 16 * ET_C;
    __shared__ __attribute__((aligned(16))) float tab[4 * ET_C];          // ia_a | ia_b | b1 | b2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kh = lane >> 5;
    for (int i = tid; i < NP * 16 * ET_C; i += ET_THREADS) {
        const int row = i & (ET_C - 1), po = i >> 7, p = po >> 4, o = po & 15;
        Ws[i] = o < a.K8a ? a.Wa3[((long)p * a.K8a + o) * a.RpadA + row] : make_uint4(0u, 0u, 0u, 0u);
        Wbs[i] = a.Wb3[(long)po * a.RpadB + row];
    }
    if (tid < ET_C) {
        tab[tid] = NP == 2 ? h3_inv(h3_scale(a.amax_wa[tid])) : 1.f;
        tab[ET_C + tid] = NP == 2 ? h3_inv(h3_scale(a.amax_wb[tid])) : 1.f;
        tab[2 * ET_C + tid] = (!DG && a.b1) ? a.b1[tid] : 0.f;
        tab[3 * ET_C + tid] = (!DG && a.b2 && tid < a.m2) ? a.b2[tid] : 0.f;
    }
    __syncthreads();
    const float* ia_a = tab;
    const float* ia_b = tab + ET_C;
    const float* bs1 = tab + 2 * ET_C;
    const float* bs2 = tab + 3 * ET_C;

    const long N = a.N;
    const long nchunks = (N + ET_CHUNK - 1) / ET_CHUNK;
    const long gw = (long)blockIdx.x * (ET_THREADS / 64) + wave, gstride = (long)gridDim.x * (ET_THREADS / 64);
    if (gw >= nchunks) return;
    const long my = (nchunks - 1 - gw) / gstride + 1;

    const unsigned ld4 = (unsigned)(a.ldx * 4);                           // bytes per row (32 rows stay below 2^32: host check)
    const unsigned xlane = 8u * (unsigned)kh * ld4;
    // kxv: the operand's row count as the caller of the moment sees it (inside the chunk loop a copy the compiler cannot prove
    // loop invariant: it would otherwise precompute the clamped row offsets of all 8 x 8 loads and spill them)
    auto load_x = [&](long ci, int t, float (&x)[8], int kxv) {
        const int rem = kxv - 16 * t;                                     // rows of this step that exist (uniform)
        if (rem <= 0) {                                                   // none: zero fragments (against zero weight cells)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = 0.f;
            return;
        }
        if (ci >= my) ci = my - 1;                                        // harmless reload of real data
        const long n0 = (gw + ci * gstride) * ET_CHUNK;
        const char* p = reinterpret_cast<const char*>(a.X + (long)(16 * t) * a.ldx + n0);
        const unsigned coff = 4u * (unsigned)min((long)nl, N - 1 - n0);
        if (rem >= 16) {                                                  // uniform row pointer + one 32-bit lane offset
            const unsigned off = xlane + coff;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float*>(p + (long)j * a.ldx * 4 + off);
        } else {                                                          // the operand's last, partial step: clamped rows
#pragma unroll                                                            // (they meet zero weight cells)
            for (int j = 0; j < 8; ++j) {
                const unsigned r = (unsigned)min(8 * kh + j, rem - 1);
                x[j] = *reinterpret_cast<const float*>(p + (r * ld4 + coff));
            }
        }
    };
    // data gradient: the two sign words of the chunk's column, one chunk ahead
    uint4 wHn = make_uint4(0u, 0u, 0u, 0u), wAn = wHn;
    auto load_bits = [&](long ci) {
        if (!DG) return;
        if (ci >= my) ci = my - 1;
        const long n = min((gw + ci * gstride) * ET_CHUNK + nl, N - 1);
        wHn = a.bitsH[n];
        wAn = a.bitsA[n];
    };

    f32x16 acc[4];
    float x[ET_D][8];
#pragma unroll
    for (int t = 0; t < ET_D; ++t) load_x(0, t, x[t], a.kx);
    load_bits(0);
    Cell16 a0[3], a1[3];
    float sx = 1.f;
    if (NP == 2) {
        float m = 0.f;
        for (int i = lane; i < a.nx; i += 64) m = fmaxf(m, a.amax_x[i]);
        sx = h3_scale(h3_wave_max(m));
    }
    const float ix = h3_inv(sx);
    const unsigned loff1 = (unsigned)(4 * kh * a.ld1 + nl) * 4u;          // bytes: (uniform row pointer, 32-bit lane offset) stores
    const unsigned loff2 = (unsigned)(4 * kh * a.ld2 + nl) * 4u;
    for (long ci = 0; ci < my; ++ci) {
        const long n0 = (gw + ci * gstride) * ET_CHUNK;
        const bool in0 = n0 + nl < N;
        // The per-row tables are read from LDS where they are used.  Their addresses depend only on the lane, so the compiler
        // would hoist all 4 x 64 reads out of the chunk loop and keep (spill: 1 KB of scratch per lane) them in registers;
        // an offset it cannot see through makes them loop-variant.
        int lo = 4 * kh;
        asm volatile("" : "+v"(lo));
        // Likewise the 2 x 64 uniform row pointers of the two outputs (Y + row * ld): hoisted out of the chunk loop they are
        // 256 registers' worth of 64-bit values.  Row strides the compiler cannot see through keep them scalar work per use.
        long ld1v = a.ld1, ld2v = a.ld2;
        int kxv = a.kx;
        asm volatile("" : "+s"(ld1v), "+s"(ld2v), "+s"(kxv));
        // ---- GEMM 1: 128 x 32 chunk of  Wa X ----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        unsigned ab[4] = {0u, 0u, 0u, 0u};               // forward: sign bits of the INPUT column (rows 16 t + 8 kh + j)
        et_load_a<NP>(Ws, 0, 0, kh, nl, a0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Cell16 bf[3];
            if (!DG && a.bitsA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ab[t >> 1] |= x[t % ET_D][j] > 0.f ? (1u << (16 * (t & 1) + j)) : 0u;
            }
            if (NP == 2) et_split2h(x[t % ET_D], sx, bf);
            else et_split<NP>(x[t % ET_D], bf);
            load_x(ci + (t + ET_D) / 8, (t + ET_D) % 8, x[t % ET_D], kxv);
            et_step_mfma<NP>(acc, Ws, t, kh, nl, a0, a1, bf);
        }
        // the data gradient's sign words of THIS chunk (loaded one chunk ahead), then the next chunk's
        unsigned wHs[4], wAs[4];
        if (DG) {
            wHs[0] = wHn.x >> (4 * kh); wHs[1] = wHn.y >> (4 * kh); wHs[2] = wHn.z >> (4 * kh); wHs[3] = wHn.w >> (4 * kh);
            wAs[0] = wAn.x >> (4 * kh); wAs[1] = wAn.y >> (4 * kh); wAs[2] = wAn.z >> (4 * kh); wAs[3] = wAn.w >> (4 * kh);
            load_bits(ci + 1);
        }
        // ---- elementwise stage on the accumulators (they become GEMM 2's streamed operand) --------------------------
        unsigned hb[4] = {0u, 0u, 0u, 0u};
        float vmax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = et_row(i, r, 0) + lo;
                float v = NP == 2 ? (acc[i][r] * ia_a[row]) * ix : acc[i][r];
                if (DG) {
                    v *= et_mask(wHs[i], 8 * (r >> 2) + (r & 3), a.slope);
                } else {
                    v += bs1[row];
                    if (ACT == ACT_LRELU) {
                        v = fmaxf(v, v * a.slope);
                        hb[i] |= v > 0.f ? (1u << (8 * (r >> 2) + (r & 3))) : 0u;      // + 4 kh: shifted below
                    } else if (ACT == ACT_TANH) {
                        v = tanhf(v);
                    }
                }
                acc[i][r] = v;
                vmax = fmaxf(vmax, fabsf(v));
                if (a.Y1 && in0) {
                    char* yrow = reinterpret_cast<char*>(a.Y1 + (long)et_row(i, r, 0) * ld1v + n0);
                    __builtin_nontemporal_store(v, reinterpret_cast<float*>(yrow + loff1));
                }
            }
        if (!DG && a.bitsH) {                            // halves hold rows (.., +4): merge, one 16-byte store per column
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hb[i] <<= 4 * kh;
                hb[i] |= (unsigned)__shfl_xor((int)hb[i], 32, 64);
                ab[i] <<= 8 * kh;
                ab[i] |= (unsigned)__shfl_xor((int)ab[i], 32, 64);
            }
            if (kh == 0 && in0) {
                a.bitsH[n0 + nl] = make_uint4(hb[0], hb[1], hb[2], hb[3]);
                a.bitsA[n0 + nl] = make_uint4(ab[0], ab[1], ab[2], ab[3]);
            }
        }
        const float gs = NP == 2 ? h3_scale(h3_wave_max(vmax)) : 1.f;
        const float ig = h3_inv(gs);
        // ---- GEMM 2: registers 8 u' .. 8 u' + 7 of a tile are the B fragment of step u --------------------------------
        // Two row tiles (64 output rows) at a time: with all four the kernel holds 128 accumulator registers beside the
        // operand ring and spills (measured: 1 KB of scratch per lane); the operand fragments are re-split per half (vector
        // ALU work in the shadow of the other wave's MFMAs), a half whose rows do not exist is skipped.
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) if (64 * hf < a.m2) {       // (uniform; no `break`: the loop must unroll completely)
            f32x16 acc2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
            et_load_a<NP>(Wbs, 0, 2 * hf, kh, nl, a0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = acc[u >> 1][8 * (u & 1) + j];
                Cell16 bf[3];
                if (NP == 2) et_split2h(xv, gs, bf);
                else et_split<NP>(xv, bf);
                et_load_a<NP>(Wbs, u, 2 * hf + 1, kh, nl, a1);
                mfma_np<NP>(acc2[0], a0, bf);
                __builtin_amdgcn_sched_barrier(0);
                et_load_a<NP>(Wbs, (u + 1) & 7, 2 * hf, kh, nl, a0);
                mfma_np<NP>(acc2[1], a1, bf);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- output rows of this half ------------------------------------------------------------------------------
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) if (32 * (2 * hf + i2) < a.m2) {      // (uniform) row tiles that exist
                const int i = 2 * hf + i2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = et_row(i, r, 0) + lo;
                    float v = NP == 2 ? (acc2[i2][r] * ia_b[row]) * ig : acc2[i2][r];
                    if (DG) v *= et_mask(wAs[i], 8 * (r >> 2) + (r & 3), a.slope);
                    else v += bs2[row];
                    char* yrow = reinterpret_cast<char*>(a.Y2 + (long)et_row(i, r, 0) * ld2v + n0);
                    if (in0 && row < a.m2) __builtin_nontemporal_store(v, reinterpret_cast<float*>(yrow + loff2));
                }
            }
        }
    }
}

}  // namespace tvae
