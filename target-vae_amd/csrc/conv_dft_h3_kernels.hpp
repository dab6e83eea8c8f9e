// The two transforms along w of the frequency-domain lifting convolution for LARGE frames (galaxy shape: 128 x 128 images,
// frame L = 160, Lh = 81 frequencies, Ho = 129 outputs; reference train_galaxy.py: P16, k = 64, p = 32) on the 16-bit matrix
// pipe in the h3 arithmetic (round 6).
//
// Why: both are small GEMMs with a constant operand (conv_dft_kernels.hpp),
//     out[w][(m,n)]      = sum_{k=(fx,ri)} E[w][k]  T[k][(m,n)]      K = 2 Lh = 162,  129 outputs
//     S'[(fx,ri)][(m,n)] = sum_w           E'[k][w] dY[(m,n)][w]     K = Ho  = 129,  162 outputs
// and the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 157 TF/s) BOUNDS them at this shape: 0.70 TFLOP per transform at 32
// images = 4.5 ms at peak, measured 6.7 / 7.5 ms (dft_out_wide / dft_dy_wide) for 9.9 / 9.9 GB of traffic that HBM moves in
// ~2 ms.  The small frames of the other configurations (Ho = 17 / 33 / 39) have a quarter of the FLOPs per byte and stay on the
// fp32 ring kernels.  Here every operand is two fp16 parts (three v_mfma_f32_32x32x16_f16 per product block, 5.3x the fp32
// rate):
//   * the constant operand E is tabulated once per launch as fragment cells under one power-of-two scale (|E| <= 2 / L resp. 1)
//     and lives in registers (wave q owns the 32 output rows of tile q);
//   * the streamed operand is staged raw in LDS (coalesced loads one tile ahead), every COLUMN (m, n) of the tile gets its own
//     exact power of two from the maximum over its reduction index -- formed in LDS, no producer involved -- and the scaled
//     values are split into cells [part][k-octet][column] by all threads; the inverse powers go onto the accumulators;
//   * a workgroup owns a tile (filter row m, 32 columns), its waves the output row tiles.  Plain loads and barriers: nothing is
//     hand counted.  (184 registers = two waves per SIMD = one workgroup per CU; an instance capped at 168 registers for two
//     resident workgroups spilled 14 / 6 dwords and measured 1.69 / 1.56 ms against 1.54 / 1.46: same-box A/B, round 6.)
// Measured at the galaxy shape, 8 images: output transform 1.89 -> 1.54 ms, dY transform 2.05 -> 1.46 ms (5.2 GB each:
// 3.4 / 3.6 TB/s).
// Accuracy: as everywhere in h3 -- an element 2^j below its column's maximum keeps min(23, 39 - j) bits; what is left to one
// scale is the reduction index (tests/test_hip_primitives.py::test_conv1_dft_matches_fp64 holds the galaxy frame to fp64).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_dft_kernels.hpp"

namespace tvae {

// EOc[wt < NWT][t < KS][part][lane]: cell of 8 reduction values k = 16 t + 8 (lane >> 5) + j, k = 2 fx + ri, of row
//   w = 32 wt + (lane & 31):  s_o * c_fx * norm * (ri ? -sin : cos)(2 pi fx w / L),  s_o = h3_scale(2 norm)
// EDc[kt < NKT][t < WS][part][lane]: cell of 8 reduction values w = 16 t + 8 (lane >> 5) + j of row kk = 32 kt + (lane & 31):
//   2^14 * (ri ? -sin : cos)(2 pi fx w / L), kk = 2 fx + ri.   Entries outside fx < Lh, w < Ho are zero.
static __global__ void dft_wtab_h3_kernel(uint4* __restrict__ EOc, uint4* __restrict__ EDc, int L, int Lh, int Ho, int KS,
                                          int NWT, int WS, int NKT, float norm) {
    const int nEO = NWT * KS * 64, nED = NKT * WS * 64;
    const float so = h3_scale(2.f * norm), sd = h3_scale(1.f);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nEO + nED; i += gridDim.x * blockDim.x) {
        const bool fwd = i < nEO;
        const int ii = fwd ? i : i - nEO;
        const int lane = ii & 63, t = (ii >> 6) % (fwd ? KS : WS), tile = (ii >> 6) / (fwd ? KS : WS);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int fx, ri, w;
            if (fwd) {
                const int k = 16 * t + 8 * (lane >> 5) + j;
                fx = k >> 1; ri = k & 1; w = 32 * tile + (lane & 31);
            } else {
                const int kk = 32 * tile + (lane & 31);
                fx = kk >> 1; ri = kk & 1; w = 16 * t + 8 * (lane >> 5) + j;
            }
            float x = 0.f;
            if (fx < Lh && w < Ho) {
                float sn, cs;
                sincospif(2.0f * (float)((fx * w) % L) / (float)L, &sn, &cs);
                x = ri ? -sn : cs;
                x *= fwd ? (((fx == 0) || (2 * fx == L)) ? 1.f : 2.f) * norm * so : sd;
            }
            v[j] = x;
        }
        Cell16 h, l;
        split2hx8(v, h, l);
        uint4* dst = (fwd ? EOc : EDc) + (long)(ii >> 6) * 128 + lane;
        dst[0] = h.u;
        dst[64] = l.u;
    }
}

// LDS atomic maximum of non-negative floats (they order like their bit patterns)
__device__ __forceinline__ void lds_amax(float* slot, float v) {
    atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(v));
}

// out[c][b][r][h][w] = act(bias[c] + sum_k E[w][k] T[k][(m,n)]).  blockDim = 64 NWT (wave wt = 32 output columns w).
// KS = ceil(2 Lh / 16) k-steps; NLD = float4 pieces of a tile per thread (16 Lh <= NLD * 64 NWT; 64 NWT a multiple of 8).
// LDS (floats): raw [16 KS][32] | cells [2][2 KS][32] x 4 | cmax [32] | patch [NWT][32 x 33]
template <int KS, int NWT, int NLD>
static __global__ __launch_bounds__(64 * NWT) void dft_out_h3_kernel(const float* __restrict__ T, const uint4* __restrict__ EOc,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int M, int R, int B, int Ho, int Lh, long NBpad, int act,
                                                                     float slope, float eo_inv, float* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float sm_h[];
    constexpr int NTHR = 64 * NWT, KR = 16 * KS;
    float* raw = sm_h;                                               // [k][32 columns]; rows >= 2 Lh stay zero
    uint4* cells = reinterpret_cast<uint4*>(sm_h + KR * 32);         // [part][k-octet < 2 KS][32 columns]
    float* cmax = sm_h + KR * 32 + 2 * 2 * KS * 32 * 4;              // [32]
    float* patch = cmax + 32 + (threadIdx.x >> 6) * (32 * 33);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wt = __builtin_amdgcn_readfirstlane(tid >> 6);
    Cell16 ea[KS][2];                                                // this wave's rows of E, for the whole kernel
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) ea[t][p].u = EOc[((long)(wt * KS + t) * 2 + p) * 64 + lane];
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    const long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long t_beg = (long)blockIdx.x * per, t_end = min(ntiles, t_beg + per);
    if (t_beg >= t_end) return;
    const int npc = Lh * 16;                                         // pieces of a tile: (fx, ri) rows x 8 float4
    float4 stage[NLD];
    auto tile_load = [&](long tile) __attribute__((always_inline)) {
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const float* t0 = T + dft_t_off(n0, m, 2 * M, Lh);
        const float* t1 = T + dft_t_off(n0, M + m, 2 * M, Lh);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int pc = min(i * NTHR + tid, npc - 1);              // (past the end: the last piece again)
            const int q4 = pc & 7, rr = pc >> 3, fx = rr >> 1;
            typedef float f4v __attribute__((ext_vector_type(4)));      // (T is read exactly once: nontemporal)
            const f4v v_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(((rr & 1) ? t1 : t0) + (long)fx * 128 + 4 * q4));
            stage[i] = make_float4(v_.x, v_.y, v_.z, v_.w);
        }
    };
    // raw stage + this thread's share of the column maxima: NTHR is a multiple of 8, so a thread's pieces always cover the
    // same four columns 4 (tid & 7) .. + 3
    auto tile_put = [&]() __attribute__((always_inline)) {
        float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* dst = reinterpret_cast<float4*>(raw);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int pc = i * NTHR + tid;
            if (pc < npc) {
                dst[pc] = stage[i];                                  // piece pc = ((fx * 2 + ri) * 8 + q4): [k][32] floats, linear
                mx.x = fmaxf(mx.x, fabsf(stage[i].x)); mx.y = fmaxf(mx.y, fabsf(stage[i].y));
                mx.z = fmaxf(mx.z, fabsf(stage[i].z)); mx.w = fmaxf(mx.w, fabsf(stage[i].w));
            }
        }
        float* cm = cmax + 4 * (tid & 7);
        lds_amax(cm, mx.x); lds_amax(cm + 1, mx.y); lds_amax(cm + 2, mx.z); lds_amax(cm + 3, mx.w);
    };
    for (int i = tid; i < KR * 32; i += NTHR) raw[i] = 0.f;
    if (tid < 32) cmax[tid] = 0.f;
    __syncthreads();
    tile_load(t_beg);
    float amx = 0.f;
    int c_prev = -1;
    for (long tile = t_beg; tile < t_end; ++tile) {
        tile_put();
        __syncthreads();                                             // raw and the column maxima of this tile are complete
        if (tile + 1 < t_end) tile_load(tile + 1);                   // in flight under the split, the products and the stores
        const float scol = h3_scale(cmax[j]);                        // this lane's output column
        for (int it = tid; it < 2 * KS * 32; it += NTHR) {           // cells: k-octet o, column cl
            const int o = it >> 5, cl = it & 31;
            const float s = h3_scale(cmax[cl]);
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = raw[(8 * o + q) * 32 + cl] * s;
            Cell16 h, l;
            split2hx8(v, h, l);
            cells[o * 32 + cl] = h.u;
            cells[(2 * KS + o) * 32 + cl] = l.u;
        }
        __syncthreads();                                             // cells complete; cmax consumed
        if (tid < 32) cmax[tid] = 0.f;                               // (the next tile's maxima arrive after this tile's last barrier)
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            Cell16 b[3];
            b[0].u = cells[(2 * t + kh) * 32 + j];
            b[1].u = cells[(2 * KS + 2 * t + kh) * 32 + j];
            Cell16 a3[3];
            a3[0] = ea[t][0]; a3[1] = ea[t][1];
            mfma3h(acc, a3, b);
        }
        const int c = m / R, r_ = m - c * R;
        if (c != c_prev && c_prev >= 0) h3_tile_flush_rd(amx, amax + c_prev, lane);
        c_prev = c;
        if (n0 < NB) {                                               // (a tile of pure padding columns has no outputs)
            const float bv = bias ? bias[c] : 0.f;
            const float inv = eo_inv * h3_inv(scol);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[r] * inv + bv;
                if (act == ACT_LRELU) x = x > 0.f ? x : x * slope;
                else if (act == ACT_TANH) x = tanhf(x);
                patch[j * 33 + (r & 3) + 8 * (r >> 2) + 4 * kh] = x;
            }
            __builtin_amdgcn_wave_barrier();
            const int tmax = (int)(NB - n0 < 32 ? NB - n0 : 32);
            const int wn = min(32, Ho - 32 * wt);                    // valid output columns of this wave's tile
            const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
            float* obase = out + (((long)c * B + b0) * R + r_) * P + (long)h0 * Ho + 32 * wt + j;
            const long jump = (long)(R - 1) * P;
#pragma unroll 4
            for (int col = kh; col < tmax; col += 2) {
                if (j < wn) {
                    const int nb = Ho >= 32 ? (h0 + col >= Ho ? 1 : 0) : (h0 + col) / Ho;      // image boundaries before this column
                    const float sv = patch[col * 33 + j];
                    amx = fmaxf(amx, fabsf(sv));
                    __builtin_nontemporal_store(sv, obase + (long)col * Ho + nb * jump);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                             // raw and cells are free for the next tile
    }
    if (c_prev >= 0) h3_tile_flush_rd(amx, amax + c_prev, lane);
}

// S'[(fx,ri)][(m,n)] = sum_w E'[(fx,ri)][w] dY[(m,n)][w].  blockDim = 64 NKT (wave kt = 32 rows kk = 2 fx + ri).
// WS = ceil(Ho / 16) w-steps; NLD = staged dwords per thread and tile (32 Ho <= NLD * 64 NKT).
// LDS (floats): raw [32][PW], PW = Ho | 1 | cells [2][2 WS][32] x 4 | cmax [32]
template <int WS, int NKT, int NLD>
static __global__ __launch_bounds__(64 * NKT) void dft_dy_h3_kernel(const float* __restrict__ dY, const uint4* __restrict__ EDc,
                                                                    float* __restrict__ Sp, int M, int R, int B, int Ho, int Lh,
                                                                    long NBpad, float ed_inv, float* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float sm_h[];
    constexpr int NTHR = 64 * NKT;
    const int PW = Ho | 1;
    const int raw_floats = (32 * PW + 3) & ~3;
    float* raw = sm_h;
    uint4* cells = reinterpret_cast<uint4*>(sm_h + raw_floats);      // [part][w-octet < 2 WS][32 columns]
    float* cmax = sm_h + raw_floats + 2 * 2 * WS * 32 * 4;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int kt = __builtin_amdgcn_readfirstlane(tid >> 6);
    Cell16 ea[WS][2];
#pragma unroll
    for (int t = 0; t < WS; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) ea[t][p].u = EDc[((long)(kt * WS + t) * 2 + p) * 64 + lane];
    const long tiles_n = NBpad / 32, ntiles = (long)M * tiles_n, NB = (long)B * Ho;
    const int P = Ho * Ho;
    const long jump = (long)(R - 1) * P;
    const long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long t_beg = (long)blockIdx.x * per, t_end = min(ntiles, t_beg + per);
    if (t_beg >= t_end) return;
    const int nel = 32 * Ho;
    float stage[NLD];
    int dsto[NLD];                                                   // element e = i NTHR + tid is the same (column, w) in every tile
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = i * NTHR + tid;
        dsto[i] = e + (e / Ho) * (PW - Ho);
    }
    auto tile_load = [&](long tile) __attribute__((always_inline)) { // the tile's 32 Ho values (zeros past the batch)
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        const int c = m / R, r_ = m - c * R;
        const int b0 = (int)(n0 / Ho), h0 = (int)(n0 - (long)b0 * Ho);
        const float* base = dY + ((((long)c * B + b0) * R + r_) * P + (long)h0 * Ho);
        const long cnt = n0 < NB ? (NB - n0 < 32 ? NB - n0 : 32) * Ho : 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = i * NTHR + tid;
            float v = 0.f;
            if (e < nel && e < cnt) {
                const int x = h0 * Ho + e;                           // image boundaries before this element: at most one when Ho >= 32
                const int nb = Ho >= 32 ? (x >= P ? 1 : 0) : x / P;
                v = base[e + nb * jump];
            }
            stage[i] = v;
        }
    };
    auto tile_put = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = i * NTHR + tid;
            if (e < nel) raw[dsto[i]] = stage[i];
        }
    };
    for (int i = tid; i < raw_floats; i += NTHR) raw[i] = 0.f;
    if (tid < 32) cmax[tid] = 0.f;
    __syncthreads();
    tile_load(t_beg);
    float mx = 0.f;
    int m_prev = -1;
    for (long tile = t_beg; tile < t_end; ++tile) {
        tile_put();
        __syncthreads();                                             // raw complete
        if (tile + 1 < t_end) tile_load(tile + 1);
        {   // column maxima: thread (column tid & 31, group tid >> 5) walks its share of the column's Ho values
            float cm = 0.f;
            const float* rc = raw + (tid & 31) * PW;
            for (int w = tid >> 5; w < Ho; w += NTHR / 32) cm = fmaxf(cm, fabsf(rc[w]));
            lds_amax(cmax + (tid & 31), cm);
        }
        __syncthreads();                                             // column maxima complete
        const float scol = h3_scale(cmax[j]);
        for (int it = tid; it < 2 * WS * 32; it += NTHR) {           // cells: w-octet o, column cl
            const int o = it >> 5, cl = it & 31;
            const float s = h3_scale(cmax[cl]);
            const float* rc = raw + cl * PW + 8 * o;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (8 * o + q < Ho) ? rc[q] * s : 0.f;
            Cell16 h, l;
            split2hx8(v, h, l);
            cells[o * 32 + cl] = h.u;
            cells[(2 * WS + o) * 32 + cl] = l.u;
        }
        __syncthreads();                                             // cells complete; cmax consumed
        if (tid < 32) cmax[tid] = 0.f;
        const int m = (int)(tile / tiles_n);
        const long n0 = (tile - (long)m * tiles_n) * 32;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < WS; ++t) {
            Cell16 b[3];
            b[0].u = cells[(2 * t + kh) * 32 + j];
            b[1].u = cells[(2 * WS + 2 * t + kh) * 32 + j];
            Cell16 a3[3];
            a3[0] = ea[t][0]; a3[1] = ea[t][1];
            mfma3h(acc, a3, b);
        }
        if (m != m_prev && m_prev >= 0) h3_tile_flush_rd(mx, amax + m_prev, lane);
        m_prev = m;
        const float inv = ed_inv * h3_inv(scol);
        float* p0 = Sp + dft_t_off(n0 + j, m, 2 * M, Lh);
        float* p1 = Sp + dft_t_off(n0 + j, M + m, 2 * M, Lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {                               // row kk = 2 fx + ri = 32 kt + (r & 3) + 8 (r >> 2) + 4 kh
            const int kk = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int fx = kk >> 1;
            const float v = acc[r] * inv;
            if (fx < Lh) {
                ((kk & 1) ? p1 : p0)[(long)fx * 128] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        }
        __syncthreads();                                             // raw and cells are free for the next tile
    }
    if (m_prev >= 0) h3_tile_flush_rd(mx, amax + m_prev, lane);
}

}  // namespace tvae
