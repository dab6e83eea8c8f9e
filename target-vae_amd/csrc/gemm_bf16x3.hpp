// Split-bf16 ("bf16x3") MFMA GEMM core for gfx950: fp32 operands, fp32 accumulate, fp32-level accuracy.
//
// Every fp32 operand x is split on the fly into x = hi + lo with hi = bf16_rne(x), lo = bf16_rne(x - hi)
// (16 significand bits kept), and each product is evaluated as  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with
// three v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator (the dropped a_lo*b_lo term and the split residuals
// are ~2^-17 relative per product, random sign).  Per-clock this is 16/3 = 5.3x the rate of the exact
// v_mfma_f32_32x32x2_f32 path at ~1e-5 relative tensor error, inside the 1e-4 parity gate of the hot path
// (measured per kernel in tests/test_hip_primitives.py).  The exact-f32 core (gemm_f32_mfma.hpp) stays available
// (TVAE_GEMM=f32) and is what the parity tests compare both against the oracle.
//
// Tile 128x128x32 per 256-thread workgroup (2x2 waves, each 2x2 tiles of 32x32).  Staging: every thread owns, per
// operand and k-step, two (x, k-octet) cells = 8 consecutive k for one row/column; it loads them as fp32 (same
// loader policies as the f32 core), splits them (3 VALU / element, v_cvt_pk_bf16_f32) and writes two 16-B cells
// into the fragment-ready LDS image  S[operand][part][k-octet][x]  (ds_write_b128, lanes along x).  MFMA operand
// map for 32x32x16 bf16: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7  ==  one 16-B cell, so fragment
// reads are conflict-free ds_read_b128.  C/D layout equals the f32 MFMA's, so the LDS epilogue is shared.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_f32_mfma.hpp"

namespace tvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int BK3 = 32;                     // k per barrier (two MFMA k-steps of 16)
constexpr int X3_CELLS = 4 * 128;           // [k-octet 4][x 128] 16-B cells per (operand, part)

union Cell16 {
    uint4 u;
    bf16x8 v;
    unsigned w[4];
};

// hi/lo split of 8 fp32 values into two packed bf16x8 cells (round-to-nearest-even on both parts).
__device__ __forceinline__ void split8(const float (&r)[8], Cell16& hi, Cell16& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2v x = {r[2 * q], r[2 * q + 1]};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        hi.w[q] = hb;
        const float h0 = __builtin_bit_cast(float, hb << 16);
        const float h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
        const f32x2v d = {x[0] - h0, x[1] - h1};
        lo.w[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2));
    }
}

// Loader with one (row, k-octet) cell per call for k-contiguous memory: element (x, k) at ptr[x*ld + k].
// x = tid >> 1, kh = tid & 1.  Uses two float4 loads when the row is 16-B aligned and the octet is in range.
struct LoadKContig8 {
    const float* ptr; long ld; int X;
    int x0, x, kh; bool vec;
    __device__ __forceinline__ void init(int x0_, int tid) {
        x0 = x0_; x = tid >> 1; kh = tid & 1;
        vec = ((ld & 3) == 0) && ((reinterpret_cast<size_t>(ptr) & 15) == 0);
    }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        const int k = k0 + kh * 8;
        const int row = x0 + x;
        if (row < X && vec && (k + 8) <= kend && (k & 3) == 0) {
            const float4* p = reinterpret_cast<const float4*>(ptr + (long)row * ld + k);
            const float4 a = p[0], b = p[1];
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
            r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (row < X && (k + j) < kend) ? ptr[(long)row * ld + k + j] : 0.f;
        }
    }
};

// A operand of conv1 wgrad in cell form: row cr = c*R + r, 8 consecutive kr = (img, p) of dY [c][img][r][p].
struct LoadConvDY8 {
    const float* dy; long ld; int M; int R; int P;
    int x0, x, kh; long rowoff; bool rok;
    __device__ __forceinline__ void init(int x0_, int tid) {
        x0 = x0_; x = tid >> 1; kh = tid & 1;
        const int m = x0 + x;
        rok = m < M;
        const int mc = rok ? m : 0;
        const int c = mc / R, rr = mc - c * R;
        rowoff = (long)c * ld + (long)rr * P;
    }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        int kr = k0 + kh * 8;
        int img = kr / P;
        int p = kr - img * P;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r[j] = (rok && (kr + j) < kend) ? dy[rowoff + (long)img * R * P + p] : 0.f;
            if (++p == P) { p = 0; ++img; }
        }
    }
};

template <class AL, class BL>
__global__ __launch_bounds__(GEMM_THREADS, 2)
void gemm_bf16x3_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int kchunk, float* ws, TileMap tm) {
    // [buf 2][operand 2][part 2][k-octet 4][x 128] 16-B cells = 64 KiB
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * 2 * 2 * X3_CELLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n, split;
    if (!tm.decode(blockIdx.x, tile_m, tile_n, split)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * kchunk;
    const int kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg + BK3 - 1) / BK3;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    al.init(m0, tid);
    bl.init(n0, tid);
    float ra[2][8], rb[2][8];
    auto load_tiles = [&](int k0) {
        al.load(ra[0], k0, kend);
        al.load(ra[1], k0 + 16, kend);
        bl.load(rb[0], k0, kend);
        bl.load(rb[1], k0 + 16, kend);
    };
    auto store_tiles = [&](int buf) {
        uint4* base = lds + buf * (4 * X3_CELLS);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            Cell16 hi, lo;
            split8(ra[o], hi, lo);
            base[(0 * 2 + 0) * X3_CELLS + (al.kh + 2 * o) * 128 + al.x] = hi.u;
            base[(0 * 2 + 1) * X3_CELLS + (al.kh + 2 * o) * 128 + al.x] = lo.u;
            split8(rb[o], hi, lo);
            base[(1 * 2 + 0) * X3_CELLS + (bl.kh + 2 * o) * 128 + bl.x] = hi.u;
            base[(1 * 2 + 1) * X3_CELLS + (bl.kh + 2 * o) * 128 + bl.x] = lo.u;
        }
    };
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();

    const int arow = wm * 64 + (lane & 31);
    const int bcol = wn * 64 + (lane & 31);
    const int h = lane >> 5;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < nk;
        if (more) load_tiles(kbeg + (t + 1) * BK3);
        const uint4* base = lds + cur * (4 * X3_CELLS);
#pragma unroll
        for (int ko = 0; ko < 2; ++ko) {
            const int oct = (2 * ko + h) * 128;
            Cell16 ah[2], alo[2], bh[2], blo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i].u = base[0 * X3_CELLS + oct + arow + 32 * i];
                alo[i].u = base[1 * X3_CELLS + oct + arow + 32 * i];
                bh[i].u = base[2 * X3_CELLS + oct + bcol + 32 * i];
                blo[i].u = base[3 * X3_CELLS + oct + bcol + 32 * i];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i].v, bh[j].v, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].v, blo[j].v, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i].v, bh[j].v, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
    }
    tile_epilogue(acc, reinterpret_cast<float*>(lds), ep, m0, M, n0 + (tid & 127), (n0 + (tid & 127)) < N, ws, split, N);
}

template <class AL, class BL>
static hipError_t launch_gemm_bf16x3(AL al, BL bl, const Epilogue& ep, int M, int N, int K, int splits_wanted,
                                     float* ws, long ws_floats, hipStream_t stream) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int tilesM = cdiv(M, BM), tilesN = cdiv(N, BN);
    int splits = splits_wanted < 1 ? 1 : splits_wanted;
    if (splits > 1) {
        const long per = (long)M * N;
        const long cap = ws ? ws_floats / per : 0;
        if (cap < 2) splits = 1; else if (splits > cap) splits = (int)cap;
        if (splits > 65535) splits = 65535;
    }
    int kchunk = cdiv(cdiv(K > 0 ? K : 1, splits), BK3) * BK3;
    splits = cdiv(K > 0 ? K : 1, kchunk);
    const TileMap tm{tilesM, tilesN, splits};
    float* wsp = splits > 1 ? ws : nullptr;
    hipLaunchKernelGGL((gemm_bf16x3_kernel<AL, BL>), dim3(tm.grid()), dim3(GEMM_THREADS), 0, stream, al, bl, ep, M, N,
                       K, kchunk, wsp, tm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (splits > 1) {
        const long total = (long)M * N;
        int blocks = cdiv(total, 64);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)ws, splits, M, N,
                           ep);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace tvae
