// Dense layers  Y[m][n] = sum_k W(m,k) X[k][n]  in the "x6" arithmetic (every fp32 operand split EXACTLY into three bf16
// numbers, six partial products, fp32 accumulation -- conv_x6_kernels.hpp), second generation.  Forward (A = W) and
// data gradient (A = W^T, X = dY) of the wide decoder layers and the batched spectral contraction of the
// frequency-domain convolution.
//
// Why this shape.  The split-pipe GEMMs of this library are POWER bound, not issue bound: with random operands the
// 32x32x16 kernels run their k-loop at 1.26 PFLOP/s of executed bf16 work at 1.85-2.1 GHz (the chip lowers its clock;
// MI355X_MICROARCH.md "DVFS give-back": 1 247 TF/s is the guide's own random-data figure for that MFMA shape) and an
// A/B on this kernel's predecessor showed the SAME flops issued as v_mfma_f32_16x16x32_bf16 finish the k-loop 12-14 %
// sooner (half the accumulator writes per flop).  So:
//   * MFMA 16x16x32, operands TRANSPOSED: the MFMA "A" operand is the X cell (8 consecutive k of one column n), the "B"
//     operand the weight cell (8 consecutive k of one row m).  D is then [n][m] with lane = (m = lane & 15, n-quad =
//     lane >> 4) and the four accumulator registers of a tile are four CONSECUTIVE n: every epilogue access (store,
//     residual, mask operand) is a 16-byte vector access and a row's bias is one register;
//   * k-step 32 (one MFMA depth): half the barriers per flop of the 16-k kernel;
//   * 512 x 128 tile, eight waves stacked along m (64 x 128 each = 4 x 8 MFMA tiles, 192 MFMAs per step), two per SIMD.
//     Weight cells go straight from L2 to registers, one PASS (two of a wave's four 16-row fragments) ahead, rotating
//     through two register sets -- 48 instead of 96 registers, which is what lets the 32-k step fit two waves per SIMD;
//   * X is split once per 512 output rows by the whole workgroup: thread (k-octet, n) loads its 8 k-values one step
//     ahead, splits them in the MFMA shadows and writes three whole 16-byte cells ([part][octet][n], double buffered);
//     fragment reads are conflict-free ds_read_b128 (ds_read_b128 serves lanes {0-3, 12-15, 20-27} together: the second
//     octet row, 2 KB further, lands on the banks the first leaves free);
//   * PERSISTENT workgroups walk the XCD-aware tile order: the 256 KB of stores a tile issues drain while the next
//     tile's k-loop runs (one tile per workgroup exposed them: 13 us of every 90, measured as the K -> 0 intercept).
// Same operand conventions, fused tails (ColDot, InTail) and implicit operands (VirtGrad, VirtAct) as the first
// generation; results differ from it only by the order of the fp32 accumulation inside a 32-k step.
#pragma once
#include <hip/hip_runtime.h>
#include "dense_x6_kernels.hpp"

namespace tvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D16_THREADS = 512;
constexpr int D16_STAGE_CELLS = 3 * 4 * 128;       // [part][octet][n] cells of one 32-k step: 24 KB
constexpr int D16_PERSIST = 256;                   // resident workgroups (one per CU; a multiple of 8 keeps id & 7 = XCD)

// six partial products into one 16 x 16 accumulator: x = cells of X (rows n of D), w = cells of W (columns m of D)
__device__ __forceinline__ void mfma6_16(f32x4& acc, const Cell16 (&x)[3], const Cell16 (&w)[3]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0].v, w[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0].v, w[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1].v, w[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1].v, w[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0].v, w[2].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[2].v, w[0].v, acc, 0, 0, 0);
}

// sum over the 16 lanes that share lane >> 4 (the m index of a D tile)
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

#ifdef TVAE_D16_STAMPS      // diagnostic build only: shader-clock totals per workgroup and phase (never in the product library)
__device__ unsigned long long d16_stamps[D16_PERSIST][4];
#define D16_STAMP(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#else
#define D16_STAMP(var_)
#endif
#ifndef TVAE_D16_EXP
#define TVAE_D16_EXP 0       // diagnostic timing builds: bit 0 no loop barrier, 1 no B build, 2 no W loads, 3 no X loads (WRONG results)
#endif

// XV: 0 = X is read from memory, 1 = implicit gradient operand (VirtGrad), 2 = implicit first-layer activation (VirtAct)
template <int XV>
static __global__ __launch_bounds__(D16_THREADS, 2)
void dense16_kernel(const uint4* __restrict__ A3, const float* __restrict__ Xg, long ldx, Epilogue epg, int M, int Mpad,
                    int N, int K, int K8pad, TileMap tm, DenseBatch bt, ColDot cd, InTail it, VirtGrad vg, VirtAct va) {
    __shared__ __attribute__((aligned(16))) uint4 Bs[2 * D16_STAGE_CELLS];
    __shared__ __attribute__((aligned(16))) float bsm[DX6_ROWS];
    __shared__ __attribute__((aligned(16))) float wsm_[2 * DX6_ROWS];     // ColDot weights, or InTail (w0, w1) pairs
    __shared__ __attribute__((aligned(16))) float vtab[XV == 1 ? 512 : (XV == 2 ? 2048 : 4)];   // implicit-operand tables
    __shared__ __attribute__((aligned(16))) float cbm_[2 * DX6_ROWS];     // (bc | lb) of the recomputed mask operand
    __shared__ __attribute__((aligned(16))) float cds[8 * 128 * 2];       // cross-wave sums of the fused tails
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o4 = lane >> 4, l16 = lane & 15;
    const int nk = K8pad >> 2;
    const long part_cells = (long)K8pad * Mpad;
    const int ob = tid >> 7, nb = tid & 127;                  // B build role: k-octet ob of the step, column nb
    const int total = (int)tm.grid();
    int tab_m = -1, tab_img_it = -1, tab_img_va = -1;          // what the LDS tables currently hold

    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        int tile_m, tile_n, split_unused;
        if (!tm.decode(vb, tile_m, tile_n, split_unused)) continue;
        Epilogue ep = epg;
        const float* X = Xg;
        int batch = 0;
        if (bt.tiles_per_batch > 0) {
            batch = tile_m / bt.tiles_per_batch;
            X += batch * bt.x_stride;
            ep.C += batch * bt.c_stride;
            if (ep.aux) ep.aux += batch * bt.c_stride;
            if (ep.res) ep.res += batch * bt.c_stride;
        }
        D16_STAMP(ts0);
        const int m0g = tile_m * DX6_ROWS;
        const int m0 = m0g - batch * bt.tiles_per_batch * DX6_ROWS, n0 = tile_n * 128;
        const int img_it = it.bc ? n0 / it.Np : 0, img_va = XV == 2 ? n0 / va.Np : 0;
        const bool new_tables = tile_m != tab_m || img_it != tab_img_it || img_va != tab_img_va;
        tab_m = tile_m; tab_img_it = img_it; tab_img_va = img_va;
        float vg_g = 0.f, va_x0 = 0.f, va_x1 = 0.f;
        if (XV == 1) vg_g = vg.gy[n0 + nb];
        if (XV == 2) {
            va_x0 = va.xr[2 * (long)(n0 + nb)];
            va_x1 = va.xr[2 * (long)(n0 + nb) + 1];
        }
        const bool vtab_lds = K <= 512;

        // weight cells of this lane: fragment f (rows 64*wave + 16*f + l16), part p, octet 4t + o4
        const uint4* w_ptr = A3 + (long)o4 * Mpad + m0g + 64 * wave + l16;
        auto load_w = [&](int t, int pair, Cell16 (&w)[2][3]) {
            const uint4* q = w_ptr + (long)(4 * t) * Mpad + pair * 32;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) w[i][p].u = q[p * part_cells + i * 16];
        };
        const float* x_ptr = X + n0 + nb;
        auto load_x = [&](int t, float (&x)[8]) {
            if (XV == 2) {                                   // recomputed operand: nothing to load
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = 0.f;
                return;
            }
            const int k0 = 32 * t + 8 * ob;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const float v = x_ptr[(long)(k < K ? k : K - 1) * ldx];        // clamped row: no branch, masked below
                x[j] = k < K ? v : 0.f;
            }
        };
        auto build_b = [&](int t, int stage, float (&x)[8]) {
            const int k0 = 32 * t + 8 * ob;
            if (XV == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + j;
                    x[j] = (k < K) ? virt_value(vg, x[j], vtab_lds ? vtab[k & 511] : vg.wo[k], vg_g) : 0.f;
                }
            }
            if (XV == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = (k0 + j) & 511;
                    const float pre = dec_l0_pre(vtab[k], vtab[512 + k], vtab[1024 + k], vtab[1536 + k], va_x0, va_x1);
                    x[j] = (k0 + j < K) ? act_apply(pre, va.act, va.slope) : 0.f;
                }
            }
            Cell16 h, m, l;
            split3x8(x, h, m, l);
            uint4* dst = Bs + stage * D16_STAGE_CELLS + ob * 128 + nb;
            dst[0] = h.u;
            dst[512] = m.u;
            dst[1024] = l.u;
        };

        f32x4 acc[8][4];
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[f][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

        Cell16 wp[2][3], wq[2][3];
        float xa[8], xn[8];
        // operand loads of the first two steps go out FIRST; the per-row tables (constant along a row of tiles, or along
        // an image) are refilled behind them and only when they change -- one memory latency per tile, not three
        load_w(0, 0, wp);
        if (TVAE_D16_EXP & 4) load_w(0, 1, wq);
        if (TVAE_D16_EXP & 8) load_x(0, xn);
        {
            float x0[8];
            load_x(0, x0);
            load_x(nk > 1 ? 1 : 0, xa);
            __syncthreads();                                 // every wave has left the previous tile (tables, cds, Bs)
            if (new_tables) {
                bsm[tid] = (ep.bias && (m0 + tid) < M) ? ep.bias[(m0 + tid) >> ep.bias_shift] : 0.f;
                if (it.xr) {
                    wsm_[2 * tid] = (m0 + tid) < M ? it.wc[2 * (m0 + tid)] : 0.f;
                    wsm_[2 * tid + 1] = (m0 + tid) < M ? it.wc[2 * (m0 + tid) + 1] : 0.f;
                } else {
                    wsm_[tid] = (cd.w && (m0 + tid) < M) ? cd.w[m0 + tid] : 0.f;
                }
                if (it.bc) {
                    cbm_[tid] = (m0 + tid) < M ? it.bc[m0 + tid] : 0.f;
                    cbm_[DX6_ROWS + tid] = (it.lb && (m0 + tid) < M) ? it.lb[(long)img_it * M + m0 + tid] : 0.f;
                }
                if (XV == 1 && tid < K && tid < 512) vtab[tid] = vg.wo[tid];
                if (XV == 2 && tid < K) {
                    vtab[tid] = va.wc[2 * tid];
                    vtab[512 + tid] = va.wc[2 * tid + 1];
                    vtab[1024 + tid] = va.bc[tid];
                    vtab[1536 + tid] = va.lb ? va.lb[(long)img_va * K + tid] : 0.f;
                }
                __syncthreads();                             // tables visible (build_b reads vtab)
            }
            build_b(0, 0, x0);
        }
        __syncthreads();
        D16_STAMP(ts1);
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            const int tn = t + 1 < nk ? t + 1 : t;
            if (!(TVAE_D16_EXP & 4)) load_w(t, 1, wq);       // second fragment pair of THIS step: one pass of flight
            if (!(TVAE_D16_EXP & 8)) load_x(t + 2 < nk ? t + 2 : t, xn);
            const uint4* bs = Bs + cur * D16_STAGE_CELLS + o4 * 128 + l16;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                Cell16 xb[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[p].u = bs[p * 512 + f * 16];
                mfma6_16(acc[f][0], xb, wp[0]);
                mfma6_16(acc[f][1], xb, wp[1]);
                if (f == 3 && !(TVAE_D16_EXP & 2)) build_b(tn, cur ^ 1, xa);   // cells of the next step, in the MFMA shadows
            }
            if (!(TVAE_D16_EXP & 4)) load_w(tn, 0, wp);      // first fragment pair of the NEXT step
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                Cell16 xb[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[p].u = bs[p * 512 + f * 16];
                mfma6_16(acc[f][2], xb, wq[0]);
                mfma6_16(acc[f][3], xb, wq[1]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) xa[j] = xn[j];
            __builtin_amdgcn_sched_barrier(0);
            if (!(TVAE_D16_EXP & 1)) __syncthreads();
        }

        D16_STAMP(ts2);
        // ---- epilogue, straight from the accumulator layout: lane = (m = l16 of fragment g, n-quad o4 of fragment f) ----
        const bool tail_in = it.xr != nullptr, tail_cd = cd.w != nullptr && !tail_in;
        const bool av = it.bc != nullptr;                    // mask operand recomputed (see InTail)
        const long ncol = n0 + 4 * o4;                       // + 16 f: first of this lane's four columns
        const long ccol = ep.ctile ? (long)tile_n * ep.ctile + 4 * o4 : ncol;
        float rs[4][3];
#pragma unroll
        for (int g = 0; g < 4; ++g) rs[g][0] = rs[g][1] = rs[g][2] = 0.f;
        int mloc[4];
        long mrow[4];
        float bv[4], w0v[4], w1v[4], cb0[4], cb1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mloc[g] = 64 * wave + 16 * g + l16;
            mrow[g] = (long)min(m0 + mloc[g], M - 1);
            bv[g] = bsm[mloc[g]];
            w0v[g] = tail_in ? wsm_[2 * mloc[g]] : (tail_cd ? wsm_[mloc[g]] : 0.f);
            w1v[g] = tail_in ? wsm_[2 * mloc[g] + 1] : 0.f;
            cb0[g] = av ? cbm_[mloc[g]] : 0.f;
            cb1[g] = av ? cbm_[DX6_ROWS + mloc[g]] : 0.f;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            float x0[4], x1[4];
            if (tail_in) {                                   // coordinates of this lane's four columns
                const float4 c0 = *reinterpret_cast<const float4*>(it.xr + 2 * (ncol + 16 * f));
                const float4 c1 = *reinterpret_cast<const float4*>(it.xr + 2 * (ncol + 16 * f) + 4);
                x0[0] = c0.x; x1[0] = c0.y; x0[1] = c0.z; x1[1] = c0.w;
                x0[2] = c1.x; x1[2] = c1.y; x0[3] = c1.z; x1[3] = c1.w;
            }
            float ys[4] = {0.f, 0.f, 0.f, 0.f}, g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool mok = m0 + mloc[g] < M;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[f][g][r] + bv[g];
                if (ep.res) {
                    const float4 rv = *reinterpret_cast<const float4*>(ep.res + mrow[g] * ep.ldres + ncol + 16 * f);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (ep.act == ACT_LRELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * ep.slope;
                } else if (ep.act == ACT_TANH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
                }
                if (ep.mask != ACT_NONE) {
                    float a[4];
                    if (av) {                                // recompute the masked layer's output (VirtAct formula)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            a[r] = dec_l0_pre(w0v[g], w1v[g], cb0[g], cb1[g], x0[r], x1[r]);
                            if (ep.mask == ACT_TANH) a[r] = tanhf(a[r]);        // LeakyReLU: only the sign is used
                        }
                    } else {
                        const float4 a4 = *reinterpret_cast<const float4*>(ep.aux + mrow[g] * ep.ldaux + ncol + 16 * f);
                        a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] *= ep.mask == ACT_LRELU ? (a[r] > 0.f ? 1.f : ep.slope) : 1.f - a[r] * a[r];
                }
                if (mok) {
                    if (ep.C)
                        *reinterpret_cast<float4*>(ep.C + mrow[g] * ep.ldc + ccol + 16 * f) =
                            make_float4(v[0], v[1], v[2], v[3]);
                    if (tail_cd) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) ys[r] += w0v[g] * v[r];
                    }
                    if (tail_in) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            rs[g][0] += v[r];
                            rs[g][1] += v[r] * x0[r];
                            rs[g][2] += v[r] * x1[r];
                            g0[r] += w0v[g] * v[r];
                            g1[r] += w1v[g] * v[r];
                        }
                    }
                }
            }
            // column sums over this wave's 64 rows: 16 lanes (m) of each n-quad, then the waves through LDS
            if (tail_cd) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = sum16(ys[r]);
                    if (l16 == 0) cds[wave * 128 + 16 * f + 4 * o4 + r] = s;
                }
            }
            if (tail_in) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s0 = sum16(g0[r]), s1 = sum16(g1[r]);
                    if (l16 == 0) {
                        cds[(wave * 128 + 16 * f + 4 * o4 + r) * 2] = s0;
                        cds[(wave * 128 + 16 * f + 4 * o4 + r) * 2 + 1] = s1;
                    }
                }
            }
        }
        if (tail_in) {
            // row sums over the panel's 128 columns: a lane holds 32 of them, the four n-quads of a row the rest
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    float s = rs[g][q];
                    s += __shfl_xor(s, 16, 64);
                    s += __shfl_xor(s, 32, 64);
                    rs[g][q] = s;
                }
                if (o4 == 0 && m0 + mloc[g] < M) {
                    float* pp = it.part + ((long)tile_n * M + m0 + mloc[g]) * 3;
                    pp[0] = rs[g][0];
                    pp[1] = rs[g][1];
                    pp[2] = rs[g][2];
                }
            }
            __syncthreads();
            if (tid < 256) {
                float g = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) g += cds[w * 256 + tid];
                it.gxr[2 * (long)n0 + tid] = g;
            }
        } else if (tail_cd) {
            __syncthreads();
            if (tid < 128) {
                float y = cd.b ? cd.b[0] : 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) y += cds[w * 128 + tid];
                cd.y[n0 + tid] = y;
            }
        }
#ifdef TVAE_D16_STAMPS
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
            if (tid == 0) {
                d16_stamps[blockIdx.x][0] += ts1 - ts0;
                d16_stamps[blockIdx.x][1] += ts2 - ts1;
                d16_stamps[blockIdx.x][2] += ts3 - ts2;
                d16_stamps[blockIdx.x][3] += 1;
            }
        }
#endif
    }
}

}  // namespace tvae
