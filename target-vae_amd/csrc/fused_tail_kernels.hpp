// Fused HBM-bound kernels around the skinny ends of the two MLPs (decoder last / first layer, encoder head
// projection).  Every [feature][batch*position] activation touched here is 1-2 GB at the headline batch, so the
// cost of this part of the step is the number of passes over those arrays; each kernel below replaces 2-3 separate
// passes (a skinny product, an activation mask and one or two row reductions) by a single one.
//
// Shared structure ("panel kernel"): a 256-thread workgroup owns a panel of CW consecutive columns x all rows.
// Wave w walks rows w, w+4, ...; lane l owns CPL columns of the panel as CPL/4 float4 groups (group q = columns
// q*256 + 4*l .. +3), so every global access is a 1-KiB contiguous wave segment.  Column-direction sums stay in
// registers across the row loop; row-direction sums are wave64 shuffle reductions amortised over CPL columns and
// written as per-panel partials part[panel][row][v], which a second tiny kernel adds up in a fixed order (bitwise
// reproducible, no atomics).  Out-of-range columns are loaded as zero and never stored.
#pragma once
#include <hip/hip_runtime.h>
#include "small_kernels.hpp"

namespace tvae {

constexpr int PANEL16 = 1024;   // 64 lanes x 16 columns
constexpr int PANEL8 = 512;     // 64 lanes x 8 columns

__device__ __forceinline__ float4 load4(const float* __restrict__ row, long col, long cend, bool vec) {
    if (vec && col + 3 < cend) return *reinterpret_cast<const float4*>(row + col);
    float4 v;
    v.x = col < cend ? row[col] : 0.f;
    v.y = col + 1 < cend ? row[col + 1] : 0.f;
    v.z = col + 2 < cend ? row[col + 2] : 0.f;
    v.w = col + 3 < cend ? row[col + 3] : 0.f;
    return v;
}
__device__ __forceinline__ void store4(float* __restrict__ row, long col, long cend, bool vec, float4 v) {
    if (vec && col + 3 < cend) { *reinterpret_cast<float4*>(row + col) = v; return; }
    if (col < cend) row[col] = v.x;
    if (col + 1 < cend) row[col + 1] = v.y;
    if (col + 2 < cend) row[col + 2] = v.z;
    if (col + 3 < cend) row[col + 3] = v.w;
}
__device__ __forceinline__ float f4get(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// out[v*M + m] = sum_panel part[(panel*M + m)*NV + v]      (one workgroup per row m; NV <= 16)
// A thread reads the NV contiguous values of its panels in one go (one cache line per panel instead of one per value).
static __global__ void part_total_kernel(const float* __restrict__ part, int npanels, int M, int NV, float* __restrict__ out) {
    __shared__ float sm[16 * 16];
    const int m = blockIdx.x;
    float s[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) s[v] = 0.f;
    for (int p = threadIdx.x; p < npanels; p += blockDim.x) {
        const float* q = part + ((long)p * M + m) * NV;
#pragma unroll
        for (int v = 0; v < 16; ++v)
            if (v < NV) s[v] += q[v];
    }
    block_sum<16>(s, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v)
            if (v < NV) out[(long)v * M + m] = s[v];
    }
}

// The same total in two coalesced stages (round 4; the kernel above reads one 32-byte piece per panel at a stride of M NV
// floats: 41 us for 4 356 panels of 128 x 8).  Stage 1: workgroup g adds its contiguous range of panels, every thread a
// float4 of the [M][NV] block (consecutive threads, consecutive addresses), into partial[g][M NV]; stage 2 adds the
// PT_GROUPS partials in order and transposes to out[v M + m].  Deterministic (fixed ranges, fixed order).
constexpr int PT_GROUPS = 64;
static __global__ __launch_bounds__(256) void part_total_s1_kernel(const float* __restrict__ part, int npanels, int MV4,
                                                                  float* __restrict__ partial) {
    const float4* p4 = reinterpret_cast<const float4*>(part);
    const int per = (npanels + PT_GROUPS - 1) / PT_GROUPS;
    const int pb = blockIdx.x * per, pe = min(npanels, pb + per);
    for (int idx = threadIdx.x; idx < MV4; idx += 256) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
        int p = pb;
        for (; p + 1 < pe; p += 2) {                     // two loads in flight; (a + b) at the end keeps a fixed order
            const float4 u = p4[(long)p * MV4 + idx], v = p4[(long)(p + 1) * MV4 + idx];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (p < pe) {
            const float4 u = p4[(long)p * MV4 + idx];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        reinterpret_cast<float4*>(partial)[(long)blockIdx.x * MV4 + idx] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
static __global__ void part_total_s2_kernel(const float* __restrict__ partial, int M, int NV, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                 // = m NV + v
    if (i >= M * NV) return;
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < PT_GROUPS; ++g) s += partial[(long)g * M * NV + i];
    const int m = i / NV, v = i - m * NV;
    out[(long)v * M + m] = s;
}

// ------------------------------------------------------------------------------------------
// Last decoder layer, backward (reference SpatialGenerator.forward src/models.py:121-123, y = Wo h + bo):
//   D[f][n]         = (sum_o Wo[o*F + f] * gy[n*NO + o]) * act'(H[f][n])       gradient w.r.t. the pre-activation of h
//   part[p][f][0]   = sum_{n in panel p} D[f][n]                               -> bias gradient of the layer producing h
//   part[p][f][1+o] = sum_{n in panel p} H[f][n] * gy[n*NO + o]                -> dWo[o][f]
// ------------------------------------------------------------------------------------------
template <int NO>
static __global__ __launch_bounds__(256) void dec_out_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ Wo,
                                                          const float* __restrict__ H, long ldh, float* __restrict__ D,
                                                          long ldd, int F, long N, int act, float slope,
                                                          float* __restrict__ part, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * PANEL16;
    const long cend = min(N, c0 + PANEL16);
    float g[4][4][NO];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long n = c0 + q * 256 + lane * 4 + e;
#pragma unroll
            for (int o = 0; o < NO; ++o) g[q][e][o] = n < cend ? gy[n * NO + o] : 0.f;
        }
    // gridDim.y slices of the F rows (round 6: with one workgroup per 1 024-column panel the galaxy shape -- 128 panels at 8 images
    // -- left half the chip without a workgroup and the rest with four waves each: 2 TB/s); every (panel, row) is still summed
    // by one wave in the same order
    const int fper = (F + (int)gridDim.y - 1) / (int)gridDim.y;
    const int f_beg = (int)blockIdx.y * fper, f_end = min(F, f_beg + fper);
    for (int f = f_beg + wave; f < f_end; f += 4) {
        float w[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) w[o] = Wo[(long)o * F + f];
        const float* hrow = H + (long)f * ldh;
        float* drow = D + (long)f * ldd;
        float acc[1 + NO];
#pragma unroll
        for (int v = 0; v <= NO; ++v) acc[v] = 0.f;
        float4 h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = load4(hrow, c0 + q * 256 + lane * 4, cend, vec);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float he = f4get(h[q], e);
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    s += w[o] * g[q][e][o];
                    acc[1 + o] += he * g[q][e][o];
                }
                dv[e] = s * act_deriv_from_out(he, act, slope);
                acc[0] += dv[e];
            }
            if (D) store4(drow, c0 + q * 256 + lane * 4, cend, vec, make_float4(dv[0], dv[1], dv[2], dv[3]));
        }
#pragma unroll
        for (int v = 0; v <= NO; ++v) acc[v] = wave_sum(acc[v]);
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v <= NO; ++v) part[((long)blockIdx.x * F + f) * (1 + NO) + v] = acc[v];
        }
    }
}

// ------------------------------------------------------------------------------------------
// First decoder layer without Fourier features, backward (reference src/models.py:107-118: h = act(Wc x' + bc + Wl z)):
// d[f][n] is the pre-activation gradient.  One panel = up to 1024 pixels of ONE image (cpi panels per image).
//   gxr[n][j]          = sum_f Wc[2f + j] * d[f][n]                              gradient w.r.t. the coordinates
//   part[p][f][0..2]   = sum_{n in panel p} d[f][n] * (1, x'_0[n], x'_1[n])      -> per-image sums (latent path),
//                                                                                   bias and coordinate-weight grads
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void dec_in_bwd_kernel(const float* __restrict__ d, long ldd,
                                                         const float* __restrict__ xr, const float* __restrict__ Wc,
                                                         int F, int Np, int cpi, float* __restrict__ gxr,
                                                         float* __restrict__ part, int vec) {
    __shared__ float sm[4 * PANEL16 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int img = blockIdx.x / cpi, ch = blockIdx.x - img * cpi;
    const long c0 = (long)img * Np + (long)ch * PANEL16;
    const long cend = min((long)(img + 1) * Np, c0 + PANEL16);
    float x0[4][4], x1[4][4], gx0[4][4], gx1[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long n = c0 + q * 256 + lane * 4 + e;
            x0[q][e] = n < cend ? xr[2 * n] : 0.f;
            x1[q][e] = n < cend ? xr[2 * n + 1] : 0.f;
            gx0[q][e] = 0.f;
            gx1[q][e] = 0.f;
        }
    for (int f = wave; f < F; f += 4) {
        const float wc0 = Wc[2 * f], wc1 = Wc[2 * f + 1];
        const float* drow = d + (long)f * ldd;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = load4(drow, c0 + q * 256 + lane * 4, cend, vec);
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float de = f4get(v[q], e);
                gx0[q][e] += wc0 * de;
                gx1[q][e] += wc1 * de;
                acc[0] += de;
                acc[1] += de * x0[q][e];
                acc[2] += de * x1[q][e];
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) part[((long)blockIdx.x * F + f) * 3 + k] = acc[k];
        }
    }
    // combine the four waves' column sums (each wave saw a quarter of the rows)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cl = q * 256 + lane * 4 + e;
            sm[(wave * PANEL16 + cl) * 2] = gx0[q][e];
            sm[(wave * PANEL16 + cl) * 2 + 1] = gx1[q][e];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < PANEL16 * 2; i += 256) {
        const long n = c0 + (i >> 1);
        if (n < cend) gxr[2 * n + (i & 1)] = sm[i] + sm[PANEL16 * 2 + i] + sm[2 * PANEL16 * 2 + i] + sm[3 * PANEL16 * 2 + i];
    }
}

// Simg[b][f] = sum_c part[b*cpi + c][f][0];  dbc[f] = sum_b Simg[b][f];  dWc[f][j] = sum_{b,c} part[..][f][1+j]
// Two coalesced stages (round 3; the single-stage form -- one workgroup per feature reading 12-byte pieces 6 KB apart -- took
// 90 us for 50 MB): (1) one workgroup per image adds its cpi panels element-wise, every read a contiguous 3 F floats, leaves
// the image's sums IN ITS FIRST PANEL (part is consumed) and writes Simg; (2) the sums over the images, 3 F outputs.
static __global__ void dec_in_total_img_kernel(float* __restrict__ part, int cpi, int F, float* __restrict__ Simg) {
    const int b = blockIdx.x, n3 = 3 * F;
    float* base = part + (long)b * cpi * n3;
    for (int e = threadIdx.x; e < n3; e += blockDim.x) {
        float s = 0.f;
#pragma unroll 4
        for (int c = 0; c < cpi; ++c) s += base[(long)c * n3 + e];
        base[e] = s;                                     // (only this thread ever touches element e of this image's panels)
        const int f = e / 3;
        if (e - 3 * f == 0) Simg[(long)b * F + f] = s;
    }
}
static __global__ void dec_in_total_sum_kernel(const float* __restrict__ part, int B, int cpi, int F, float* __restrict__ dbc,
                                               float* __restrict__ dWc) {
    __shared__ float sm[4][64];
    const int n3 = 3 * F, e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (e < n3)
        for (int b = q; b < B; b += 4) s += part[(long)b * cpi * n3 + e];
    sm[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && e < n3) {
        const float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
        const int f = e / 3, k = e - 3 * f;
        if (k == 0) dbc[f] = t;
        else dWc[2 * f + k - 1] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Encoder head projection (conv_a / conv_r / conv_z as ONE stacked 1x1x1 convolution, reference
// src/models.py:390-392): NO = 3 + 2*z_dim output rows, far below an MFMA tile, so it is a streaming product.
//   Y[j][n] = b[j] + sum_c W[j*C + c] * X[c][n]            thread = 4 columns, all C rows
// ------------------------------------------------------------------------------------------
template <int NO>
static __global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                        long ldx, const float* __restrict__ bias,
                                                        float* __restrict__ Y, long ldy, int C, long N, int vec) {
    const long col = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= N) return;
    float acc[NO][4];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const float b = bias ? bias[j] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = b;
    }
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float4 x = load4(X + (long)c * ldx, col, N, vec);
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const float w = W[(long)j * C + c];
            acc[j][0] += w * x.x; acc[j][1] += w * x.y; acc[j][2] += w * x.z; acc[j][3] += w * x.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j)
        store4(Y + (long)j * ldy, col, N, vec, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
}

// Backward of the head projection fused with the activation mask of its input and all row reductions:
//   dX[c][n]        = act'(X[c][n]) * sum_j W[j*C + c] * dY[j][n]            (not stored when dX == NULL: sums only)
//   part[p][c][j]   = sum_{n in panel p} dY[j][n] * X[c][n]        -> dW[j][c]
//   part[p][c][NO]  = sum_{n in panel p} dX[c][n]                  -> bias gradient of the layer producing X
template <int NO>
static __global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ W, const float* __restrict__ dY,
                                                        long ldy, const float* __restrict__ X, long ldx,
                                                        float* __restrict__ dX, long lddx, int C, long N, int act,
                                                        float slope, float* __restrict__ part, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * PANEL8;
    const long cend = min(N, c0 + PANEL8);
    float g[2][4][NO];
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 t = load4(dY + (long)j * ldy, c0 + q * 256 + lane * 4, cend, vec);
            g[q][0][j] = t.x; g[q][1][j] = t.y; g[q][2][j] = t.z; g[q][3][j] = t.w;
        }
    for (int c = wave; c < C; c += 4) {
        float w[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) w[j] = W[(long)j * C + c];
        const float* xrow = X + (long)c * ldx;
        float* orow = dX + (long)c * lddx;
        float4 x[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) x[q] = load4(xrow, c0 + q * 256 + lane * 4, cend, vec);
        float acc[NO + 1];
#pragma unroll
        for (int v = 0; v <= NO; ++v) acc[v] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float dv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xe = f4get(x[q], e);
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    s += w[j] * g[q][e][j];
                    acc[j] += g[q][e][j] * xe;
                }
                dv[e] = s * act_deriv_from_out(xe, act, slope);
                acc[NO] += dv[e];
            }
            if (dX) store4(orow, c0 + q * 256 + lane * 4, cend, vec, make_float4(dv[0], dv[1], dv[2], dv[3]));
        }
        if (NO == 7) {                                   // eight sums: one butterfly (10 exchange steps instead of 48)
            float a8[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) a8[v] = acc[v < NO + 1 ? v : 0];
            const float tot = wave_sum8(a8, lane);
            if (lane < 8) part[((long)blockIdx.x * C + c) * (NO + 1) + lane] = tot;
        } else {
#pragma unroll
            for (int v = 0; v <= NO; ++v) acc[v] = wave_sum(acc[v]);
            if (lane == 0) {
#pragma unroll
                for (int v = 0; v <= NO; ++v) part[((long)blockIdx.x * C + c) * (NO + 1) + v] = acc[v];
            }
        }
    }
}

}  // namespace tvae
