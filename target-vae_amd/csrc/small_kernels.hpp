// HBM-bound kernels of the TARGET-VAE hot path for gfx950: rotated filter bank, attention head
// (prior + log-softmax + Gumbel-softmax + expected-value pooling + KL), coordinate transform,
// first/last decoder layers, likelihoods, segmented row/column reductions, fused Adam.
//
// Layout convention (see gemm_f32_mfma.hpp): activations are feature-major [feature][batch*position]
// with the position index contiguous; every kernel here keeps lanes along the position index so
// that global accesses are coalesced 256-B wave segments.  Reductions are wave64 shuffles followed
// by one LDS exchange per workgroup; no float atomics (bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_f32_mfma.hpp"

namespace tvae {

constexpr float EPS_STD = 1e-6f;      // reference train_mnist.py:197

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// Sums EIGHT values over the 64 lanes of a wave in 4 + 2 + 1 + 3 exchange steps (a butterfly that halves the number of
// live values at each of the first three steps) instead of 8 x 6: lane l returns the total of value (l & 7).
__device__ __forceinline__ float wave_sum8(const float (&a)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float b[4], c[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)      // lane keeps index 2i + b0, its partner (lane ^ 1) sends exactly that one
        b[i] = (b0 ? a[2 * i + 1] : a[2 * i]) + dpp_get<0xB1>(b0 ? a[2 * i] : a[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 2; ++i)      // b index 2i + b1 -> original index 4i + 2 b1 + b0
        c[i] = (b1 ? b[2 * i + 1] : b[2 * i]) + dpp_get<0x4E>(b1 ? b[2 * i] : b[2 * i + 1]);
    float d = (b2 ? c[1] : c[0]) + __shfl_xor(b2 ? c[0] : c[1], 4, 64);     // original index 4 b2 + 2 b1 + b0 = lane & 7
    d += __shfl_xor(d, 8, 64);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    return d;
}

// the same for FOUR values (2 + 1 + 4 exchange steps): lane l returns the total of value (l & 3)
__device__ __forceinline__ float wave_sum4(const float (&a)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = (b0 ? a[2 * i + 1] : a[2 * i]) + dpp_get<0xB1>(b0 ? a[2 * i] : a[2 * i + 1]);
    float d = (b1 ? b[1] : b[0]) + dpp_get<0x4E>(b1 ? b[0] : b[1]);          // original index 2 b1 + b0 = lane & 3
    d += __shfl_xor(d, 4, 64);
    d += __shfl_xor(d, 8, 64);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    return d;
}

// Sum NV values over the workgroup; every thread returns with the totals.  sm: >= NV*16 floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum(v[i]);
        if (lane == 0) sm[i * 16 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += sm[i * 16 + w];
        v[i] = s;
    }
}
__device__ __forceinline__ float block_max(float v, float* sm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    const float s = wave_max(v);
    if (lane == 0) sm[wave] = s;
    __syncthreads();
    float m = sm[0];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, sm[w]);
    return m;
}

__device__ __forceinline__ float act_apply(float x, int act, float slope) {
    if (act == ACT_LRELU) return x > 0.f ? x : x * slope;
    if (act == ACT_TANH) return tanhf(x);
    return x;
}
__device__ __forceinline__ float act_deriv_from_out(float y, int act, float slope) {
    if (act == ACT_LRELU) return y > 0.f ? 1.f : slope;
    if (act == ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// (rotated filter bank: rotate_bank_kernels.hpp)

// ------------------------------------------------------------------------------------------
// Segmented row reductions:  out[seg][m][o] = sum_{n in segment} X[m][n] * V[n][o]   (V == nullptr -> 1, no = 1)
// grid (M, nseg), block 256.  Used for bias grads, per-image sums, coordinate-layer weight grads.
// ------------------------------------------------------------------------------------------
template <int NO>
static __global__ void rowdot_seg_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ V, int N,
                                  int seglen, float* __restrict__ out, int M, float* __restrict__ amax) {
    __shared__ float sm[NO * 16];
    const int m = blockIdx.x, seg = blockIdx.y;
    const int nbeg = seg * seglen;
    const int nend = min(N, nbeg + seglen);
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.f;
    float mx = 0.f;                                      // amax (optional): max |X| by the way -- the h3 bound of X's consumer
    const float* xr = X + (long)m * ldx;
    if (!V && ((ldx | nbeg) & 3) == 0 && (reinterpret_cast<size_t>(X) & 15) == 0) {
        // plain row sums of a large tensor (head / hidden gradients: up to 1 GB per call at the galaxy shape): 16-byte loads, two
        // in flight per thread -- the dword loop below streamed 2.9 TB/s (round 6).  Same per-thread summation tree for every
        // launch of a shape (deterministic), but not the dword loop's: the two forms never mix within a row.
        const float4* x4 = reinterpret_cast<const float4*>(xr + nbeg);
        const int n4 = (nend - nbeg) >> 2;
        float a0 = 0.f, a1 = 0.f;
        int i = threadIdx.x;
        for (; i + (int)blockDim.x < n4; i += 2 * blockDim.x) {
            const float4 u = x4[i], w = x4[i + blockDim.x];
            a0 += (u.x + u.y) + (u.z + u.w);
            a1 += (w.x + w.y) + (w.z + w.w);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        }
        if (i < n4) {
            const float4 u = x4[i];
            a0 += (u.x + u.y) + (u.z + u.w);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
        }
        acc[0] = a0 + a1;
        for (int n = nbeg + 4 * n4 + threadIdx.x; n < nend; n += blockDim.x) {
            acc[0] += xr[n];
            mx = fmaxf(mx, fabsf(xr[n]));
        }
    } else
    for (int n = nbeg + threadIdx.x; n < nend; n += blockDim.x) {
        const float x = xr[n];
        mx = fmaxf(mx, fabsf(x));
        if (V) {
#pragma unroll
            for (int o = 0; o < NO; ++o) acc[o] += x * V[(long)n * NO + o];
        } else {
            acc[0] += x;
        }
    }
    block_sum<NO>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int o = 0; o < NO; ++o) out[((long)seg * M + m) * NO + o] = acc[o];
    }
    if (amax) {                                          // one atomic per workgroup, and only when it brings something new
        mx = block_max(mx, sm);
        if (threadIdx.x == 0) {
            const unsigned b = __float_as_uint(mx);
            if (b > __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(amax))) atomicMax(reinterpret_cast<unsigned*>(amax), b);
        }
    }
}

// out[i] (+)= scale * sum_s in[s*L + i]
static __global__ void seg_sum_kernel(const float* __restrict__ in, int S, long L, float* __restrict__ out, float scale,
                               int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += in[(long)k * L + i];
        s *= scale;
        if (accumulate) out[i] += s; else out[i] = s;
    }
}

// The same for FEW outputs and many segments (row sums of a head-gradient / output-gradient tensor: L = 1 .. 7 outputs, S = 256
// segments): the kernel above then runs L threads that each walk S dependent loads (17-23 us for 2 KB of data); here one
// wave per output, lanes over the segments, a fixed-order butterfly at the end (deterministic).
static __global__ __launch_bounds__(64) void seg_sum_wave_kernel(const float* __restrict__ in, int S, long L, float* __restrict__ out,
                                                               float scale, int accumulate) {
    const long i = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < S; k += 64) s += in[(long)k * L + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        s *= scale;
        if (accumulate) out[i] += s; else out[i] = s;
    }
}

// ... and for MANY outputs and many segments (the per-image sums of a Fourier decoder's first-layer gradient: L = 512 features,
// S = 256 images -- seg_sum_kernel ran 512 threads that each walked 256 dependent loads: 70 us for 512 KB): 64 outputs x 16
// segment lanes per workgroup, four independent partial sums per lane, a fixed-order reduction through LDS (deterministic).
static __global__ __launch_bounds__(1024) void seg_sum_tile_kernel(const float* __restrict__ in, int S, long L,
                                                                   float* __restrict__ out, float scale, int accumulate) {
    __shared__ float sm[16][64];
    const int li = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + li;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < L) {
        int k = sl;
        for (; k + 48 < S; k += 64) {
            s0 += in[(long)k * L + i];
            s1 += in[(long)(k + 16) * L + i];
            s2 += in[(long)(k + 32) * L + i];
            s3 += in[(long)(k + 48) * L + i];
        }
        for (; k < S; k += 16) s0 += in[(long)k * L + i];
    }
    sm[sl][li] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < L) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += sm[j][li];
        s *= scale;
        if (accumulate) out[i] += s; else out[i] = s;
    }
}

// Column "dot" with a skinny matrix: out[n*NO + o] = b[o] + sum_m W[m*wsm + o*wso] * X[m*ldx + n]
// (last decoder layer n_out <= 4; coordinate gradient dx'[pix][2]).  Lanes run along n.
template <int NO>
static __global__ void coldot_kernel(const float* __restrict__ X, long ldx, int M, int N, const float* __restrict__ W,
                              int wsm, int wso, const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ float wsh[];   // [M][NO]
    for (int i = threadIdx.x; i < M * NO; i += blockDim.x) {
        const int m = i / NO, o = i - m * NO;
        wsh[i] = W[(long)m * wsm + (long)o * wso];
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = bias ? bias[o] : 0.f;
    for (int m = 0; m < M; ++m) {
        const float x = X[(long)m * ldx + n];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] += wsh[m * NO + o] * x;
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) out[(long)n * NO + o] = acc[o];
}

// D[m][n] = (sum_o W[m*wsm + o*wso] * dy[n*NO + o]) * act'(H[m][n])      (backward of the last decoder layer)
template <int NO>
static __global__ void outer_mask_kernel(const float* __restrict__ dy, const float* __restrict__ W, int wsm, int wso,
                                  const float* __restrict__ H, long ldh, float* __restrict__ D, long ldd, int M, int N,
                                  int act, float slope) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float g[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) g[o] = dy[(long)n * NO + o];
    const int mbeg = blockIdx.y * 16;
    const int mend = min(M, mbeg + 16);
    for (int m = mbeg; m < mend; ++m) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < NO; ++o) s += W[(long)m * wsm + (long)o * wso] * g[o];
        const float h = H[(long)m * ldh + n];
        D[(long)m * ldd + n] = s * act_deriv_from_out(h, act, slope);
    }
}

// dpre = dY * act'(Y) elementwise
static __global__ void act_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ dpre,
                               long n, int act, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dpre[i] = dY[i] * act_deriv_from_out(Y[i], act, slope);
}

// ------------------------------------------------------------------------------------------
// Coordinate transform (reference train_mnist.py:222,234-239):  x' = (x - dx) * [[c, s], [-s, c]]
// ------------------------------------------------------------------------------------------
static __global__ void coord_fwd_kernel(const float* __restrict__ xc, const float* __restrict__ dx,
                                 const float* __restrict__ theta, float* __restrict__ xr, int B, int Np) {
    const long total = (long)B * Np;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / Np), p = (int)(i - (long)b * Np);
        const float c = cosf(theta[b]), s = sinf(theta[b]);
        const float x0 = xc[2 * p] - dx[2 * b], x1 = xc[2 * p + 1] - dx[2 * b + 1];
        xr[2 * i] = x0 * c - x1 * s;
        xr[2 * i + 1] = x0 * s + x1 * c;
    }
}
// one workgroup per image: d_theta[b], d_dx[b][2] from gxr[b][p][2]
static __global__ void coord_bwd_kernel(const float* __restrict__ xc, const float* __restrict__ dx,
                                 const float* __restrict__ theta, const float* __restrict__ gxr,
                                 float* __restrict__ gdx, float* __restrict__ gtheta, int Np) {
    __shared__ float sm[3 * 16];
    const int b = blockIdx.x;
    const float c = cosf(theta[b]), s = sinf(theta[b]);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < Np; p += blockDim.x) {
        const float x0 = xc[2 * p] - dx[2 * b], x1 = xc[2 * p + 1] - dx[2 * b + 1];
        const float g0 = gxr[2 * ((long)b * Np + p)], g1 = gxr[2 * ((long)b * Np + p) + 1];
        acc[0] += -(g0 * c + g1 * s);
        acc[1] += g0 * s - g1 * c;
        const float r0 = x0 * c - x1 * s, r1 = x0 * s + x1 * c;
        acc[2] += -g0 * r1 + g1 * r0;
    }
    block_sum<3>(acc, sm);
    if (threadIdx.x == 0) { gdx[2 * b] = acc[0]; gdx[2 * b + 1] = acc[1]; gtheta[b] = acc[2]; }
}

// ------------------------------------------------------------------------------------------
// Decoder first layer without Fourier features (reference src/models.py:107-118, in_dim = 2):
//   h[f][pix] = act( Wc[f][0]*x0 + Wc[f][1]*x1 + bc[f] + LB[img][f] )
// ------------------------------------------------------------------------------------------
// pre-activation of the first decoder layer, with a FIXED operation order: the kernels that recompute this layer
// instead of reading its stored output (dense_x6_kernels.hpp, VirtAct) must reproduce it bit for bit so that the
// activation masks of forward and backward agree
__device__ __forceinline__ float dec_l0_pre(float w0, float w1, float bc, float lb, float x0, float x1) {
    return __fmaf_rn(w1, x1, __fmaf_rn(w0, x0, bc)) + lb;
}
static __global__ void dec_l0_fwd_kernel(const float* __restrict__ xr, const float* __restrict__ Wc,
                                  const float* __restrict__ bc, const float* __restrict__ LB, float* __restrict__ h,
                                  long ldh, int F, long Ntot, int Np, int act, float slope) {
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Ntot) return;
    const float x0 = xr[2 * n], x1 = xr[2 * n + 1];
    const int img = (int)(n / Np);
    const int fbeg = blockIdx.y * 16, fend = min(F, fbeg + 16);
    for (int f = fbeg; f < fend; ++f) {
        const float v = dec_l0_pre(Wc[2 * f], Wc[2 * f + 1], bc[f], LB ? LB[(long)img * F + f] : 0.f, x0, x1);
        h[(long)f * ldh + n] = act_apply(v, act, slope);
    }
}
// LB[img][f] = sum_d Wl[f][d] * z[img][d]     (latent_linear, no bias; models.py:111-116)
static __global__ void latent_bias_kernel(const float* __restrict__ Wl, const float* __restrict__ z, float* __restrict__ LB,
                                   int B, int F, int zd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * F) return;
    const int img = i / F, f = i - img * F;
    float s = 0.f;
    for (int d = 0; d < zd; ++d) s += Wl[f * zd + d] * z[img * zd + d];
    LB[i] = s;
}
// From S[img][f] = sum_{pix in img} dpre0[f][pix]:  dWl[f][d] = sum_img S*z,  dz[img][d] = sum_f Wl[f][d]*S
// Block = 64 outputs x 16 reduction slices (the sums run over the batch resp. the features: a thread per output walking
// all of them was a 128 us latency chain); slice sums are added in slice order.
static __global__ __launch_bounds__(1024) void latent_bwd_kernel(const float* __restrict__ S, const float* __restrict__ Wl,
                                                                 const float* __restrict__ z, float* __restrict__ dWl,
                                                                 float* __restrict__ dz, int B, int F, int zd) {
    __shared__ float part[2][16][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float s0 = 0.f, s1 = 0.f;
    if (i < F * zd) {
        const int f = i / zd, d = i - f * zd;
        for (int b = sl; b < B; b += 16) s0 += S[(long)b * F + f] * z[b * zd + d];
    }
    if (i < B * zd) {
        const int b = i / zd, d = i - b * zd;
        for (int f = sl; f < F; f += 16) s1 += Wl[f * zd + d] * S[(long)b * F + f];
    }
    part[0][sl][o] = s0;
    part[1][sl][o] = s1;
    __syncthreads();
    if (sl < 2) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[sl][q][o];
        if (sl == 0 && i < F * zd) dWl[i] = t;
        if (sl == 1 && i < B * zd) dz[i] = t;
    }
}

// Random Fourier features (reference RandomFourierEmbedding2d.forward, models.py:53-58):
//   feat[f][pix] = cos( (Wf[f][0]/sigma)*x0 + (Wf[f][1]/sigma)*x1 + bf[f] )
static __global__ void fourier_fwd_kernel(const float* __restrict__ xr, const float* __restrict__ Wf,
                                   const float* __restrict__ bf, float sigma, float* __restrict__ feat, long ld, int F,
                                   long Ntot) {
    // (the workgroup's 16 features once into LDS: the two divisions per feature were done by every thread -- round 4)
    __shared__ float wsm[16][3];
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int fbeg = blockIdx.y * 16, fend = min(F, fbeg + 16);
    if ((int)threadIdx.x < fend - fbeg) {
        const int f = fbeg + threadIdx.x;
        wsm[threadIdx.x][0] = Wf[2 * f] / sigma;
        wsm[threadIdx.x][1] = Wf[2 * f + 1] / sigma;
        wsm[threadIdx.x][2] = bf[f];
    }
    __syncthreads();
    if (n >= Ntot) return;
    const float x0 = xr[2 * n], x1 = xr[2 * n + 1];
    for (int f = fbeg; f < fend; ++f) {
        const float w0 = wsm[f - fbeg][0], w1 = wsm[f - fbeg][1];
        feat[(long)f * ld + n] = cosf(x0 * w0 + x1 * w1 + wsm[f - fbeg][2]);
    }
}
// gxr[pix][j] = sum_f -sin(arg_f) * (Wf[f][j]/sigma) * dfeat[f][pix]
static __global__ void fourier_bwd_kernel(const float* __restrict__ xr, const float* __restrict__ Wf,
                                   const float* __restrict__ bf, float sigma, const float* __restrict__ dfeat, long ld,
                                   int F, long Ntot, float* __restrict__ gxr) {
    // (features in chunks of 256 through LDS: the divisions once per workgroup instead of once per thread and feature; the sum
    //  over f keeps its order -- round 4)
    __shared__ float wsm[256][3];
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = n < Ntot;
    const float x0 = live ? xr[2 * n] : 0.f, x1 = live ? xr[2 * n + 1] : 0.f;
    float g0 = 0.f, g1 = 0.f;
    for (int f0 = 0; f0 < F; f0 += 256) {
        __syncthreads();
        if (f0 + (int)threadIdx.x < F) {
            const int f = f0 + threadIdx.x;
            wsm[threadIdx.x][0] = Wf[2 * f] / sigma;
            wsm[threadIdx.x][1] = Wf[2 * f + 1] / sigma;
            wsm[threadIdx.x][2] = bf[f];
        }
        __syncthreads();
        const int fe = min(256, F - f0);
        if (live)
            for (int q = 0; q < fe; ++q) {
                const float w0 = wsm[q][0], w1 = wsm[q][1];
                // v_sin_f32 instead of the ~40-instruction libm routine (this loop is F sines per pixel: 0.58 -> 0.2 ms of the
                // 28 x 28 Fourier step, round 5) -- with the range reduction done HERE (round 6): the instruction takes
                // revolutions and is specified for +-256 of them only, so the argument is brought to [0, 1) first, and the
                // product with 1 / 2 pi carries its rounding error along (two-term constant, one FMA for the product's own
                // rounding): the reduced phase is exact to ~1e-8 |arg| rad, below the rounding of the fp32 argument itself
                // (6e-8 |arg|) whatever sigma and the coordinates are.
                const float a_ = x0 * w0 + x1 * w1 + wsm[q][2];
                const float t_ = a_ * 0.15915494f;                                        // fp32(1 / 2 pi)
                const float e_ = __fmaf_rn(a_, 6.4206383e-09f, __fmaf_rn(a_, 0.15915494f, -t_));
                const float t = -__builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t_) + e_) * dfeat[(long)(f0 + q) * ld + n];
                g0 += t * w0;
                g1 += t * w1;
            }
    }
    if (live) {
        gxr[2 * n] = g0;
        gxr[2 * n + 1] = g1;
    }
}

// ------------------------------------------------------------------------------------------
// Likelihoods.  One workgroup per image over the FLAT per-image vectors (the reference compares
// y_hat.view(b,-1) with y.view(b,-1): train_mnist.py:288-291, train_galaxy.py:288-292,
// train_particles.py:284-296,336-338).  kind: 0 BCE-with-logits, 1 Gaussian, 2 Gaussian with learned log-variance
// (mu = yh[i], logvar = yh[L+i], i < L).
// ------------------------------------------------------------------------------------------
static __global__ void loglik_fwd_kernel(const float* __restrict__ yh, const float* __restrict__ y, float* __restrict__ lp,
                                  int L, int kind) {
    __shared__ float sm[16];
    const int b = blockIdx.x;
    const long ldy = kind == 2 ? 2L * L : L;
    const float* a = yh + (long)b * ldy;
    const float* t = y + (long)b * L;
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const float x = a[i], yy = t[i];
        if (kind == 0) acc[0] -= fmaxf(x, 0.f) - x * yy + log1pf(expf(-fabsf(x)));
        else if (kind == 1) acc[0] -= 0.5f * (x - yy) * (x - yy);
        else { const float lv = a[L + i]; acc[0] -= 0.5f * ((x - yy) * (x - yy) * expf(-lv) + lv); }
    }
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) lp[b] = acc[0];
}
static __global__ void loglik_bwd_kernel(const float* __restrict__ yh, const float* __restrict__ y,
                                  const float* __restrict__ glp, float* __restrict__ gyh, int B, int L, int kind) {
    const long total = (long)B * L;
    const long ldy = kind == 2 ? 2L * L : L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), j = (int)(i - (long)b * L);
        const float g = glp[b];
        const float x = yh[(long)b * ldy + j], yy = y[i];
        if (kind == 0) gyh[(long)b * ldy + j] = -g * (1.f / (1.f + expf(-x)) - yy);
        else if (kind == 1) gyh[(long)b * ldy + j] = -g * (x - yy);
        else {
            const float lv = yh[(long)b * ldy + L + j];
            const float e = expf(-lv), d = x - yy;
            gyh[(long)b * ldy + j] = -g * d * e;
            gyh[(long)b * ldy + L + j] = -0.5f * g * (1.f - d * d * e);
        }
    }
}


// ------------------------------------------------------------------------------------------
// Particle likelihood tail (reference train_particles.py:298-338).
// (1) per-image CTF filter: depthwise cross-correlation with an odd kc x kc kernel, zero padding kc/2
//     (F.conv2d(y_mu.view(1,B,n,n), ctf, padding=pad, groups=B), :298-302).  flip = 1 applies the 180-degree rotated
//     kernel = the gradient w.r.t. the input.  One thread per output pixel; the filter row is wave-uniform.
// (2) circular mask centred at the inferred translation (:309-333): pixel (i,j) is kept iff
//     (dx0/s - gx_j)^2 + (dx1/s - gy_i)^2 < radius^2 with gx_j = -ceil(n/2) + j, gy_i = floor(n/2) - i; masked pixels
//     contribute nothing to the Gaussian log-likelihood and get no gradient (the mask itself carries no gradient).
// ------------------------------------------------------------------------------------------
static __global__ void ctf_corr_kernel(const float* __restrict__ in, const float* __restrict__ ctf, float* __restrict__ out,
                                int n, int kc, int flip) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= n * n) return;
    const int i = pix / n, j = pix - i * n;
    const int p = kc / 2;
    const float* src = in + (long)b * n * n;
    const float* w = ctf + (long)b * kc * kc;
    float acc = 0.f;
    for (int a = 0; a < kc; ++a) {
        const int ii = i + a - p;
        if (ii < 0 || ii >= n) continue;
        const float* wr = flip ? (w + (long)(kc - 1 - a) * kc) : (w + (long)a * kc);
        const float* sr = src + (long)ii * n;
        for (int c = 0; c < kc; ++c) {
            const int jj = j + c - p;
            if (jj >= 0 && jj < n) acc += sr[jj] * (flip ? wr[kc - 1 - c] : wr[c]);
        }
    }
    out[(long)b * n * n + pix] = acc;
}

__device__ __forceinline__ bool mask_keep(int i, int j, int n, float cx, float cy, float r2) {
    const float gx = (float)(j - (n + 1) / 2);
    const float gy = (float)(n / 2 - i);
    const float ddx = cx - gx, ddy = cy - gy;
    return ddx * ddx + ddy * ddy < r2;
}
// Gaussian log-likelihood with optional circular mask: lp[b] = -0.5 * sum_{kept pixels} (yh - y)^2
static __global__ void loglik_masked_fwd_kernel(const float* __restrict__ yh, const float* __restrict__ y,
                                         const float* __restrict__ dx, float inv_spacing, float radius, int n,
                                         float* __restrict__ lp) {
    __shared__ float sm[16];
    const int b = blockIdx.x;
    const int L = n * n;
    const float cx = dx[2 * b] * inv_spacing, cy = dx[2 * b + 1] * inv_spacing, r2 = radius * radius;
    float acc[1] = {0.f};
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
        const int i = t / n, j = t - i * n;
        if (mask_keep(i, j, n, cx, cy, r2)) {
            const float d = yh[(long)b * L + t] - y[(long)b * L + t];
            acc[0] -= 0.5f * d * d;
        }
    }
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) lp[b] = acc[0];
}
static __global__ void loglik_masked_bwd_kernel(const float* __restrict__ yh, const float* __restrict__ y,
                                         const float* __restrict__ dx, float inv_spacing, float radius, int n,
                                         const float* __restrict__ glp, float* __restrict__ gyh, int B) {
    const long L = (long)n * n, total = (long)B * L;
    const float r2 = radius * radius;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / L);
        const int r = (int)(t - (long)b * L);
        const int i = r / n, j = r - i * n;
        const bool keep = mask_keep(i, j, n, dx[2 * b] * inv_spacing, dx[2 * b + 1] * inv_spacing, r2);
        gyh[t] = keep ? -glp[b] * (yh[t] - y[t]) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// Attention head (reference models.py:358-401 + train_mnist.py:192-282): one workgroup per image.
// heads[ch][img*RP + j], ch: 0 logit, 1 theta_mu, 2 theta_logstd, 3..3+zd-1 z_mu, 3+zd.. z_logstd.
// ------------------------------------------------------------------------------------------
struct HeadParams {
    const float* heads; long ldh;
    const float* E;        // [B][RP]  Exp(1) draws of the Gumbel-softmax
    const float* eps_z;    // [B][zd]
    const float* eps_t;    // [B]
    const float* p_r;      // [R]   log prior over rotations
    const float* off;      // [R]   rotation offsets (zeros without refinement)
    const float* p_tr;     // [RP]  log-softmax of p_t + p_r (joint prior, float64 on host -> f32)
    const float* grid;     // [P][2] translation grid
    int R, P, zd;
    float sigma_p;         // pi / R  (train_mnist.py:269-272)
    float theta_off_scale; // 1 if offsets are added to theta_mu (rot_refinement), else 0
};

// ---- phases of the head over a range [j0, j1) of one image's R*P positions: shared by the one-workgroup-per-image
// kernels (small R*P: 256 images x 8 712 positions at cfg4) and the chunked ones (cfg5: 8 images x 266 256 positions) ----
__device__ __forceinline__ void head_p1(const HeadParams& hp, long base, int j0, int j1, float* __restrict__ attn,
                                        float& m1, float& m2) {
    const float* logit = hp.heads + base;
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const float l = logit[j] + hp.p_r[j / hp.P];
        attn[base + j] = l;
        m1 = fmaxf(m1, l);
        m2 = fmaxf(m2, l - logf(hp.E[base + j]));
    }
}
__device__ __forceinline__ void head_p2(const HeadParams& hp, long base, int j0, int j1, const float* __restrict__ attn,
                                        float m1, float m2, float (&s)[2]) {
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const float l = attn[base + j];
        s[0] += expf(l - m1);
        s[1] += expf(l - logf(hp.E[base + j]) - m2);
    }
}
// q, a and the translation pooling + val1 sums
__device__ __forceinline__ void head_p3(const HeadParams& hp, long base, int j0, int j1, const float* __restrict__ attn,
                                        float lse, float m2, float inv2, float* __restrict__ q, float* __restrict__ a,
                                        float (&t)[3]) {
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const float l = attn[base + j];
        const float qq = l - lse;
        const float aa = expf(l - logf(hp.E[base + j]) - m2) * inv2;
        q[base + j] = qq;
        a[base + j] = aa;
        const int hw = j % hp.P;
        t[0] += aa * hp.grid[2 * hw];
        t[1] += aa * hp.grid[2 * hw + 1];
        t[2] += expf(qq) * (qq - hp.p_tr[j]);
    }
}
// theta (c = -1) or latent dim c: pooled mean / std and E_q[KL] sums
__device__ __forceinline__ void head_p4(const HeadParams& hp, long base, int j0, int j1, int c, const float* __restrict__ q,
                                        const float* __restrict__ a, float (&u)[3]) {
    const float* mu_p = hp.heads + (long)(c < 0 ? 1 : 3 + c) * hp.ldh + base;
    const float* ls_p = hp.heads + (long)(c < 0 ? 2 : 3 + hp.zd + c) * hp.ldh + base;
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const float aa = a[base + j];
        const float eq = expf(q[base + j]);
        float mu = mu_p[j];
        float sd = expf(ls_p[j]) + EPS_STD;
        float klv;
        if (c < 0) {
            const float o = hp.off[j / hp.P];
            mu += hp.theta_off_scale * o;
            u[0] += aa * mu;
            u[1] += aa * sd;
            if (eq == 0.f) { mu = 0.f; sd = 1.f; }
            const float vr = (sd / hp.sigma_p) * (sd / hp.sigma_p);
            const float t1 = ((mu - o) / hp.sigma_p) * ((mu - o) / hp.sigma_p);
            klv = 0.5f * (vr + t1 - 1.f - logf(vr));
        } else {
            u[0] += aa * mu;
            u[1] += aa * sd;
            if (eq == 0.f) { mu = 0.f; sd = 1.f; }
            klv = 0.5f * (sd * sd + mu * mu - 1.f - logf(sd * sd));
        }
        u[2] += eq * klv;
    }
}

// The same for HEAD_CG latent dimensions c0 .. c0 + HEAD_CG - 1 at once (round 6): with z_dim = 50 (galaxy) the per-dimension
// passes above were 51 dependent load -> reduce rounds per chunk, each re-reading a and q and re-evaluating exp(q) (0.58 ms for
// 0.88 GB of head rows); a group shares those per position and has 2 HEAD_CG independent loads in flight.  Dimensions beyond zd
// are clamped to the last one (their sums are discarded).  Per dimension the sums run over j in the same order as head_p4:
// bitwise the same results.
// (measured at the galaxy shape, chunked kernels: forward 584 -> 454 us, backward 421 + 579 -> 264 + 491 us)
template <int HEAD_CG>
__device__ __forceinline__ void head_p4g(const HeadParams& hp, long base, int j0, int j1, int c0, const float* __restrict__ q,
                                         const float* __restrict__ a, float (&u)[3 * HEAD_CG]) {
    const float* mu_p[HEAD_CG];
    const float* ls_p[HEAD_CG];
#pragma unroll
    for (int k = 0; k < HEAD_CG; ++k) {
        const int c = min(c0 + k, hp.zd - 1);
        mu_p[k] = hp.heads + (long)(3 + c) * hp.ldh + base;
        ls_p[k] = hp.heads + (long)(3 + hp.zd + c) * hp.ldh + base;
    }
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const float aa = a[base + j];
        const float eq = expf(q[base + j]);
        float mu[HEAD_CG], ls[HEAD_CG];
#pragma unroll
        for (int k = 0; k < HEAD_CG; ++k) { mu[k] = mu_p[k][j]; ls[k] = ls_p[k][j]; }
#pragma unroll
        for (int k = 0; k < HEAD_CG; ++k) {
            float m = mu[k];
            float sd = expf(ls[k]) + EPS_STD;
            u[3 * k] += aa * m;
            u[3 * k + 1] += aa * sd;
            if (eq == 0.f) { m = 0.f; sd = 1.f; }
            u[3 * k + 2] += eq * (0.5f * (sd * sd + m * m - 1.f - logf(sd * sd)));
        }
    }
}

static __global__ void attn_head_fwd_kernel(HeadParams hp, float* __restrict__ attn, float* __restrict__ q,
                                     float* __restrict__ a, float* __restrict__ zs, float* __restrict__ th,
                                     float* __restrict__ dxo, float* __restrict__ kl) {
    constexpr int HEAD_CG = 4;
    __shared__ float sm[3 * HEAD_CG * 16];
    const int b = blockIdx.x;
    const int RP = hp.R * hp.P;
    const long base = (long)b * RP;
    float m1 = -INFINITY, m2 = -INFINITY;
    head_p1(hp, base, 0, RP, attn, m1, m2);
    m1 = block_max(m1, sm);
    m2 = block_max(m2, sm);
    float s[2] = {0.f, 0.f};
    head_p2(hp, base, 0, RP, attn, m1, m2, s);
    block_sum<2>(s, sm);
    const float lse = m1 + logf(s[0]);
    const float inv2 = 1.f / s[1];
    float t[3] = {0.f, 0.f, 0.f};
    head_p3(hp, base, 0, RP, attn, lse, m2, inv2, q, a, t);
    block_sum<3>(t, sm);
    float klsum = t[2];
    if (threadIdx.x == 0) { dxo[2 * b] = t[0]; dxo[2 * b + 1] = t[1]; }
    {
        float u[3] = {0.f, 0.f, 0.f};
        head_p4(hp, base, 0, RP, -1, q, a, u);
        block_sum<3>(u, sm);
        klsum += u[2];
        if (threadIdx.x == 0) th[b] = u[1] * hp.eps_t[b] + u[0];
    }
    for (int c0 = 0; c0 < hp.zd; c0 += HEAD_CG) {
        float u[3 * HEAD_CG];
#pragma unroll
        for (int k = 0; k < 3 * HEAD_CG; ++k) u[k] = 0.f;
        head_p4g<HEAD_CG>(hp, base, 0, RP, c0, q, a, u);
        block_sum<3 * HEAD_CG>(u, sm);
#pragma unroll
        for (int k = 0; k < HEAD_CG; ++k) {
            const int c = c0 + k;
            if (c < hp.zd) {                             // (dimension by dimension, in order: the same sum as before)
                klsum += u[3 * k + 2];
                if (threadIdx.x == 0) zs[b * hp.zd + c] = u[3 * k + 1] * hp.eps_z[b * hp.zd + c] + u[3 * k];
            }
        }
    }
    if (threadIdx.x == 0) kl[b] = klsum;
}

// ---- chunked forward: G workgroups per image, three launches.  part layout per (image, chunk):
//   [0..3]  = (m1, s0 relative to m1, m2, s1 relative to m2)      written by _a
//   [4..]   = t[3], then u[3] for c = -1 .. zd-1                   written by _b
// The chunk results are combined in chunk order (deterministic); a softmax over chunks is rescaled to the common maximum.
constexpr int HEAD_PART_A = 4;
__host__ __device__ inline int head_part_floats(int zd) { return HEAD_PART_A + 3 + 3 * (zd + 1); }

static __global__ void attn_head_fwd_a_kernel(HeadParams hp, int G, int chunk, float* __restrict__ attn,
                                              float* __restrict__ part) {
    __shared__ float sm[4 * 16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int RP = hp.R * hp.P;
    const long base = (long)b * RP;
    const int j0 = g * chunk, j1 = min(RP, j0 + chunk);
    float m1 = -INFINITY, m2 = -INFINITY;
    head_p1(hp, base, j0, j1, attn, m1, m2);
    m1 = block_max(m1, sm);
    m2 = block_max(m2, sm);
    float s[2] = {0.f, 0.f};
    head_p2(hp, base, j0, j1, attn, m1, m2, s);
    block_sum<2>(s, sm);
    if (threadIdx.x == 0) {
        float* p = part + (long)blockIdx.x * head_part_floats(hp.zd);
        p[0] = m1; p[1] = s[0]; p[2] = m2; p[3] = s[1];
    }
}
// combined (max, sum) over the G chunks of image b
__device__ __forceinline__ void head_combine(const float* __restrict__ part, int b, int G, int pf, float& m1, float& s0,
                                             float& m2, float& s1) {
    m1 = -INFINITY; m2 = -INFINITY;
    for (int g = 0; g < G; ++g) {
        const float* p = part + ((long)b * G + g) * pf;
        m1 = fmaxf(m1, p[0]);
        m2 = fmaxf(m2, p[2]);
    }
    s0 = 0.f; s1 = 0.f;
    for (int g = 0; g < G; ++g) {
        const float* p = part + ((long)b * G + g) * pf;
        s0 += p[1] * expf(p[0] - m1);
        s1 += p[3] * expf(p[2] - m2);
    }
}
static __global__ void attn_head_fwd_b_kernel(HeadParams hp, int G, int chunk, const float* __restrict__ attn,
                                              float* __restrict__ q, float* __restrict__ a, float* __restrict__ part) {
    constexpr int HEAD_CG = 4;          // (8 measured 624 us against 454: 1 024-thread workgroups leave a wave 128 registers)
    __shared__ float sm[3 * HEAD_CG * 16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int RP = hp.R * hp.P;
    const long base = (long)b * RP;
    const int j0 = g * chunk, j1 = min(RP, j0 + chunk);
    const int pf = head_part_floats(hp.zd);
    float m1, s0, m2, s1;
    head_combine(part, b, G, pf, m1, s0, m2, s1);
    const float lse = m1 + logf(s0);
    const float inv2 = 1.f / s1;
    float* p = part + (long)blockIdx.x * pf + HEAD_PART_A;
    float t[3] = {0.f, 0.f, 0.f};
    head_p3(hp, base, j0, j1, attn, lse, m2, inv2, q, a, t);
    block_sum<3>(t, sm);
    if (threadIdx.x == 0) { p[0] = t[0]; p[1] = t[1]; p[2] = t[2]; }
    {
        float u[3] = {0.f, 0.f, 0.f};
        head_p4(hp, base, j0, j1, -1, q, a, u);
        block_sum<3>(u, sm);
        if (threadIdx.x == 0) { p[3] = u[0]; p[4] = u[1]; p[5] = u[2]; }
    }
    for (int c0 = 0; c0 < hp.zd; c0 += HEAD_CG) {
        float u[3 * HEAD_CG];
#pragma unroll
        for (int k = 0; k < 3 * HEAD_CG; ++k) u[k] = 0.f;
        head_p4g<HEAD_CG>(hp, base, j0, j1, c0, q, a, u);
        block_sum<3 * HEAD_CG>(u, sm);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < HEAD_CG; ++k) {
                const int c = c0 + k;
                if (c < hp.zd) { p[3 + 3 * (c + 1)] = u[3 * k]; p[4 + 3 * (c + 1)] = u[3 * k + 1]; p[5 + 3 * (c + 1)] = u[3 * k + 2]; }
            }
        }
    }
}
// one thread per (image, output): sums the chunk partials in chunk order
static __global__ void attn_head_fwd_c_kernel(HeadParams hp, int B, int G, const float* __restrict__ part,
                                              float* __restrict__ zs, float* __restrict__ th, float* __restrict__ dxo,
                                              float* __restrict__ kl) {
    const int pf = head_part_floats(hp.zd);
    const int nout = 3 + 3 * (hp.zd + 1);
    __shared__ float tot[3 + 3 * 65];
    const int b = blockIdx.x;
    for (int o = threadIdx.x; o < nout; o += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += part[((long)b * G + g) * pf + HEAD_PART_A + o];
        tot[o] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dxo[2 * b] = tot[0];
        dxo[2 * b + 1] = tot[1];
        float klsum = tot[2];
        for (int c = -1; c < hp.zd; ++c) {
            const float u0 = tot[3 + 3 * (c + 1)], u1 = tot[4 + 3 * (c + 1)];
            klsum += tot[5 + 3 * (c + 1)];
            if (c < 0) th[b] = u1 * hp.eps_t[b] + u0;
            else zs[b * hp.zd + c] = u1 * hp.eps_z[b * hp.zd + c] + u0;
        }
        kl[b] = klsum;
    }
}

// Backward of the head.  Upstream: gz[B][zd], gth[B], gdx[B][2], gkl[B] and (optional, may be null)
// g_attn, g_q, g_a [B][RP] for the module-level 7-tuple API.  Output dheads[ch][img*RP + j].
// One pass over [j0, j1): pass 0 accumulates acc = (sum a*da, sum dq); pass 1 writes dheads given (sa, sq).
// HEAD_BD: latent dimensions per group of loads (2 in the one-workgroup-per-image kernel: z_dim is typically 2 there; 8 chunked)
template <int HEAD_BD>
__device__ __forceinline__ void head_bwd_pass(const HeadParams& hp, int b, long base, int j0, int j1, int pass, float sa,
                                              float sq, const float* __restrict__ q, const float* __restrict__ a,
                                              const float* __restrict__ gz, const float* __restrict__ gth,
                                              const float* __restrict__ gdx, const float* __restrict__ gkl,
                                              const float* __restrict__ g_attn, const float* __restrict__ g_q,
                                              const float* __restrict__ g_a, float* __restrict__ dheads, float (&acc)[2]) {
    const float w = gkl[b];
    const float gt = gth[b];
    const float et = hp.eps_t[b];
    const float gd0 = gdx[2 * b], gd1 = gdx[2 * b + 1];
    const float isp2 = 1.f / (hp.sigma_p * hp.sigma_p);
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const float aa = a[base + j];
        const float qq = q[base + j];
        const float eq = expf(qq);
        const bool dead = (eq == 0.f);
        const int r = j / hp.P, hw = j - r * hp.P;
        const float o = hp.off[r];
        // theta
        const float tmu_raw = hp.heads[1 * hp.ldh + base + j];
        const float tls = hp.heads[2 * hp.ldh + base + j];
        const float tmu = tmu_raw + hp.theta_off_scale * o;
        const float tex = expf(tls);
        const float tsd = tex + EPS_STD;
        float da = gt * (tmu + et * tsd) + gd0 * hp.grid[2 * hw] + gd1 * hp.grid[2 * hw + 1];
        float klsum;
        {
            const float mu = dead ? 0.f : tmu, sd = dead ? 1.f : tsd;
            const float vr = (sd / hp.sigma_p) * (sd / hp.sigma_p);
            const float t1 = ((mu - o) / hp.sigma_p) * ((mu - o) / hp.sigma_p);
            klsum = 0.5f * (vr + t1 - 1.f - logf(vr));
        }
        if (pass == 1) {
            float dmu = aa * gt, dsd = aa * gt * et;
            if (!dead) { dmu += w * eq * (tmu - o) * isp2; dsd += w * eq * (tsd * isp2 - 1.f / tsd); }
            dheads[1 * hp.ldh + base + j] = dmu;
            dheads[2 * hp.ldh + base + j] = dsd * tex;
        }
        // latent dimensions HEAD_BD at a time, all their loads before the first store (round 6): `dheads` and `hp.heads` may alias
        // as far as the compiler knows, so in the one-by-one loop every dimension's loads waited for the previous one's stores --
        // 50 dependent round trips per position at the galaxy's z_dim (0.42 + 0.58 ms for the two passes).  Same order of the
        // sums over d: bitwise the same results.
        for (int d0 = 0; d0 < hp.zd; d0 += HEAD_BD) {
            float zmu4[HEAD_BD], zls4[HEAD_BD];
#pragma unroll
            for (int k = 0; k < HEAD_BD; ++k) {
                const int d = min(d0 + k, hp.zd - 1);
                zmu4[k] = hp.heads[(long)(3 + d) * hp.ldh + base + j];
                zls4[k] = hp.heads[(long)(3 + hp.zd + d) * hp.ldh + base + j];
            }
#pragma unroll
            for (int k = 0; k < HEAD_BD; ++k) {
                const int d = d0 + k;
                if (d < hp.zd) {
                    const float zmu = zmu4[k];
                    const float zex = expf(zls4[k]);
                    const float zsd = zex + EPS_STD;
                    const float g = gz[b * hp.zd + d], ez = hp.eps_z[b * hp.zd + d];
                    da += g * (zmu + ez * zsd);
                    const float mu = dead ? 0.f : zmu, sd = dead ? 1.f : zsd;
                    klsum += 0.5f * (sd * sd + mu * mu - 1.f - logf(sd * sd));
                    if (pass == 1) {
                        float dmu = aa * g, dsd = aa * g * ez;
                        if (!dead) { dmu += w * eq * zmu; dsd += w * eq * (zsd - 1.f / zsd); }
                        dheads[(long)(3 + d) * hp.ldh + base + j] = dmu;
                        dheads[(long)(3 + hp.zd + d) * hp.ldh + base + j] = dsd * zex;
                    }
                }
            }
        }
        if (g_a) da += g_a[base + j];
        float dq = w * eq * (qq - hp.p_tr[j] + 1.f + klsum);
        if (g_q) dq += g_q[base + j];
        if (pass == 0) {
            acc[0] += aa * da;
            acc[1] += dq;
        } else {
            float dl = aa * (da - sa) + dq - eq * sq;
            if (g_attn) dl += g_attn[base + j];
            dheads[base + j] = dl;
        }
    }
}

static __global__ void attn_head_bwd_kernel(HeadParams hp, const float* __restrict__ q, const float* __restrict__ a,
                                     const float* __restrict__ gz, const float* __restrict__ gth,
                                     const float* __restrict__ gdx, const float* __restrict__ gkl,
                                     const float* __restrict__ g_attn, const float* __restrict__ g_q,
                                     const float* __restrict__ g_a, float* __restrict__ dheads) {
    __shared__ float sm[2 * 16];
    const int b = blockIdx.x;
    const int RP = hp.R * hp.P;
    const long base = (long)b * RP;
    float acc[2] = {0.f, 0.f};
    head_bwd_pass<2>(hp, b, base, 0, RP, 0, 0.f, 0.f, q, a, gz, gth, gdx, gkl, g_attn, g_q, g_a, dheads, acc);
    block_sum<2>(acc, sm);
    float dummy[2] = {0.f, 0.f};
    head_bwd_pass<2>(hp, b, base, 0, RP, 1, acc[0], acc[1], q, a, gz, gth, gdx, gkl, g_attn, g_q, g_a, dheads, dummy);
}
// chunked backward: _a writes the (sum a*da, sum dq) partial of its chunk, _b sums the G partials and writes dheads
static __global__ void attn_head_bwd_a_kernel(HeadParams hp, int G, int chunk, const float* __restrict__ q,
                                              const float* __restrict__ a, const float* __restrict__ gz,
                                              const float* __restrict__ gth, const float* __restrict__ gdx,
                                              const float* __restrict__ gkl, const float* __restrict__ g_q,
                                              const float* __restrict__ g_a, float* __restrict__ part) {
    __shared__ float sm[2 * 16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int RP = hp.R * hp.P;
    const int j0 = g * chunk, j1 = min(RP, j0 + chunk);
    float acc[2] = {0.f, 0.f};
    head_bwd_pass<8>(hp, b, (long)b * RP, j0, j1, 0, 0.f, 0.f, q, a, gz, gth, gdx, gkl, nullptr, g_q, g_a, nullptr, acc);
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = acc[0]; part[2 * blockIdx.x + 1] = acc[1]; }
}
static __global__ void attn_head_bwd_b_kernel(HeadParams hp, int G, int chunk, const float* __restrict__ q,
                                              const float* __restrict__ a, const float* __restrict__ gz,
                                              const float* __restrict__ gth, const float* __restrict__ gdx,
                                              const float* __restrict__ gkl, const float* __restrict__ g_attn,
                                              const float* __restrict__ g_q, const float* __restrict__ g_a,
                                              const float* __restrict__ part, float* __restrict__ dheads) {
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int RP = hp.R * hp.P;
    const int j0 = g * chunk, j1 = min(RP, j0 + chunk);
    float sa = 0.f, sq = 0.f;
    for (int gg = 0; gg < G; ++gg) { sa += part[2 * (b * G + gg)]; sq += part[2 * (b * G + gg) + 1]; }
    float dummy[2] = {0.f, 0.f};
    head_bwd_pass<8>(hp, b, (long)b * RP, j0, j1, 1, sa, sq, q, a, gz, gth, gdx, gkl, g_attn, g_q, g_a, dheads, dummy);
}


// ------------------------------------------------------------------------------------------
// Rotation pooling of the translation-attention encoder (reference src/models.py:301-304: fc_r = nn.Linear(R, 1) applied
// over the rotation axis of act(conv1(x))):  X[c][b*P + p] = fb + sum_r fw[r] * A1[c][(b*R + r)*P + p].
// Backward: dA1 = fw[r] * dX * act'(A1)  (the activation sits between conv1 and the pooling), and per-block partial sums
// part[block][0..R) = sum A1 * dX  (= d fw),  part[block][R] = sum dX  (= d fb), reduced by seg_sum_kernel.
// ------------------------------------------------------------------------------------------
static __global__ void rot_pool_fwd_kernel(const float* __restrict__ A1, const float* __restrict__ fw,
                                           const float* __restrict__ fb, float* __restrict__ X, int C, int B, int R, int P) {
    const long total = (long)C * B * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const long cb = i / P;                           // c*B + b
        const float* a = A1 + cb * R * P + p;
        float s = fb[0];
        for (int r = 0; r < R; ++r) s += fw[r] * a[(long)r * P];
        X[i] = s;
    }
}
static __global__ __launch_bounds__(256) void rot_pool_bwd_kernel(const float* __restrict__ A1, const float* __restrict__ dX,
                                                                  const float* __restrict__ fw, float* __restrict__ dA1,
                                                                  float* __restrict__ part, int C, int B, int R, int P,
                                                                  int act, float slope) {
    __shared__ float sm[17 * 16];
    const long total = (long)C * B * P;
    float acc[17];
#pragma unroll
    for (int r = 0; r < 17; ++r) acc[r] = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const long cb = i / P;
        const float g = dX[i];
        acc[16] += g;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < R) {
                const long j = (cb * R + r) * P + p;
                const float a = A1[j];
                acc[r] += a * g;
                const float d = act == ACT_LRELU ? (a > 0.f ? 1.f : slope) : (act == ACT_TANH ? 1.f - a * a : 1.f);
                dA1[j] = fw[r] * g * d;
            }
        }
    }
    block_sum<17>(acc, sm);
    if (threadIdx.x == 0) {
        for (int r = 0; r < R; ++r) part[(long)blockIdx.x * (R + 1) + r] = acc[r];
        part[(long)blockIdx.x * (R + 1) + R] = acc[16];
    }
}

// ------------------------------------------------------------------------------------------
// Inference epilogue get_latent (reference clustering_mnist.py:123-161): per image, the most probable (r,h,w) under
// attn = logit + log p(r); content vector (z_mu, exp(z_logstd)) and theta_mu gathered there; translation = expected
// grid position under softmax(attn) summed over rotations.  One workgroup per image; first index wins ties.
// ------------------------------------------------------------------------------------------
static __global__ void get_latent_kernel(HeadParams hp, float* __restrict__ zc, float* __restrict__ th,
                                  float* __restrict__ dxo) {
    __shared__ float smv[16];
    __shared__ int smi[16];
    __shared__ float sm[3 * 16];
    const int b = blockIdx.x;
    const int RP = hp.R * hp.P;
    const long base = (long)b * RP;
    const float* logit = hp.heads + base;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int j = threadIdx.x; j < RP; j += blockDim.x) {
        const float l = logit[j] + hp.p_r[j / hp.P];
        if (l > best || (l == best && j < bidx)) { best = l; bidx = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(best, o, 64);
        const int oi = __shfl_down(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { smv[wave] = best; smi[wave] = bidx; }
    __syncthreads();
    best = smv[0]; bidx = smi[0];
    for (int w = 1; w < nw; ++w)
        if (smv[w] > best || (smv[w] == best && smi[w] < bidx)) { best = smv[w]; bidx = smi[w]; }
    float t[3] = {0.f, 0.f, 0.f};
    for (int j = threadIdx.x; j < RP; j += blockDim.x) {
        const float e = expf(logit[j] + hp.p_r[j / hp.P] - best);
        const int hw = j % hp.P;
        t[0] += e;
        t[1] += e * hp.grid[2 * hw];
        t[2] += e * hp.grid[2 * hw + 1];
    }
    block_sum<3>(t, sm);
    if (threadIdx.x == 0) {
        dxo[2 * b] = t[1] / t[0];
        dxo[2 * b + 1] = t[2] / t[0];
        th[b] = hp.heads[1 * hp.ldh + base + bidx] + hp.theta_off_scale * hp.off[bidx / hp.P];
    }
    for (int d = threadIdx.x; d < hp.zd; d += blockDim.x) {
        zc[b * 2 * hp.zd + d] = hp.heads[(long)(3 + d) * hp.ldh + base + bidx];
        zc[b * 2 * hp.zd + hp.zd + d] = expf(hp.heads[(long)(3 + hp.zd + d) * hp.ldh + base + bidx]);
    }
}

// ------------------------------------------------------------------------------------------
// Fused Adam over the flat parameter buffer (torch.optim.Adam defaults, reference train_mnist.py:579).
// grad_scale folds the data-parallel 1/world averaging into the update.
// ------------------------------------------------------------------------------------------
static __global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                                 float bc2_sqrt, float grad_scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i] * grad_scale;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mm / denom);
    }
}

// ------------------------------------------------------------------------------------------
// ELBO scalars of a minibatch in ONE launch (reference train_mnist.py:282,291-292: log_p = mean_b lp (float32), kl_div =
// mean_b kl (float64), elbo = log_p - kl_div) and their backward in one more -- the ATen form is a dozen 5-us launches
// (two means, a cast, a subtraction, and the chain rule of each).  One workgroup; fixed summation order.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void elbo_reduce_kernel(const float* __restrict__ lp, const float* __restrict__ kl, int B,
                                                                double* __restrict__ elbo, float* __restrict__ logp,
                                                                double* __restrict__ kld) {
    __shared__ double sm[2 * 4];
    double a = 0.0, k = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        a += (double)lp[b];
        k += (double)kl[b];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        k += __shfl_xor(k, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sm[2 * (threadIdx.x >> 6)] = a;
        sm[2 * (threadIdx.x >> 6) + 1] = k;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sa = (sm[0] + sm[2]) + (sm[4] + sm[6]), sk = (sm[1] + sm[3]) + (sm[5] + sm[7]);
        const float lpm = (float)(sa / (double)B);       // the reference's log-likelihood term is a float32 scalar
        const double klm = sk / (double)B;
        logp[0] = lpm;
        kld[0] = klm;
        elbo[0] = (double)lpm - klm;
    }
}
// g_lp[b] = (g_elbo + g_logp) / B,  g_kl[b] = (g_kld - g_elbo) / B   (NULL upstream gradients count as zero)
static __global__ void elbo_reduce_bwd_kernel(const double* __restrict__ g_elbo, const float* __restrict__ g_logp,
                                              const double* __restrict__ g_kld, int B, float* __restrict__ g_lp,
                                              float* __restrict__ g_kl) {
    const double ge = g_elbo ? g_elbo[0] : 0.0, gl = g_logp ? (double)g_logp[0] : 0.0, gk = g_kld ? g_kld[0] : 0.0;
    const float a = (float)((ge + gl) / (double)B), k = (float)((gk - ge) / (double)B);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        g_lp[b] = a;
        g_kl[b] = k;
    }
}

}  // namespace tvae
