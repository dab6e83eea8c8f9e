// Rotated filter bank of the P_n lifting convolution (reference GroupConv.trans_filter, src/models.py:174-197) and its
// transposed operator.  Test infrastructure counterpart: oracle/ (rotated_bank).  Only abi_small.hip includes this header.
#pragma once
#include <hip/hip_runtime.h>

namespace tvae {

// ------------------------------------------------------------------------------------------
// Rotated filter bank (reference GroupConv.trans_filter, src/models.py:174-197).
// The R fixed rotations are a constant sparse interpolation operator: 4 taps per output pixel.
// bank[(c*R + r)][ci*k2 + d] = sum_t w[r][d][t] * weight[(c*Cin + ci)*k2 + idx[r][d][t]]
// ------------------------------------------------------------------------------------------
// Round 6: one thread per (rotation, destination tap) and a run of RB_CH (channel, input channel) pairs -- the thread's table entry
// (one int4 + one float4) is loaded ONCE and the gathers of four pairs are in flight together.  (The element-per-thread form
// spent its time in four runtime integer divisions and two dependent load levels per element: 33 us for 1 M elements at the
// 64 x 64 shape, 172 us at the galaxy shape.)  Same four-term sum in the same order: bitwise the old result.
// A workgroup is an 8 x 32 PATCH of destination taps of one rotation (a wave: 2 rows x 32 taps = two whole 128-byte lines of the
// bank per store): the rotated patch is a compact piece of the source filter, where a row of 256 destination taps crossed it
// from edge to edge.
constexpr int RB_CH = 16;
static __global__ __launch_bounds__(256) void rotate_bank_fwd_kernel(const float* __restrict__ weight, const int* __restrict__ tap_idx,
                                                                     const float* __restrict__ tap_w, float* __restrict__ bank,
                                                                     int C, int Cin, int ksz, int R) {
    const int k2 = ksz * ksz;
    const int npx = (ksz + 31) >> 5, npy = (ksz + 7) >> 3;
    const int r = (int)blockIdx.x / (npx * npy), pp = (int)blockIdx.x - r * npx * npy;
    const int dy = 8 * (pp / npx) + ((int)threadIdx.x >> 5), dx = 32 * (pp % npx) + ((int)threadIdx.x & 31);
    if (dy >= ksz || dx >= ksz) return;
    const unsigned d = (unsigned)(dy * ksz + dx), rd = (unsigned)r * k2 + d;
    const int4 id = reinterpret_cast<const int4*>(tap_idx)[rd];
    const float4 tw = reinterpret_cast<const float4*>(tap_w)[rd];
    const unsigned npair = (unsigned)C * Cin;
    const unsigned p0 = blockIdx.y * RB_CH, p1 = min(p0 + RB_CH, npair);
    // (an absent tap loads index 0 -- in bounds -- and is left out of the sum, as before)
    const unsigned ix = id.x >= 0 ? id.x : 0, iy = id.y >= 0 ? id.y : 0, iz = id.z >= 0 ? id.z : 0, iw = id.w >= 0 ? id.w : 0;
    for (unsigned p = p0; p < p1; p += 4) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* wsrc = weight + (size_t)min(p + u, p1 - 1) * k2;
            v[u][0] = wsrc[ix]; v[u][1] = wsrc[iy]; v[u][2] = wsrc[iz]; v[u][3] = wsrc[iw];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s_ = 0.f;
            if (id.x >= 0) s_ += tw.x * v[u][0];
            if (id.y >= 0) s_ += tw.y * v[u][1];
            if (id.z >= 0) s_ += tw.z * v[u][2];
            if (id.w >= 0) s_ += tw.w * v[u][3];
            const unsigned c = (p + u) / (unsigned)Cin, ci = (p + u) - c * (unsigned)Cin;
            if (p + u < p1) bank[((size_t)(c * R + r) * Cin + ci) * k2 + d] = s_;
        }
    }
}

// Transposed operator in gather (CSR) form: deterministic, no atomics.
// dweight[(c*Cin+ci)*k2 + s] = sum_{e in [ptr[s],ptr[s+1])} w[e] * dbank[(c*R + r[e])][ci*k2 + dst[e]]
// Round 6: the entry lists are the same for every (c, ci) pair, and lane s reading ITS list from global memory touches a line of
// its own per load (the lists of neighbouring s are ~4 R entries apart): 1.2 GB of index traffic at the galaxy shape, 366 us.
// Here a workgroup owns 64 consecutive s: their lists are one contiguous CSR range, staged in LDS by coalesced loads and walked
// by 64 s x 4 pair lanes with NPT (c, ci) pairs per thread.  The lists are sorted by rotation (tables.rotation_taps_csr), 3 - 5
// entries each: the walk is synchronised PER ROTATION, so the 64 lanes of a gather instruction read neighbouring taps of ONE
// rotated filter (a few lines) instead of taps of whatever rotations their list positions happen to hold (up to 64 lines).
// Each output still adds its entries in list order: bitwise the old result.
// A workgroup's 64 outputs are an 8 x 8 PATCH of the filter, not a row: under a rotation a row of 64 taps crosses ~100 lines of
// dbank per (pair, rotation), of which it uses 2 - 4 floats each (2.5 GB of L2 fills at the galaxy shape = the 350 us the row
// form took, whatever its index handling); a rotated patch stays within ~15.
constexpr int RBB_CAP = 4096;                            // staged entries per round (32 KB)
template <int NPT>
static __global__ __launch_bounds__(256) void rotate_bank_bwd_kernel(const float* __restrict__ dbank, const int* __restrict__ csr_ptr,
                                                                     const int* __restrict__ csr_r, const int* __restrict__ csr_dst,
                                                                     const float* __restrict__ csr_w, float* __restrict__ dweight,
                                                                     int C, int Cin, int ksz, int R, int accumulate) {
    __shared__ int e_rd[RBB_CAP];                        // (r << 24) | dst   (dst < k2 <= 2^24, checked by the caller)
    __shared__ float e_w[RBB_CAP];
    const int k2 = ksz * ksz;
    const unsigned ldb = (unsigned)Cin * k2;
    const int ls = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int npx = (ksz + 7) >> 3;                      // patches per filter row
    const int py = (int)blockIdx.x / npx, px = (int)blockIdx.x - py * npx;
    const int sy = 8 * py + (ls >> 3), sx = 8 * px + (ls & 7);
    const bool sv = sy < ksz && sx < ksz;
    const int s = sv ? sy * ksz + sx : 0;
    // the patch's lists: eight runs (one per patch row) of up to eight consecutive s, staged back to back
    const int xw = min(8, ksz - 8 * px);                 // valid columns of the patch
    int seg_lo = 0, before = 0, total = 0;               // this lane's run: its first CSR entry, entries staged before it
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int y = 8 * py + j;
        const int a = y < ksz ? csr_ptr[y * ksz + 8 * px] : 0;
        const int b = y < ksz ? csr_ptr[y * ksz + 8 * px + xw] : 0;
        if (j == (ls >> 3)) { seg_lo = a; before = total; }
        total += b - a;
    }
    const int my0 = sv ? before + (csr_ptr[s] - seg_lo) : 0;         // positions in the staged sequence
    const int my1 = sv ? before + (csr_ptr[s + 1] - seg_lo) : 0;
    const unsigned npair = (unsigned)C * Cin;
    const float* src[NPT];
    float acc[NPT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const unsigned p = blockIdx.y * (4 * NPT) + 4 * u + q;
        const unsigned p_ = p < npair ? p : 0;           // (in-bounds addresses for idle lanes)
        src[u] = dbank + (size_t)(p_ / Cin) * R * ldb + (p_ % Cin) * (unsigned)k2;
        acc[u] = 0.f;
    }
    for (int base = 0; base < total; base += RBB_CAP) {
        const int n = min(RBB_CAP, total - base);
        __syncthreads();
        {   // staged position i (global: base + i) -> run j, CSR entry
            int off = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int y = 8 * py + j;
                const int a = y < ksz ? csr_ptr[y * ksz + 8 * px] : 0;
                const int b = y < ksz ? csr_ptr[y * ksz + 8 * px + xw] : 0;
                // entries [a, b) live at staged positions [off, off + b - a)
                const int i0 = max(off, base), i1 = min(off + (b - a), base + n);
                for (int i = i0 + (int)threadIdx.x; i < i1; i += 256) {
                    const int g = a + (i - off);
                    e_rd[i - base] = (csr_r[g] << 24) | csr_dst[g];
                    e_w[i - base] = csr_w[g];
                }
                off += b - a;
            }
        }
        __syncthreads();
        int e = max(my0, base) - base;
        const int e1 = min(my1, base + n) - base;
        int rd = e < e1 ? e_rd[e] : -1;
        for (int r = 0; r < R; ++r) {
            while ((rd >> 24) == r) {                    // (rd == -1: the list is exhausted, no rotation matches)
                const float w = e_w[e];
                const unsigned off = (unsigned)r * ldb + (unsigned)(rd & 0xffffff);
                float v[NPT];
#pragma unroll
                for (int u = 0; u < NPT; ++u) v[u] = src[u][off];
#pragma unroll
                for (int u = 0; u < NPT; ++u) acc[u] += w * v[u];
                ++e;
                rd = e < e1 ? e_rd[e] : -1;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const unsigned p = blockIdx.y * (4 * NPT) + 4 * u + q;
        if (sv && p < npair) {
            float* o = dweight + (size_t)p * k2 + s;
            *o = accumulate ? *o + acc[u] : acc[u];
        }
    }
}

}  // namespace tvae
