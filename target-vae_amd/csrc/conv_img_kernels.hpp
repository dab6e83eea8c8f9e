// Image-resident lifting-convolution kernels for gfx950 (forward and weight gradient).
//
// The im2col operand of the GroupConv GEMMs is a Toeplitz view of the zero-padded image:
//     patch(k = (ci,u,v), p = (h,w)) = ypad[ci][h+u][w+v]  =  img_lds[ (ci*Hp + u)*Wp + v   +   h*Wp + w ]
// i.e. a tap offset that only depends on k PLUS a position offset that only depends on the output position.
// So the padded image (Hp x Wp fp32, Wp = Hp+1 to break row-wrap bank conflicts; 36 KiB at 64x64/pad 16) is
// loaded into LDS once and the MFMA B fragments are read STRAIGHT from it:
//   forward : lane = position  -> per-lane constant position offset + wave-uniform tap offset (ktab)
//   wgrad   : lane = tap (u,v) -> per-lane constant tap offset + wave-uniform position offset (ptab)
// Consecutive lanes read consecutive LDS words (conflict-free ds_read_b32); no im2col staging, no per-element
// guards, the zero padding lives in the LDS image.  Only the A operand (filter bank / dY) is staged through LDS.
// Same 128x128x16 tile / 2x2 waves / v_mfma_f32_32x32x2_f32 / LDS epilogue as gemm_f32_mfma.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_f32_mfma.hpp"

namespace tvae {

constexpr int CONV_A_FLOATS = 2 * BK * LDS_LD;     // double-buffered A tile
constexpr int CONV_TAB_INTS = 2 * BK;              // double-buffered uniform-offset table

// Rows of the padded image a forward tile needs: BN consecutive positions span at most (BN-1)/Ho + 2 output rows,
// each of which reads ksz input rows.
static inline int conv_fwd_img_rows(int n, int ksz, int pad) {
    const int Hp = n + 2 * pad, Ho = Hp - ksz + 1;
    const int rows = (BN - 1) / Ho + 2 + ksz - 1;
    return rows < Hp ? rows : Hp;
}
// Rows a weight-gradient tile needs (single input channel): BN consecutive taps span at most (BN-1)/ksz + 2 tap rows,
// each combined with Ho output rows.  With several input channels the whole padded image is kept.
static inline int conv_wgrad_img_rows(int Cin, int n, int ksz, int pad, int nh = 1) {
    const int Hp = n + 2 * pad, Ho = Hp - ksz + 1;
    if (Cin != 1) return Hp;
    const int rows = (BN * nh - 1) / ksz + 2 + Ho - 1;
    return rows < Hp ? rows : Hp;
}
static inline size_t conv_img_lds_bytes(int Cin, int rows, int n, int pad, int mh = 1, int kb = BK) {
    const int Wp = n + 2 * pad + 1;
    long tot = 2 * kb * (mh * 128 + 4) + 2 * kb + (long)Cin * rows * Wp;
    if (tot < 64 * 128) tot = 64 * 128;            // the epilogue staging tile aliases the whole region
    return (size_t)tot * sizeof(float);
}

// rows [row0, row0 + rows) of the zero-padded image of every input channel -> img[ci][rows][Wp]
__device__ __forceinline__ void load_padded_image(float* img, const float* __restrict__ y, int b, const ConvGeom& g,
                                                  int row0, int rows, int Wp) {
    const int total = g.Cin * rows * Wp;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int ci = i / (rows * Wp);
        const int r = i - ci * (rows * Wp);
        const int yy = r / Wp, xx = r - yy * Wp;
        const int iy = row0 + yy - g.pad, ix = xx - g.pad;
        float v = 0.f;
        if (iy >= 0 && iy < g.n && ix >= 0 && ix < g.n) v = y[((long)(b * g.Cin + ci) * g.n + iy) * g.n + ix];
        img[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Forward:  out[cr][img, p] = act( sum_k bank[cr][k] * patch(k, p) + bias[c] )
// grid.x = tilesM * B * tilesPerImg (position tiles fastest -> concurrently running workgroups share one bank panel
// in L2).  VEC: K % 16 == 0 and M % 128 == 0 -> unguarded float4 loads of the bank with incremented pointers.
// ------------------------------------------------------------------------------------------
template <bool VEC, int MH>
static __global__ __launch_bounds__(GEMM_THREADS, (MH == 1 ? 3 : 2))
void conv1_fwd_img_kernel(const float* __restrict__ bank, const float* __restrict__ y, ConvGeom g, Epilogue ep, int M,
                          int K, int tilesPerImg, int rows) {
    // MH = number of 128-row halves per workgroup tile: 1 -> 128 x 128 (wave 64 x 64), 2 -> 256 x 128 (wave 128 x 64:
    // 8 MFMAs per operand wait, half the B reads per MFMA, image / tap table shared by twice the rows)
    constexpr int ALD = MH * 128 + 4;               // A tile row stride (floats)
    constexpr int AFL = 2 * BK * ALD;               // double-buffered A tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    int* ktab = reinterpret_cast<int*>(smem + AFL);
    float* img = smem + AFL + CONV_TAB_INTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int Hp = g.n + 2 * g.pad, Wp = Hp + 1;
    const int per_m = g.B * tilesPerImg;
    const int tile_m = blockIdx.x / per_m;
    const int rest = blockIdx.x - tile_m * per_m;
    const int b = rest / tilesPerImg;
    const int p0 = (rest - b * tilesPerImg) * BN;
    const int m0 = tile_m * (BM * MH);
    const int hmin = p0 / g.Ho;                     // first output row of the tile = first padded-image row kept
    // Zero skipping: tap row u only meets image rows h+u-pad in [0,n) for h in the tile's output rows [hmin,hmax];
    // every other tap row multiplies pure zero padding.  (Single channel: the kept taps are one contiguous k range,
    // rounded outwards to the k-step so the float4 bank loads stay aligned.)
    int kbeg = 0, kstop = K;
    if (g.Cin == 1) {
        const int plast = min(g.P - 1, p0 + BN - 1);
        const int hmax = plast / g.Ho;
        const int ulo = max(0, g.pad - hmax);
        const int uhi = min(g.ksz - 1, g.pad + g.n - 1 - hmin);
        kbeg = (ulo * g.ksz) / BK * BK;
        kstop = min(K, ((uhi + 1) * g.ksz + BK - 1) / BK * BK);
        if (kstop < kbeg) kstop = kbeg;
    }
    const int nk = (kstop - kbeg + BK - 1) / BK;

    load_padded_image(img, y, b, g, hmin, rows, Wp);

    int boff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int p = p0 + wn * 64 + j * 32 + (lane & 31);
        if (p >= g.P) p = g.P - 1;                  // padded columns read a valid address; never stored
        const int h = p / g.Ho, w = p - h * g.Ho;
        boff[j] = (h - hmin) * Wp + w;
    }

    // tap-offset table for one k-step: entry kk = (ci*rows + u)*Wp + v  (0 beyond K: the A tile is zero there).
    // Thread tid < 16 owns entry kk = tid and walks its (ci,u,v) INCREMENTALLY (k advances by 16 per fill), so wave 0
    // carries no integer divisions in the k-loop -- every barrier waits for the slowest wave.
    int tk = kbeg + (tid & 15), tci, tu, tv;
    {
        tci = tk / g.K2;
        const int rem = tk - tci * g.K2;
        tu = rem / g.ksz;
        tv = rem - tu * g.ksz;
    }
    auto fill_ktab = [&](int* tab, int /*k0*/) {
        if (tid < BK) {
            tab[(tid & 1) * 8 + (tid >> 1)] = (tk < K) ? (tci * rows + tu) * Wp + tv : 0;
            tk += BK;
            tv += BK;
            while (tv >= g.ksz) { tv -= g.ksz; if (++tu == g.ksz) { tu = 0; ++tci; } }
        }
    };

    f32x16 acc[MH][2][2];
#pragma unroll
    for (int hh = 0; hh < MH; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[hh][i][j][r] = 0.f;

    // A staging: rows (tid >> 2) + 64 i, k quad (tid & 3) * 4
    constexpr int NA = 2 * MH;
    float ra[MH][8];
    float4 va[NA];
    const float* pa[NA];
    const int am = tid >> 2, akq = (tid & 3) * 4;
    const int kkA = tid & 15, xbA = tid >> 4;       // generic (guarded) mapping
    auto load_generic = [&](int k0) {
        const int k = k0 + kkA;
#pragma unroll
        for (int hh = 0; hh < MH; ++hh)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = m0 + hh * 128 + xbA + 16 * j;
                ra[hh][j] = (k < K && m < M) ? bank[(long)m * K + k] : 0.f;
            }
    };
    if (VEC) {
#pragma unroll
        for (int i = 0; i < NA; ++i) pa[i] = bank + (long)(m0 + am + 64 * i) * K + kbeg + akq;
        if (nk > 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) va[i] = *reinterpret_cast<const float4*>(pa[i]);
        }
    } else {
        load_generic(kbeg);
    }
    auto store_a = [&](float* S) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = am + 64 * i;
                S[(akq + 0) * ALD + m] = va[i].x;
                S[(akq + 1) * ALD + m] = va[i].y;
                S[(akq + 2) * ALD + m] = va[i].z;
                S[(akq + 3) * ALD + m] = va[i].w;
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < MH; ++hh)
#pragma unroll
                for (int j = 0; j < 8; ++j) S[kkA * ALD + hh * 128 + xbA + 16 * j] = ra[hh][j];
        }
    };
    if (nk > 0) store_a(As);
    fill_ktab(ktab, kbeg);
    __syncthreads();

    const int arow = wm * 64 + (lane & 31);
    const int khalf = lane >> 5;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < nk;
        if (more) {
            if (VEC) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    pa[i] += BK;
                    va[i] = *reinterpret_cast<const float4*>(pa[i]);
                }
            } else {
                load_generic(kbeg + (t + 1) * BK);
            }
        }
        const float* as = As + cur * (BK * ALD);
        // this lane half's 8 tap offsets of the k-step: two broadcast ds_read_b128, so that the fragment reads
        // below carry no LDS->LDS dependency and can be pipelined under the MFMAs
        const int4* kt4 = reinterpret_cast<const int4*>(ktab + cur * BK + khalf * 8);
        const int4 t0 = kt4[0], t1 = kt4[1];
        const int kos[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + khalf;
            const float b0 = img[boff[0] + kos[s]];
            const float b1 = img[boff[1] + kos[s]];
#pragma unroll
            for (int hh = 0; hh < MH; ++hh) {
                const float a0 = as[kk * ALD + hh * 128 + arow];
                const float a1 = as[kk * ALD + hh * 128 + arow + 32];
                acc[hh][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[hh][0][0], 0, 0, 0);
                acc[hh][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[hh][0][1], 0, 0, 0);
                acc[hh][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[hh][1][0], 0, 0, 0);
                acc[hh][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[hh][1][1], 0, 0, 0);
            }
        }
        if (more) {
            store_a(As + (cur ^ 1) * (BK * ALD));
            fill_ktab(ktab + (cur ^ 1) * BK, kbeg + (t + 1) * BK);
        }
        __syncthreads();
    }
    const int p = p0 + (tid & 127);
#pragma unroll
    for (int hh = 0; hh < MH; ++hh) {
        if (hh) __syncthreads();
        tile_epilogue(acc[hh], smem, ep, m0 + hh * 128, M, b * g.P + p, p < g.P, nullptr, 0, g.B * g.P);
    }
}


// ------------------------------------------------------------------------------------------
// Weight gradient:  dbank[cr][n = (ci,u,v)] = sum_{img, p} dY[cr][img, p] * patch(n, p)
// grid.x = tilesM * tilesN (n fastest), grid.y = split over images.  dY is feature-major [c][img][r][p] (ld = lddy).
// ------------------------------------------------------------------------------------------
template <int MH, int KB, int NH>
static __global__ __launch_bounds__(GEMM_THREADS, ((MH == 1 && NH == 1) ? 3 : 2))
void conv1_wgrad_img_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ y, ConvGeom g,
                            Epilogue ep, int M, int N, int imgs_per_split, float* ws, int tilesN, int rows, int nsplits,
                            int ngroups) {
    // NH = 128-column (tap) halves per workgroup tile: NH = 2 halves how often a dY panel is re-read and the A reads
    // per MFMA.  MH = 128-row halves per workgroup tile (1: 128 x 128, 2: 256 x 128 with 128 x 64 per wave); KB = positions per
    // k-step (16 or 32: the image rows of a wgrad tile are small, so a 32-deep step halves the barriers at equal occupancy)
    constexpr int ALD = MH * 128 + 4;
    constexpr int AFL = 2 * KB * ALD;
    constexpr int RG = GEMM_THREADS / KB;           // row groups of the A staging map
    constexpr int NJ = (128 * MH) / RG;             // A rows per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    int* ptab = reinterpret_cast<int*>(smem + AFL);
    float* img = smem + AFL + 2 * KB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int Hp = g.n + 2 * g.pad, Wp = Hp + 1;
    // XCD-aware mapping (1-D grid): the tilesN tap-tiles of one group g = (row-tile, image-slice) all read the same dY
    // panel, and workgroup ids are dealt round-robin over the 8 XCDs, so group g takes the ids congruent to g mod 8:
    // the panel is then fetched into ONE L2 instead of eight (only speed depends on the placement, never results).
    int tile_n, gidx;
    {
        const int bid = blockIdx.x;
        if ((ngroups & 7) == 0) {
            const int x = bid & 7, q = bid >> 3;
            tile_n = q % tilesN;
            gidx = (q / tilesN) * 8 + x;
        } else {
            tile_n = bid % tilesN;
            gidx = bid / tilesN;
        }
    }
    const int split = gidx % nsplits, tile_m = gidx / nsplits;
    const int m0 = tile_m * (BM * MH), n0 = tile_n * (BN * NH);
    const int ib = split * imgs_per_split;
    const int ie = min(g.B, ib + imgs_per_split);
    // rows == Hp: whole padded image(s) resident; otherwise (single channel) only rows [ulo, ulo + rows)
    const int ulo = (rows == Hp) ? 0 : (n0 / g.ksz);
    // Zero skipping (single channel): the tile's taps have rows u in [ua, ub]; output row h only meets the image if
    // pad <= h+u <= pad+n-1 for some such u, i.e. h in [pad-ub, pad+n-1-ua] -- a contiguous position range.
    int pbeg = 0, pend = g.P;
    if (g.Cin == 1) {
        const int ua = n0 / g.ksz, ub = min(N - 1, n0 + BN * NH - 1) / g.ksz;
        const int hlo = max(0, g.pad - ub), hhi = min(g.Ho - 1, g.pad + g.n - 1 - ua);
        pbeg = hlo * g.Ho;
        pend = max(pbeg, (hhi + 1) * g.Ho);
    }
    const int nk = (pend - pbeg + KB - 1) / KB;

    int noff[NH][2];
#pragma unroll
    for (int hn = 0; hn < NH; ++hn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int nn = n0 + hn * 128 + wn * 64 + j * 32 + (lane & 31);
            if (nn >= N) nn = N - 1;
            const int ci = nn / g.K2, rem = nn - ci * g.K2;
            const int u = rem / g.ksz, v = rem - u * g.ksz;
            noff[hn][j] = (ci * rows + (u - ulo)) * Wp + v;
        }
    // A rows (mapping K: kk = tid % KB, rows xb + RG j)
    const int kkA = tid % KB, xb = tid / KB;
    long rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int m = m0 + xb + RG * j;
        if (m >= M) m = M - 1;                      // clamped rows are never stored
        const int c = m / g.R, rr = m - c * g.R;
        rowoff[j] = (long)c * lddy + (long)rr * g.P;
    }
    // position-offset table: thread tid < 16 owns entry tid and walks (h, w) incrementally (no divisions in the loop)
    int tp, th, tw;
    auto reset_ptab = [&]() {
        tp = pbeg + (tid % KB);
        th = tp / g.Ho;
        tw = tp - th * g.Ho;
    };
    auto fill_ptab = [&](int* tab, int /*pfrom*/) {
        if (tid < KB) {
            tab[(tid & 1) * (KB / 2) + (tid >> 1)] = (tp < pend) ? th * Wp + tw : 0;
            tp += KB;
            tw += KB;
            while (tw >= g.Ho) { tw -= g.Ho; ++th; }
        }
    };

    f32x16 acc[MH * NH][2][2];
#pragma unroll
    for (int hh = 0; hh < MH * NH; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[hh][i][j][r] = 0.f;

    const int arow = wm * 64 + (lane & 31);
    const int khalf = lane >> 5;
    float ra[NJ];
    for (int b = ib; b < ie; ++b) {
        __syncthreads();                             // previous image fully consumed
        load_padded_image(img, y, b, g, ulo, rows, Wp);
        reset_ptab();
        const float* dyb = dy + (long)b * g.R * g.P + pbeg + kkA;
        {
            const bool kok = (pbeg + kkA) < pend;
#pragma unroll
            for (int j = 0; j < NJ; ++j) ra[j] = kok ? dyb[rowoff[j]] : 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) As[kkA * ALD + xb + RG * j] = ra[j];
        }
        fill_ptab(ptab, pbeg);
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            const bool more = (t + 1) < nk;
            if (more) {
                const int pk = (t + 1) * KB;
                const bool kok = (pbeg + pk + kkA) < pend;
#pragma unroll
                for (int j = 0; j < NJ; ++j) ra[j] = kok ? dyb[rowoff[j] + pk] : 0.f;
            }
            const float* as = As + cur * (KB * ALD);
            const int4* pt4 = reinterpret_cast<const int4*>(ptab + cur * KB + khalf * (KB / 2));
            int pos[KB / 2];
#pragma unroll
            for (int q = 0; q < KB / 8; ++q) {
                const int4 tq = pt4[q];
                pos[4 * q] = tq.x; pos[4 * q + 1] = tq.y; pos[4 * q + 2] = tq.z; pos[4 * q + 3] = tq.w;
            }
#pragma unroll
            for (int s = 0; s < KB / 2; ++s) {
                const int kk = 2 * s + khalf;
                float bq[NH][2];
#pragma unroll
                for (int hn = 0; hn < NH; ++hn) {
                    bq[hn][0] = img[noff[hn][0] + pos[s]];
                    bq[hn][1] = img[noff[hn][1] + pos[s]];
                }
#pragma unroll
                for (int hh = 0; hh < MH; ++hh) {
                    const float a0 = as[kk * ALD + hh * 128 + arow];
                    const float a1 = as[kk * ALD + hh * 128 + arow + 32];
#pragma unroll
                    for (int hn = 0; hn < NH; ++hn) {
                        f32x16 (&ac)[2][2] = acc[hh * NH + hn];
                        ac[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[hn][0], ac[0][0], 0, 0, 0);
                        ac[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[hn][1], ac[0][1], 0, 0, 0);
                        ac[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[hn][0], ac[1][0], 0, 0, 0);
                        ac[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[hn][1], ac[1][1], 0, 0, 0);
                    }
                }
            }
            if (more) {
                float* an = As + (cur ^ 1) * (KB * ALD);
#pragma unroll
                for (int j = 0; j < NJ; ++j) an[kkA * ALD + xb + RG * j] = ra[j];
                fill_ptab(ptab + (cur ^ 1) * KB, pbeg + (t + 1) * KB);
            }
            __syncthreads();
        }
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < MH; ++hh)
#pragma unroll
        for (int hn = 0; hn < NH; ++hn) {
            if (hh + hn) __syncthreads();
            const int n = n0 + hn * 128 + (tid & 127);
            tile_epilogue(acc[hh * NH + hn], smem, ep, m0 + hh * 128, M, n, n < N, ws, split, N);
        }
}

}  // namespace tvae
