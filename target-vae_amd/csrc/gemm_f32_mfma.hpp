// f32-in / f32-accumulate MFMA GEMM core for gfx950 (MI355X), with pluggable operand loaders.
//
//   C[m][n] = epilogue( sum_k A(m,k) * B(k,n) )          m<M (feature rows), n<N (positions / pixels)
//
// All activations of the TARGET-VAE hot path are kept "feature-major": [feature][batch*position]
// with the position index contiguous, so every layer is W[M][K] * X[K][N] and the MFMA D tile
// (lane = column) stores 128-B coalesced row segments.  The same core serves
//   * conv1 forward  (B operand = implicit im2col window of the zero-padded image)
//   * conv1 wgrad    (A = dY gathered [cr][(img,p)], B = implicit window, reduction over img*p)
//   * 1x1x1 conv / linear layers: fwd, dgrad (A col-major), wgrad (both operands k-contiguous).
//
// Tile: 128x128x16 per 256-thread workgroup, 2x2 waves, each wave 2x2 tiles of
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 cycles/instr, 157 TF peak).  Operands are
// staged global -> registers -> LDS (double buffered, one barrier per k-step); LDS tiles are
// k-major  S[k][x]  with a 132-float row so that
//   * MFMA fragment reads (32 consecutive x at fixed k) are conflict-free ds_read_b32,
//   * both staging mappings below write with at most 2-way conflicts (free for ds_write_b32).
//
// Operand lane maps (cdna_hip_programming.md section 3): for 32x32x2f32 lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; D: col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#pragma once
#include <hip/hip_runtime.h>

namespace tvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 16;
constexpr int GEMM_THREADS = 256;
constexpr int LDS_LD = 132;

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2 };

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// Epilogue: bias / per-image bias / residual / activation / activation-derivative mask / store
// ------------------------------------------------------------------------------------------
struct Epilogue {
    float* C = nullptr;
    long ldc = 0;
    const float* bias = nullptr;   // bias[m >> bias_shift]
    int bias_shift = 0;
    const float* gbias = nullptr;  // gbias[(n / group) * ldg + m]   (per-image latent term)
    long ldg = 0;
    int group = 1;
    const float* res = nullptr;    // res[m*ldres + n] added before act / mask
    long ldres = 0;
    const float* aux = nullptr;    // aux[m*ldaux + n]: saved activation whose derivative masks the result
    long ldaux = 0;
    int act = ACT_NONE;            // applied to (acc + bias + gbias + res)
    int mask = ACT_NONE;           // multiply by act'(aux): lrelu -> (aux>0 ? 1 : slope), tanh -> 1-aux^2
    float slope = 0.01f;
    int accumulate = 0;            // C += instead of C =
    long ctile = 0;                // dense_x6_kernel only: != 0 -> column tile t (128 columns) starts at C + t*ctile
    float* amax_out = nullptr;     // dense_x6_kernel (generic epilogue) only: atomic max |stored value| into this (zeroed) word --
                                   // the h3 bound of the launch that streams this output next (round 6)
    // conv1 output remap: column n = img*convP + p, row m = c*convR + r  ->  C[c*ldc + img*convR*convP + r*convP + p]
    int convR = 0;                 // power of two (reference allows R in {4,8,16})
    int conv_shift = 0;            // log2(convR)
    int convP = 0;

    // Column-dependent terms are prepared once per thread (each thread owns one column in the epilogue).
    struct Col {
        long coff;             // offset of column n inside a row of C
        const float* gb;       // gbias row of this column's image (or nullptr)
    };
    __device__ __forceinline__ Col prep(int n) const {
        Col c;
        if (convP > 0) {
            const int img = n / convP, p = n - img * convP;
            c.coff = (long)img * convR * convP + p;
        } else {
            c.coff = n;
        }
        c.gb = gbias ? gbias + (long)(n / group) * ldg : nullptr;
        return c;
    }
    __device__ __forceinline__ void store(int m, int n, const Col& c, float v) const {
        float x = v;
        if (bias) x += bias[m >> bias_shift];
        if (c.gb) x += c.gb[m];
        if (res) x += res[(long)m * ldres + n];
        if (act == ACT_LRELU) x = x > 0.f ? x : x * slope;
        else if (act == ACT_TANH) x = tanhf(x);
        if (mask != ACT_NONE) {
            const float a = aux[(long)m * ldaux + n];
            x *= (mask == ACT_LRELU) ? (a > 0.f ? 1.f : slope) : (1.f - a * a);
        }
        long i;
        if (convP > 0) i = (long)(m >> conv_shift) * ldc + (long)(m & ((1 << conv_shift) - 1)) * convP + c.coff;
        else i = (long)m * ldc + c.coff;
        if (accumulate) C[i] += x; else C[i] = x;
    }
};

// ------------------------------------------------------------------------------------------
// Operand loaders.  Each thread owns 8 elements of a [BK][128] tile.
//   mapping K ("k fast"):  kk = tid & 15,      x = (tid >> 4) + 16 j      -- for k-contiguous memory
//   mapping X ("x fast"):  x  = tid & 127,     kk = (tid >> 7) * 8 + j    -- for x-contiguous memory
// ------------------------------------------------------------------------------------------
struct LoadKContig {           // element (x, k) at ptr[x*ld + k]
    const float* ptr; long ld; int X;
    int x0, kk, xb;
    float r_[8]; int k_, kend_;
    __device__ __forceinline__ void begin(int x0_, int tid, int kbeg, int kend) { init(x0_, tid); k_ = kbeg; kend_ = kend; }
    __device__ __forceinline__ void fetch() { load(r_, k_, kend_); k_ += BK; }
    __device__ __forceinline__ void commit(float* S) const { store(S, r_); }
    __device__ __forceinline__ void init(int x0_, int tid) { x0 = x0_; kk = tid & 15; xb = tid >> 4; }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        const int k = k0 + kk;
        const bool kok = k < kend;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = x0 + xb + 16 * j;
            r[j] = (kok && x < X) ? ptr[(long)x * ld + k] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* S, const float (&r)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) S[kk * LDS_LD + xb + 16 * j] = r[j];
    }
};

struct LoadXContig {           // element (x, k) at ptr[k*ld + x]
    const float* ptr; long ld; int X;
    int x0, x, kh;
    float r_[8]; int k_, kend_;
    __device__ __forceinline__ void begin(int x0_, int tid, int kbeg, int kend) { init(x0_, tid); k_ = kbeg; kend_ = kend; }
    __device__ __forceinline__ void fetch() { load(r_, k_, kend_); k_ += BK; }
    __device__ __forceinline__ void commit(float* S) const { store(S, r_); }
    __device__ __forceinline__ void init(int x0_, int tid) { x0 = x0_; x = tid & 127; kh = tid >> 7; }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        const bool xok = (x0 + x) < X;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + kh * 8 + j;
            r[j] = (xok && k < kend) ? ptr[(long)k * ld + x0 + x] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* S, const float (&r)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) S[(kh * 8 + j) * LDS_LD + x] = r[j];
    }
};


// ------------------------------------------------------------------------------------------
// Fast-path loaders (no guards, float4 global loads, incremented pointers): require the tile dimension to be a
// multiple of 128, the reduction range a multiple of 16, ld % 4 == 0 and a 16-B aligned base.
// ------------------------------------------------------------------------------------------
struct LoadKContigV4 {         // element (x, k) at ptr[x*ld + k]; thread: rows (tid>>2) + 64 i, k quad (tid&3)*4
    const float* ptr; long ld; int X;
    const float* p_[2]; float4 v_[2]; int am, akq;
    __device__ __forceinline__ void begin(int x0, int tid, int kbeg, int) {
        am = tid >> 2; akq = (tid & 3) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) p_[i] = ptr + (long)(x0 + am + 64 * i) * ld + kbeg + akq;
    }
    __device__ __forceinline__ void fetch() {
#pragma unroll
        for (int i = 0; i < 2; ++i) { v_[i] = *reinterpret_cast<const float4*>(p_[i]); p_[i] += BK; }
    }
    __device__ __forceinline__ void commit(float* S) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = am + 64 * i;
            S[(akq + 0) * LDS_LD + m] = v_[i].x;
            S[(akq + 1) * LDS_LD + m] = v_[i].y;
            S[(akq + 2) * LDS_LD + m] = v_[i].z;
            S[(akq + 3) * LDS_LD + m] = v_[i].w;
        }
    }
};

struct LoadXContigV4 {         // element (x, k) at ptr[k*ld + x]; thread: x quad (tid&31)*4, k rows (tid>>5) + 8 i
    const float* ptr; long ld; int X;
    const float* p_[2]; float4 v_[2]; int x4, kr;
    __device__ __forceinline__ void begin(int x0, int tid, int kbeg, int) {
        x4 = (tid & 31) * 4; kr = tid >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) p_[i] = ptr + (long)(kbeg + kr + 8 * i) * ld + x0 + x4;
    }
    __device__ __forceinline__ void fetch() {
#pragma unroll
        for (int i = 0; i < 2; ++i) { v_[i] = *reinterpret_cast<const float4*>(p_[i]); p_[i] += (long)BK * ld; }
    }
    __device__ __forceinline__ void commit(float* S) const {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<float4*>(S + (kr + 8 * i) * LDS_LD + x4) = v_[i];
    }
};

// Geometry of the lifting convolution (GroupConv.forward, reference src/models.py:202-225).
struct ConvGeom {
    int B, Cin, n, ksz, pad, Ho, R;
    int P;     // Ho*Ho
    int K2;    // ksz*ksz
};

// B operand of conv1 forward: element (k = (ci,u,v), col = (img,p)) = ypad[img][ci][h+u][w+v].  Mapping X.
struct LoadConvPatchFwd {
    const float* y; ConvGeom g; int Ntot;
    int x, kh, img, h, w; bool nok;
    float r_[8]; int k_, kend_;
    __device__ __forceinline__ void begin(int x0_, int tid, int kbeg, int kend) { init(x0_, tid); k_ = kbeg; kend_ = kend; }
    __device__ __forceinline__ void fetch() { load(r_, k_, kend_); k_ += BK; }
    __device__ __forceinline__ void commit(float* S) const { store(S, r_); }
    __device__ __forceinline__ void init(int n0, int tid) {
        x = tid & 127; kh = tid >> 7;
        const int nn = n0 + x;
        nok = nn < Ntot;
        const int nc = nok ? nn : 0;
        img = nc / g.P;
        const int p = nc - img * g.P;
        h = p / g.Ho;
        w = p - h * g.Ho;
    }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        int k = k0 + kh * 8;
        int ci = k / g.K2;
        int rem = k - ci * g.K2;
        int u = rem / g.ksz;
        int v = rem - u * g.ksz;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = h + u - g.pad, ix = w + v - g.pad;
            const bool ok = nok && (k + j) < kend && iy >= 0 && iy < g.n && ix >= 0 && ix < g.n;
            r[j] = ok ? y[((long)(img * g.Cin + ci) * g.n + iy) * g.n + ix] : 0.f;
            if (++v == g.ksz) { v = 0; if (++u == g.ksz) { u = 0; ++ci; } }
        }
    }
    __device__ __forceinline__ void store(float* S, const float (&r)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) S[(kh * 8 + j) * LDS_LD + x] = r[j];
    }
};

// B operand of conv1 wgrad: element (kr = (img,p), col = (ci,u,v)) = ypad[img][ci][h+u][w+v].  Mapping X.
struct LoadConvPatchWgrad {
    const float* y; ConvGeom g; int Ntot;   // Ntot = Cin*K2
    int x, kh, ci, u, v; bool nok;
    float r_[8]; int k_, kend_;
    __device__ __forceinline__ void begin(int x0_, int tid, int kbeg, int kend) { init(x0_, tid); k_ = kbeg; kend_ = kend; }
    __device__ __forceinline__ void fetch() { load(r_, k_, kend_); k_ += BK; }
    __device__ __forceinline__ void commit(float* S) const { store(S, r_); }
    __device__ __forceinline__ void init(int n0, int tid) {
        x = tid & 127; kh = tid >> 7;
        const int nn = n0 + x;
        nok = nn < Ntot;
        const int nc = nok ? nn : 0;
        ci = nc / g.K2;
        const int rem = nc - ci * g.K2;
        u = rem / g.ksz;
        v = rem - u * g.ksz;
    }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        int kr = k0 + kh * 8;
        int img = kr / g.P;
        int p = kr - img * g.P;
        int h = p / g.Ho;
        int w = p - h * g.Ho;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = h + u - g.pad, ix = w + v - g.pad;
            const bool ok = nok && (kr + j) < kend && iy >= 0 && iy < g.n && ix >= 0 && ix < g.n;
            r[j] = ok ? y[((long)(img * g.Cin + ci) * g.n + iy) * g.n + ix] : 0.f;
            if (++w == g.Ho) { w = 0; if (++h == g.Ho) { h = 0; ++img; } }
        }
    }
    __device__ __forceinline__ void store(float* S, const float (&r)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) S[(kh * 8 + j) * LDS_LD + x] = r[j];
    }
};

// A operand of conv1 wgrad: element (row = cr = c*R + r, kr = (img,p)) of the pre-activation gradient stored
// feature-major [c][img][r][p] (ld = B*R*P).  Mapping K.
struct LoadConvDY {
    const float* dy; long ld; int M; int R; int P;
    int x0, kk, xb;
    long rowoff[8];
    float r_[8]; int k_, kend_;
    __device__ __forceinline__ void begin(int x0_, int tid, int kbeg, int kend) { init(x0_, tid); k_ = kbeg; kend_ = kend; }
    __device__ __forceinline__ void fetch() { load(r_, k_, kend_); k_ += BK; }
    __device__ __forceinline__ void commit(float* S) const { store(S, r_); }
    __device__ __forceinline__ void init(int x0_, int tid) {
        x0 = x0_; kk = tid & 15; xb = tid >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = x0 + xb + 16 * j;
            const int mc = m < M ? m : 0;
            const int c = mc / R, rr = mc - c * R;
            rowoff[j] = (long)c * ld + (long)rr * P;
        }
    }
    __device__ __forceinline__ void load(float (&r)[8], int k0, int kend) const {
        const int kr = k0 + kk;
        const bool kok = kr < kend;
        const int krc = kok ? kr : 0;
        const int img = krc / P;
        const int p = krc - img * P;
        const long col = (long)img * R * P + p;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = x0 + xb + 16 * j;
            r[j] = (kok && m < M) ? dy[rowoff[j] + col] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* S, const float (&r)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) S[kk * LDS_LD + xb + 16 * j] = r[j];
    }
};

// ------------------------------------------------------------------------------------------
// Shared tile epilogue through LDS (the k-loop's LDS is free after its last barrier): two passes of 64 rows x 128
// columns.  Accumulator lane map: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Reading back one
// column per thread gives 512-B coalesced global rows and keeps bias/activation/mask out of the unrolled part.
// `ct` needs 64*128 floats.  `n` is this thread's global column (tid&127 within the tile), `nok` its validity.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_epilogue(f32x16 (&acc)[2][2], float* ct, const Epilogue& ep, int m0, int M, int n,
                                              bool nok, float* ws, int split, int N) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ecol = tid & 127;
    const Epilogue::Col ecl = ep.prep(nok ? n : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ct[rl * 128 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int rl = (tid >> 7) + 2 * it;                       // 0..63
            const int m = m0 + (rl >> 5) * 64 + i * 32 + (rl & 31);
            if (nok && m < M) {
                const float v = ct[rl * 128 + ecol];
                if (ws) ws[((long)split * M + m) * N + n] = v;
                else ep.store(m, n, ecl, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Vector epilogue of the aligned fast path (M % 128 == 0, N % 128 == 0, every pointer 16-B aligned, leading
// dimensions % 4 == 0, plain row-major C: no conv remap, no per-image bias, no accumulate).  The tile is staged
// 64 rows at a time through LDS with a 136-float row pitch (lanes 32..63 of the accumulator layout sit 4 rows
// lower = +544 floats = bank +32: conflict-free ds_write_b32), then every thread moves float4s: 32 lanes cover one
// 512-B output row, so bias / residual / mask / store are one instruction per 4 elements instead of per element.
// `ct` needs 64*136 floats.
// ------------------------------------------------------------------------------------------
constexpr int EP_LD = 136;

__device__ __forceinline__ float4 act_mask4(const Epilogue& ep, float4 x, const float* __restrict__ auxp) {
    if (ep.act == ACT_LRELU) {
        x.x = x.x > 0.f ? x.x : x.x * ep.slope; x.y = x.y > 0.f ? x.y : x.y * ep.slope;
        x.z = x.z > 0.f ? x.z : x.z * ep.slope; x.w = x.w > 0.f ? x.w : x.w * ep.slope;
    } else if (ep.act == ACT_TANH) {
        x.x = tanhf(x.x); x.y = tanhf(x.y); x.z = tanhf(x.z); x.w = tanhf(x.w);
    }
    if (ep.mask != ACT_NONE) {
        const float4 a = *reinterpret_cast<const float4*>(auxp);
        if (ep.mask == ACT_LRELU) {
            x.x *= a.x > 0.f ? 1.f : ep.slope; x.y *= a.y > 0.f ? 1.f : ep.slope;
            x.z *= a.z > 0.f ? 1.f : ep.slope; x.w *= a.w > 0.f ? 1.f : ep.slope;
        } else {
            x.x *= 1.f - a.x * a.x; x.y *= 1.f - a.y * a.y; x.z *= 1.f - a.z * a.z; x.w *= 1.f - a.w * a.w;
        }
    }
    return x;
}

__device__ __forceinline__ void tile_epilogue_v4(f32x16 (&acc)[2][2], float* ct, const Epilogue& ep, int m0, int n0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c4 = (tid & 31) * 4;
    const int r8 = tid >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ct[rl * EP_LD + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 8 + r8;                                   // staged row 0..63
            const int m = m0 + (rl >> 5) * 64 + i * 32 + (rl & 31);
            float4 x = *reinterpret_cast<const float4*>(ct + rl * EP_LD + c4);
            if (ep.bias) {
                const float b = ep.bias[m >> ep.bias_shift];
                x.x += b; x.y += b; x.z += b; x.w += b;
            }
            if (ep.res) {
                const float4 q = *reinterpret_cast<const float4*>(ep.res + (long)m * ep.ldres + n0 + c4);
                x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
            }
            x = act_mask4(ep, x, ep.aux + (long)m * ep.ldaux + n0 + c4);
            *reinterpret_cast<float4*>(ep.C + (long)m * ep.ldc + n0 + c4) = x;
        }
    }
}

// ------------------------------------------------------------------------------------------
// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (workgroup id b runs on XCD b & 7) and each
// XCD has a private L2, so tiles that share an HBM operand panel must (a) carry ids congruent mod 8 and (b) be
// adjacent in that XCD's dispatch order so that they are resident together and the panel is fetched from HBM once:
//   no split-K : the tilesM tiles of one column panel X[:, n-tile] form a group (the weight operand is tiny and
//                L2-resident everywhere);
//   split-K    : all tilesM*tilesN tiles of one K-slice form a group (each A / B panel of the slice is shared by
//                tilesN / tilesM of them).
// Group g lives on XCD g & 7.  The 1-D grid is padded to 8*ceil(groups/8)*group_size; padded ids return false.
// Measured on the 512x512 decoder layers (FETCH_SIZE): fwd 8.9 -> GB, wgrad 22.9 -> GB per launch.
// ------------------------------------------------------------------------------------------
struct TileMap {
    int tilesM, tilesN, splits;
    // batched GEMMs with per-problem operands (bt = row tiles per problem, nch = column chunks per problem): a group is
    // (problem, column chunk): the problem's A cells stay in ONE L2 while its column tiles stream, and the bt row tiles
    // of a column panel are adjacent so the panel is fetched once.
    int bt = 0, nch = 1;
    // pack (round 6; plain tiles only: bt == 0, splits <= 1): EIGHT CONSECUTIVE column tiles on one XCD instead of one in eight.
    // The decoder's sign bits are one 128-byte line per row and 1 024 columns = eight column tiles; dealt round robin, the eight
    // tiles that share every line ran on eight different XCDs and each L2 fetched it (0.73 GB fetched by the data gradient for
    // 0.08 GB of bits, PMC round 6) and wrote its 16-byte piece of it separately.
    int pack = 0;
    __host__ __device__ int chunk() const { return (tilesN + nch - 1) / nch; }
    __host__ __device__ int groups() const { return bt > 0 ? (tilesM / bt) * nch : (splits > 1 ? splits : tilesN); }
    __host__ __device__ int group_size() const {
        return bt > 0 ? bt * chunk() : (splits > 1 ? tilesM * tilesN : tilesM);
    }
    __host__ __device__ unsigned grid() const {
        if (pack && bt == 0 && splits <= 1) return (unsigned)(64 * ((tilesN + 63) / 64) * tilesM);
        return (unsigned)(8 * ((groups() + 7) / 8) * group_size());
    }
    __device__ __forceinline__ bool decode(int bid, int& tile_m, int& tile_n, int& split) const {
        const int xcd = bid & 7, j = bid >> 3;
        if (bt > 0) {
            const int gs = bt * chunk();
            const int g = (j / gs) * 8 + xcd, r = j % gs;
            const int batch = g / nch, ch = g - batch * nch;
            split = 0;
            tile_n = ch * chunk() + r / bt;
            tile_m = batch * bt + r % bt;
            return g < groups() && tile_n < tilesN;
        }
        if (splits > 1) {
            const int T = tilesM * tilesN;
            const int tile = j % T;
            split = (j / T) * 8 + xcd;
            tile_m = tile / tilesN;
            tile_n = tile - tile_m * tilesN;
            return split < splits;
        }
        split = 0;
        tile_m = j % tilesM;
        if (pack) {                                      // workgroups 64 g .. 64 g + 63 (x tilesM): XCD x takes tiles 64 g + 8 x .. + 7
            const int jn = j / tilesM;
            tile_n = (jn >> 3) * 64 + xcd * 8 + (jn & 7);
            return tile_n < tilesN;
        }
        tile_n = (j / tilesM) * 8 + xcd;
        return tile_n < tilesN;
    }
};

// ------------------------------------------------------------------------------------------
// The kernel.  1-D grid in TileMap order (XCD-aware), covering tiles x split-K slices.
// With ws != nullptr the raw partial tile is written to ws[split][M][N]; the epilogue then runs in
// splitk_finalize_kernel (deterministic reduction order).
// ------------------------------------------------------------------------------------------
template <class AL, class BL>
static __global__ __launch_bounds__(GEMM_THREADS, 3)
void gemm_f32_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int kchunk, float* ws, TileMap tm) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * LDS_LD];   // [buf][A|B][BK][LDS_LD]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n, split;
    if (!tm.decode(blockIdx.x, tile_m, tile_n, split)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * kchunk;
    const int kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    al.begin(m0, tid, kbeg, kend);
    bl.begin(n0, tid, kbeg, kend);
    if (nk > 0) {
        al.fetch();
        bl.fetch();
        al.commit(lds);
        bl.commit(lds + BK * LDS_LD);
    }
    __syncthreads();

    const int arow = wm * 64 + (lane & 31);
    const int bcol = wn * 64 + (lane & 31);
    const int khalf = lane >> 5;

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < nk;
        if (more) {
            al.fetch();
            bl.fetch();
        }
        const float* as = lds + cur * (2 * BK * LDS_LD);
        const float* bs = as + BK * LDS_LD;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + khalf;
            const float a0 = as[kk * LDS_LD + arow];
            const float a1 = as[kk * LDS_LD + arow + 32];
            const float b0 = bs[kk * LDS_LD + bcol];
            const float b1 = bs[kk * LDS_LD + bcol + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            float* an = lds + (cur ^ 1) * (2 * BK * LDS_LD);
            al.commit(an);
            bl.commit(an + BK * LDS_LD);
        }
        __syncthreads();
    }

    tile_epilogue(acc, lds, ep, m0, M, n0 + (tid & 127), (n0 + (tid & 127)) < N, ws, split, N);
}


// ------------------------------------------------------------------------------------------
// LDS-DMA variant of the aligned fast path: the B operand X[k][n] (n contiguous) is staged with
// global_load_lds_dwordx4 (16 B per lane straight into LDS: no VGPR round trip, no ds_write, no address VALU in
// the loop) into an UNPADDED [BK][128] tile -- the DMA destination is wave-uniform base + lane*16, i.e. lane-linear,
// and the k-major unpadded tile is already conflict-free for the fragment reads (32 consecutive columns per
// half-wave).  One wave instruction moves 2 k-rows x 128 columns; 4 waves x 2 instructions cover the tile.
// The A operand keeps the register-staged loader (weights are k-contiguous and must be transposed on the way in).
// ------------------------------------------------------------------------------------------
template <class AL>
static __global__ __launch_bounds__(GEMM_THREADS, 4)
void gemm_f32_glds_kernel(AL al, const float* __restrict__ X, long ldx, Epilogue ep, int M, int N, int K, TileMap tm,
                          int vec_ep) {
    constexpr int AT = BK * LDS_LD, BT = BK * BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * (AT + BT) > 64 * EP_LD ? 2 * (AT + BT) : 64 * EP_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n, split_unused;
    if (!tm.decode(blockIdx.x, tile_m, tile_n, split_unused)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane DMA source: chunk c = i*4 + wave covers k rows 2c, 2c+1 of the tile; lane -> (row 2c + lane>>5, col (lane&31)*4)
    const float* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * 4 + wave;
        bsrc[i] = X + (long)(2 * c + (lane >> 5)) * ldx + n0 + (lane & 31) * 4;
    }
    auto dma_b = [&](float* Bs) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* dst = Bs + (i * 4 + wave) * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bsrc[i],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            bsrc[i] += (long)BK * ldx;
        }
    };
    al.begin(m0, tid, 0, K);
    if (nk > 0) {
        dma_b(lds + AT);
        al.fetch();
        al.commit(lds);
    }
    __syncthreads();
    const int arow = wm * 64 + (lane & 31);
    const int bcol = wn * 64 + (lane & 31);
    const int khalf = lane >> 5;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < nk;
        float* nxt = lds + (cur ^ 1) * (AT + BT);
        if (more) {
            dma_b(nxt + AT);
            al.fetch();
        }
        const float* as = lds + cur * (AT + BT);
        const float* bs = as + AT;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + khalf;
            const float a0 = as[kk * LDS_LD + arow];
            const float a1 = as[kk * LDS_LD + arow + 32];
            const float b0 = bs[kk * BN + bcol];
            const float b1 = bs[kk * BN + bcol + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) al.commit(nxt);
        __syncthreads();
    }
    if (vec_ep) tile_epilogue_v4(acc, lds, ep, m0, n0);
    else tile_epilogue(acc, lds, ep, m0, M, n0 + (tid & 127), (n0 + (tid & 127)) < N, nullptr, 0, N);
}


// Both operands by LDS-DMA: A(row x, red k) at At[k*lda + x] (x contiguous: W^T for forward, W itself for dgrad),
// B(k, n) at X[k*ldx + n].  No register staging at all: per k-step a wave issues 4 DMA instructions, reads 32
// fragments and issues 32 MFMAs.  Both LDS tiles are unpadded [BK][128].
static __global__ __launch_bounds__(GEMM_THREADS, 4)
void gemm_f32_glds2_kernel(const float* __restrict__ At, long lda, const float* __restrict__ X, long ldx, Epilogue ep,
                           int M, int N, int K, TileMap tm, int vec_ep) {
    constexpr int TT = BK * BN;                      // one operand tile (floats)
    __shared__ __attribute__((aligned(16))) float lds[4 * TT > 64 * EP_LD ? 4 * TT : 64 * EP_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n, split_unused;
    if (!tm.decode(blockIdx.x, tile_m, tile_n, split_unused)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* asrc[2];
    const float* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * 4 + wave;                  // chunk c = k rows 2c, 2c+1 of a tile
        asrc[i] = At + (long)(2 * c + (lane >> 5)) * lda + m0 + (lane & 31) * 4;
        bsrc[i] = X + (long)(2 * c + (lane >> 5)) * ldx + n0 + (lane & 31) * 4;
    }
    auto dma = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 4 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)asrc[i],
                                             (__attribute__((address_space(3))) void*)(buf + c * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bsrc[i],
                                             (__attribute__((address_space(3))) void*)(buf + TT + c * 256), 16, 0, 0);
            asrc[i] += (long)BK * lda;
            bsrc[i] += (long)BK * ldx;
        }
    };
    if (nk > 0) dma(lds);
    __syncthreads();
    const int arow = wm * 64 + (lane & 31);
    const int bcol = wn * 64 + (lane & 31);
    const int khalf = lane >> 5;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) dma(lds + (cur ^ 1) * (2 * TT));
        const float* as = lds + cur * (2 * TT);
        const float* bs = as + TT;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + khalf;
            const float a0 = as[kk * BN + arow];
            const float a1 = as[kk * BN + arow + 32];
            const float b0 = bs[kk * BN + bcol];
            const float b1 = bs[kk * BN + bcol + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    if (vec_ep) tile_epilogue_v4(acc, lds, ep, m0, n0);
    else tile_epilogue(acc, lds, ep, m0, M, n0 + (tid & 127), (n0 + (tid & 127)) < N, nullptr, 0, N);
}

// out[c][r] = in[r][c]   (small weight transposes feeding the DMA forward GEMM)
static __global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) t[j][threadIdx.x] = in[(long)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(long)c * rows + r] = t[threadIdx.x][j];
    }
}

// Deterministic reduction of the split-K slabs: a workgroup owns 64 consecutive outputs, its four thread rows sum
// every fourth slab (independent loads in flight, 256-byte coalesced rows) and the four partial sums are added in a
// fixed order.  (A thread per output walking all slabs is latency bound when there are hundreds of small slabs.)
static __global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* ws, int splits, int M, int N, Epilogue ep) {
    __shared__ float part[4][64];
    const long total = (long)M * N;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long i = base + lane;
        float s = 0.f;
        if (i < total) {
            int k = slice;
            for (; k + 12 < splits; k += 16) {
                const float a = ws[(long)k * total + i], b = ws[(long)(k + 4) * total + i];
                const float c = ws[(long)(k + 8) * total + i], d = ws[(long)(k + 12) * total + i];
                s += (a + b) + (c + d);
            }
            for (; k < splits; k += 4) s += ws[(long)k * total + i];
        }
        part[slice][lane] = s;
        __syncthreads();
        if (slice == 0 && i < total) {
            const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
            const int m = (int)(i / N), n = (int)(i - (long)m * N);
            ep.store(m, n, ep.prep(n), v);
        }
        __syncthreads();
    }
}

// Host-side launcher.  `splits_wanted` <= 1 means no split-K.  `ws_floats` is the capacity of ws.
template <class AL, class BL>
static hipError_t launch_gemm(AL al, BL bl, const Epilogue& ep, int M, int N, int K, int splits_wanted,
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
    int kchunk = cdiv(cdiv(K > 0 ? K : 1, splits), BK) * BK;
    splits = cdiv(K > 0 ? K : 1, kchunk);
    const TileMap tm{tilesM, tilesN, splits};
    float* wsp = splits > 1 ? ws : nullptr;
    hipLaunchKernelGGL((gemm_f32_kernel<AL, BL>), dim3(tm.grid()), dim3(GEMM_THREADS), 0, stream, al, bl, ep, M, N, K,
                       kchunk, wsp, tm);
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
