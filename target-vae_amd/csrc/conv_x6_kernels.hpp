// Lifting convolution (forward + weight gradient) on the bf16 matrix pipe with fp32-equivalent results:
// "x6" = every fp32 operand is written EXACTLY as the sum of three bf16 numbers  x = h + m + l  (8+8+8 significand
// bits, round-to-nearest residuals) and a product a*b is evaluated as the six partial products
//     ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh                      (fp32 accumulation inside the MFMA)
// dropping am*bl + al*bm + al*bl <= ~2^-23 |a*b|, i.e. below the rounding of an fp32 FMA chain (measured: 1.3e-7
// relative on K = 4096 dot products vs 3.0e-7 for a plain fp32 matmul; the 3-product variant is 4.4e-6).
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32 on gfx950, so six of them per product
// block are still 2.67x faster than the exact-fp32 pipe -- if the operands reach the MFMAs already split:
//   * filter bank / dY: split ONCE per step by a streaming pre-pass into fragment-ready 16-byte cells
//       cell(part, octet, row) = 8 consecutive-k bf16 of one row          [part][octet][row]  (row fastest)
//     so that a workgroup stages its A tile with global_load_lds_dwordx4 (64 consecutive cells per wave instruction,
//     lane-linear on both sides) and an MFMA A fragment is ONE conflict-free ds_read_b128 per part;
//   * image: split when it is loaded into LDS (once per workgroup).  The im2col operand needs, per lane, 8 consecutive
//     pixels of one padded-image row starting at an arbitrary pixel; a 16-byte LDS read must be 16-byte aligned
//     (misaligned b128 is replayed at 64 cycles), so every part is kept in TWO copies, the second shifted by one
//     pixel: a lane whose start pixel is odd reads the shifted copy, every read is then 4-byte aligned and a fragment
//     is two ds_read2_b32 per part.  The parity of the start pixel is a per-lane constant in both kernels (forward:
//     parity of the lane's output column; wgrad: parity of the lane's tap column) because the row pitch, the octet
//     starts and the position octets are all even.
// K ordering: forward  k = (ci, u, v) with every kernel row padded to a whole number of octets (zero weights);
//             wgrad    k = (image, h, w) with every output row padded to a whole number of octets (zero dY).
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_f32_mfma.hpp"
#include "conv_img_kernels.hpp"

namespace tvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Cell16 {                 // one fragment cell: 8 consecutive-k bf16 (or fp16) of one row (an MFMA operand register quad)
    uint4 u;
    bf16x8 v;
    f16x8 h;
    unsigned w[4];
};

constexpr int X6_STAGE_CELLS_FWD = 3 * 2 * 256;      // [part][octet half][256 rows]
constexpr int X6_STAGE_CELLS_WG = 3 * 2 * 128;       // [part][octet half][128 rows]
constexpr int X6_TAB_BYTES = 64;

static inline int x6_round_up(int v, int q) { return (v + q - 1) / q * q; }
// per-array element count of the LDS image: >= elems + 16 slack, and == 32 (mod 64), i.e. 16 dwords (mod 32): the
// 4-byte reads use 32 banks, and consecutive arrays -- hence also the two parity copies of one part, 3 arrays apart --
// then start 16 banks apart, so the even lanes (copy 0) and odd lanes (copy 1) of a fragment read never collide
static inline int x6_arr_elems(int elems) { return x6_round_up(elems + 16, 64) + 32; }

// exact three-way bf16 split of one fp32 value (RNE residuals); returns the raw bf16 bit patterns
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 bh = (__bf16)x;
    const float r1 = x - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}
// the same split for two values at once, results packed (x0 in the low half): v_cvt_pk_bf16_f32 rounds both, the parts
// are widened again with a shift / a mask and the residuals come from one packed subtraction -- 9 VALU instructions
// per pair instead of ~9 per value, bitwise the same parts as split3
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& hw, unsigned& mw, unsigned& lw) {
    const f32x2v x = {x0, x1};
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v));
    const f32x2v hf = {__uint_as_float(hw << 16), __uint_as_float(hw & 0xffff0000u)};
    const f32x2v r1 = x - hf;
    mw = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2v));
    const f32x2v mf = {__uint_as_float(mw << 16), __uint_as_float(mw & 0xffff0000u)};
    const f32x2v r2 = r1 - mf;
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2v));
}
__device__ __forceinline__ void split3x8(const float (&r)[8], Cell16& h, Cell16& m, Cell16& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) split3_pair(r[2 * q], r[2 * q + 1], h.w[q], m.w[q], l.w[q]);
}

// ------------------------------------------------------------------------------------------
// "h3" arithmetic (round 3): TWO fp16 parts per operand, THREE partial products.
// fp16 carries 11 significant bits (unit roundoff 2^-11), so with round-to-nearest parts  h = fp16(x), l = fp16(x - h)
//   |x - h| <= 2^-11 |x|,   |x - h - l| <= 2^-23 |x|   (as long as l is a normal fp16 number):
// x - h is an fp32 number of at most 13 significant bits, of which l keeps 11 -- two parts represent an fp32 value to ONE
// ulp (exactly, whenever x - h fits 11 bits, which is the common case).  h k + h k' + l k  leaves out  l l' <= 2^-22 |x y|.
// Worst case per product: 2^-23 + 2^-23 + 2^-22 = 2^-21 |x y| -- 8x the rounding of one fp32 FMA, but unbiased and not
// accumulating through the sum the way an FMA chain's own roundings do: products of fp16 numbers are exact in the fp32
// accumulator (22 bits).  What carries the accuracy claim is therefore the MEASUREMENT, not this bound: against fp64 the
// result is at least as accurate as the fp32 matrix pipe and as the six-product bf16 split for every distribution and
// reduction length probed (profiles/experiments/f16_split_probe.hip: 2.6e-7 vs 4.1e-7 (fp32 MFMA) vs 3.5e-7 (x6) at
// K = 512), with HALF the matrix instructions of x6.
// The price is fp16's 5-bit exponent: an operand is multiplied by a power of two (exact) that brings the largest
// magnitude of its scale group -- or an upper bound of it -- into [2^14, 2^15), and the accumulators by the inverse powers
// in the epilogue.  An element 2^j below the maximum of its group keeps min(23, 39 - j) significant bits (beyond j = 16
// the low part enters fp16's subnormal range, absolute error 2^-25 of the scaled value): 2^-16 -> exact, 2^-24 -> 3e-5,
// 2^-28 -> 5e-4 relative to ITSELF.  Round 3 used ONE group per operand tensor, which is fine normwise but lets a whole
// row of small values (a dead unit, a dim image) come out with few correct bits.  Round 4: the scale group is a ROW of the
// operand in the sense of the product (H3Scale, dense_x6_kernels.hpp) wherever its producer can supply row maxima -- a
// row / column of the OUTPUT then has the full two-part precision relative to its own magnitude, and what remains under
// one scale is the reduction index, where an element far below its row's maximum is negligible in the sum it enters
// (error <= 2^-39 of max_row |a| max_col |b| per term).  The remaining per-tensor scales and why they are safe are listed
// in DESIGN.md section 4 ("validity domain of h3"); tests/test_hip_primitives.py::test_h3_row_dynamic_range holds rows
// scaled by 2^-16 .. 2^-32 to 1e-5 per row against fp64.
// ------------------------------------------------------------------------------------------
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
// the power of two s with 2^14 <= s * amax < 2^15 (amax = 0, or absurdly small / large: clamped, s stays a normal number
// whose inverse is one too)
__device__ __forceinline__ float h3_scale(float amax) {
    int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);        // amax in [2^(e-127), 2^(e-126))
    if (e == 0) e = 127;                                         // zero (or denormal) maximum: scale 2^14
    int se = 268 - e;                                            // biased exponent of 2^(14 - (e - 127))
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float h3_inv(float s) {               // 1 / s for a power of two produced by h3_scale
    return __uint_as_float((254u - (__float_as_uint(s) >> 23)) << 23);
}
// both parts of two (already scaled) values at once, packed (x0 in the low half)
__device__ __forceinline__ void split2h_pair(float x0, float x1, unsigned& hw, unsigned& lw) {
    const f32x2v x = {x0, x1};
    const f16x2v h = __builtin_convertvector(x, f16x2v);
    hw = __builtin_bit_cast(unsigned, h);
    const f32x2v hf = __builtin_convertvector(h, f32x2v);
    const f32x2v r = x - hf;
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
__device__ __forceinline__ void split2hx8(const float (&r)[8], Cell16& h, Cell16& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) split2h_pair(r[2 * q], r[2 * q + 1], h.w[q], l.w[q]);
}
// three partial products; the small ones first is not needed (see mfma6)
__device__ __forceinline__ void mfma3h(f32x16& acc, const Cell16 (&a)[3], const Cell16 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0].h, b[0].h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0].h, b[1].h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1].h, b[0].h, acc, 0, 0, 0);
}
// max |x| into a device word by atomic max on the bit pattern (non-negative floats order like unsigned integers); NaN
// / Inf propagate as a huge maximum -> scale clamped, the result is then non-finite as it would be in any arithmetic
__device__ __forceinline__ void h3_atomic_amax(float* slot, float v) {
    const unsigned b = __float_as_uint(fabsf(v));
    // most callers arrive with less than what is already there: a plain read spares the L2 their atomic
    if (b > __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(slot))) atomicMax(reinterpret_cast<unsigned*>(slot), b);
}
__device__ __forceinline__ float h3_wave_max(float v) {          // in every lane
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Round 4 (h3 scales per row): the transforms along w leave one maximum per filter row / channel.  A wave's running
// maximum is flushed into the slot of a row when the wave moves on to another row (and at its end): wave reduction, then ONE
// atomic without return from lane 0.  (A first version flushed after every tile: 270 000 atomics on the 128 words = four
// cache lines of the channel maxima serialised in the memory system and DOUBLED the output transform, 0.83 -> 1.51 ms; the
// ring kernels now walk contiguous tile ranges, so a wave sees at most a handful of rows.)  In the ring kernels the atomic
// is NOT part of the hand-counted waits: a wait that does not know about it merely asks for one more of the oldest stores.
__device__ __forceinline__ void h3_tile_flush(float& mx, float* slot, int lane) {
    const float m = h3_wave_max(mx);
    if (lane == 0) __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(slot), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mx = 0.f;
}
// the same where the kernel's loads are the compiler's to count (generic / register-staged transforms, whose tiles go round
// robin): read first, most tiles bring nothing new
__device__ __forceinline__ void h3_tile_flush_rd(float& mx, float* slot, int lane) {
    const float m = h3_wave_max(mx);
    if (lane == 0) h3_atomic_amax(slot, m);
    mx = 0.f;
}

// block-wide maximum, then ONE atomic per workgroup (per-wave atomics on a single word serialise at the L2: 12 000 of
// them made a 5 us reduction take 140).  Every thread of the workgroup must call it.
__device__ __forceinline__ void h3_block_amax(float v, float* slot) {
    __shared__ float red_[16];
    v = h3_wave_max(v);
    __syncthreads();                                             // red_ may still be read by the previous call
    if ((threadIdx.x & 63) == 0) red_[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red_[0];
        for (int i = 1; i < (int)((blockDim.x + 63) >> 6); ++i) m = fmaxf(m, red_[i]);
        h3_atomic_amax(slot, m);
    }
}

// six partial products, in the order the A parts arrive from LDS (h, m, l): the first MFMA of a fragment then waits
// for ONE read, not three (the running sum already dwarfs every term, so the order is irrelevant for accuracy)
__device__ __forceinline__ void mfma6(f32x16& acc, const Cell16 (&a)[3], const Cell16 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// Pre-pass 1: filter bank fp32 [M][Cin*ksz*ksz] -> cells [part][octet o = (ci*ksz + u)*opr + vo][row m < Mpad]
// (v = 8*vo + j; zero beyond ksz, beyond M and in the octets o >= Cin*ksz*opr that pad the count to K8pad).
// ------------------------------------------------------------------------------------------
static __global__ void bank_split3_kernel(const float* __restrict__ bank, uint4* __restrict__ A3, int M, int Mpad, int Cin,
                                   int ksz, int opr, int K8pad) {
    const long total = (long)K8pad * Mpad;
    const int K = Cin * ksz * ksz;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % Mpad);
        const int o = (int)(i / Mpad);
        const int cu = o / opr, vo = o - cu * opr;
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = vo * 8 + j;
            r[j] = (m < M && cu < Cin * ksz && v < ksz) ? bank[(long)m * K + (long)cu * ksz + v] : 0.f;
        }
        Cell16 h, mm, l;
        split3x8(r, h, mm, l);
        A3[i] = h.u;
        A3[total + i] = mm.u;
        A3[2 * total + i] = l.u;
    }
}

// ------------------------------------------------------------------------------------------
// Pre-pass 2: dY fp32 feature-major [c][img][r][p = h*Ho + w] (row stride lddy) ->
// cells [part][img][q < QP][row m = c*R + r < Mpad] in the K order of conv1_wgrad_x6_kernel (X6WgK below):
//   q <  row_cells : h = q / opwf, 8 consecutive w = 8*(q % opwf) + j
//   q <  cells     : leftover column w = 8*opwf + (q - row_cells) / opc, 8 consecutive h = 8*((q - row_cells) % opc) + j
//   q >= cells     : zero (pads the count to an even number)
// Zero for h >= Ho, m >= M.  One workgroup per (img, c): coalesced reads of R*P floats, 16*R-byte write runs.
// ------------------------------------------------------------------------------------------
static __global__ void dy_split3_kernel(const float* __restrict__ dy, long lddy, uint4* __restrict__ D3, int B, int C, int R,
                                 int Ho, int opwf, int opc, int row_cells, int ncells, int QP, int Mpad) {
    // the R*P floats of (c, img) are contiguous: stage them through LDS with coalesced loads, gather cells from LDS
    extern __shared__ float tile[];
    const int img = blockIdx.x, c = blockIdx.y;
    const int P = Ho * Ho;
    const long part_stride = (long)B * QP * Mpad;
    if (c < C) {
        const float* src = dy + (long)c * lddy + (long)img * R * P;
        for (int i = threadIdx.x; i < R * P; i += blockDim.x) tile[i] = src[i];
    }
    __syncthreads();
    const int cells = QP * R;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
        const int r = i % R, q = i / R;
        float v[8];
        int h0, w0, step;                              // first position of the cell and its stride in p
        if (q < row_cells) {
            h0 = q / opwf;
            w0 = 8 * (q - h0 * opwf);
            step = 1;
        } else {
            const int qc = q - row_cells;
            const int wr = qc / opc;
            w0 = 8 * opwf + wr;
            h0 = 8 * (qc - wr * opc);
            step = Ho;
        }
        const bool ok = c < C && q < ncells;
        const int nvalid = ok ? (step == 1 ? 8 : min(8, Ho - h0)) : 0;
        const float* s0 = tile + r * P + h0 * Ho + w0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (j < nvalid) ? s0[j * step] : 0.f;
        Cell16 hh, mm, ll;
        split3x8(v, hh, mm, ll);
        const long at = ((long)img * QP + q) * Mpad + c * R + r;
        D3[at] = hh.u;
        D3[part_stride + at] = mm.u;
        D3[2 * part_stride + at] = ll.u;
    }
}

// rows [row0, row0+rows) of the zero-padded image, split, into the six LDS arrays (copy c, part p) at
// img + (c*3 + p)*arr:  copy0[e] = x[e], copy1[e] = x[e+1].  With imgT != nullptr the padded columns
// [tcol0, tcol0 + nct) are ALSO stored transposed (element (ci*nct + col - tcol0)*PT + row) in six more arrays at
// imgT + (c*3 + p)*arrT (imgT must directly follow the six image arrays: one zero fill covers both).
// Everything is zeroed with 16-byte stores first; then only the pixels that exist are fetched, eight global loads in
// flight per thread before the first is consumed (with one workgroup per CU nothing else hides that latency).
// Ends with a barrier.
__device__ __forceinline__ void load_split_image(unsigned short* img, int arr, const float* __restrict__ y, int b,
                                                 const ConvGeom& g, int row0, int rows, int Wp,
                                                 unsigned short* imgT = nullptr, int arrT = 0, int PT = 0,
                                                 int tcol0 = 0, int nct = 0) {
    uint4* z = reinterpret_cast<uint4*>(img);
    const int n16 = (6 * (arr + (imgT ? arrT : 0)) * 2) >> 4;      // arr, arrT are multiples of 32 elements
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int iy0 = max(0, row0 - g.pad), iy1 = min(g.n, row0 + rows - g.pad);
    const int vrows = max(0, iy1 - iy0);
    const int per_c = vrows * g.n;
    const int count = g.Cin * per_c;
    const float* yb = y + (long)b * g.Cin * g.n * g.n;
    for (int base = 0; base < count; base += 8 * (int)blockDim.x) {
        float v[8];
        int at[8], att[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * (int)blockDim.x + (int)threadIdx.x;
            at[u] = -1;
            att[u] = -1;
            v[u] = 0.f;
            if (idx < count) {
                const int ci = idx / per_c;
                const int rem = idx - ci * per_c;
                const int ry = rem / g.n, ix = rem - ry * g.n;
                const int iy = iy0 + ry;
                v[u] = yb[((long)ci * g.n + iy) * g.n + ix];
                const int rl = iy + g.pad - row0, cp = ix + g.pad;
                at[u] = (ci * rows + rl) * Wp + cp;
                if (imgT && cp >= tcol0 && cp < tcol0 + nct) att[u] = (ci * nct + cp - tcol0) * PT + rl;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (at[u] >= 0) {
                unsigned short h, m, l;
                split3(v[u], h, m, l);
                const int e = at[u];
                img[0 * arr + e] = h;
                img[1 * arr + e] = m;
                img[2 * arr + e] = l;
                if (e > 0) {
                    img[3 * arr + e - 1] = h;
                    img[4 * arr + e - 1] = m;
                    img[5 * arr + e - 1] = l;
                }
                const int et = att[u];
                if (et >= 0) {
                    imgT[0 * arrT + et] = h;
                    imgT[1 * arrT + et] = m;
                    imgT[2 * arrT + et] = l;
                    if (et > 0) {
                        imgT[3 * arrT + et - 1] = h;
                        imgT[4 * arrT + et - 1] = m;
                        imgT[5 * arrT + et - 1] = l;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// one B fragment (3 parts) from the LDS image: 4 consecutive dwords per part at dword index `at`
__device__ __forceinline__ void read_b_frag(const unsigned* __restrict__ imgdw, int at, int part_stride_dw,
                                            Cell16 (&b)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const unsigned* q = imgdw + at + p * part_stride_dw;
        b[p].w[0] = q[0];
        b[p].w[1] = q[1];
        b[p].w[2] = q[2];
        b[p].w[3] = q[3];
    }
}

// ------------------------------------------------------------------------------------------
// Forward.  Tile 256 (rows c*R+r) x 128 (positions of ONE image); four waves STACKED along the rows: wave w owns rows
// [64w, 64w+64) x all 128 positions = 2 x 4 MFMA tiles, 48 MFMAs per k-step (2 octets = 16 taps of one kernel row).
// Because a wave's A rows are read by nobody else, each wave runs a PRIVATE A pipeline: it DMAs its own 64 rows of
// the split bank (6 global_load_lds per step) into its own three-slot ring, two steps ahead, waits on its own vmcnt,
// and never meets a barrier inside the k-loop -- the four waves drift apart, so their LDS bursts, DMA waits and
// bookkeeping overlap each other's MFMAs (with one workgroup per CU nothing else would).  The B operand is the
// static LDS image.  Measured steps of the barrier version: 2300 cycles for 1536 cycles of MFMA (s_memtime).
// The global_load_lds are issued through inline asm so that the compiler does NOT track them: it would put
// s_waitcnt vmcnt(0) in front of every later LDS read (it cannot tell the ring slots apart) and expose the full
// L2 / Infinity-Cache latency in every step.  The waits are placed by hand (vmcnt(6) = all but the youngest stage
// have landed); compiler-generated vmcnt waits elsewhere can only become stricter.
// LDS: [A rings 4 waves x 3 slots x 6 KiB][bias table 1 KiB][6 image arrays].
// ------------------------------------------------------------------------------------------
constexpr int X6_FWD_SLOT_CELLS = 3 * 2 * 64;        // one wave's stage: [part][octet half][64 rows]
constexpr int X6_FWD_RING_BYTES = 4 * 3 * X6_FWD_SLOT_CELLS * 16;
constexpr int X6_FWD_BIAS_BYTES = 1024;

static __global__ __launch_bounds__(GEMM_THREADS, 1)
void conv1_fwd_x6_kernel(const uint4* __restrict__ A3, const float* __restrict__ y, ConvGeom g, Epilogue ep, int M,
                         int Mpad, int K8pad, int opr, int tilesPerImg, int rows, int Wp, int arr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* bsm = reinterpret_cast<float*>(smem_raw + X6_FWD_RING_BYTES);
    unsigned short* img = reinterpret_cast<unsigned short*>(smem_raw + X6_FWD_RING_BYTES + X6_FWD_BIAS_BYTES);
    const unsigned* imgdw = reinterpret_cast<const unsigned*>(img);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint4* ring = reinterpret_cast<const uint4*>(smem_raw) + wave * 3 * X6_FWD_SLOT_CELLS;
    const int per_m = g.B * tilesPerImg;
    const int tile_m = blockIdx.x / per_m;
    const int rest = blockIdx.x - tile_m * per_m;
    const int b = rest / tilesPerImg;
    const int p0 = (rest - b * tilesPerImg) * BN;
    const int m0 = tile_m * 256;
    const int hmin = p0 / g.Ho;
    // zero skipping (single channel): kernel rows whose image rows are pure padding for every output row of the tile
    int obeg = 0, oend = K8pad;
    if (g.Cin == 1) {
        const int plast = min(g.P - 1, p0 + BN - 1);
        const int hmax = plast / g.Ho;
        const int ulo = max(0, g.pad - hmax);
        const int uhi = min(g.ksz - 1, g.pad + g.n - 1 - hmin);
        obeg = (ulo * opr) & ~1;
        oend = min(K8pad, ((uhi + 1) * opr + 1) & ~1);
        if (oend < obeg) oend = obeg;
    }
    const int nk = (oend - obeg) >> 1;
    bsm[tid] = (ep.bias && (m0 + tid) < M) ? ep.bias[(m0 + tid) >> ep.bias_shift] : 0.f;

    load_split_image(img, arr, y, b, g, hmin, rows, Wp);      // ends with a barrier (also publishes bsm)

    const int psd = arr >> 1;                          // part stride in dwords
    int bdw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int p = p0 + j * 32 + (lane & 31);
        if (p >= g.P) p = g.P - 1;                     // padded columns read valid data; never stored
        const int h = p / g.Ho, w = p - h * g.Ho;
        const int boff = (h - hmin) * Wp + w;
        const int c = boff & 1;
        bdw[j] = ((c * 3) * arr + boff - c) >> 1;
    }
    // this lane half's octet o = obeg + 2t + khalf = (ci*ksz + u)*opr + vo, walked incrementally in registers (no
    // divisions in the loop):  dword offset ((ci*rows + u)*Wp + 8*vo) / 2, 0 for the padding octets
    const int khalf = lane >> 5;
    int o_ci, o_u, o_vo;
    {
        const int oo = obeg + khalf;
        const int cu = oo / opr;
        o_vo = oo - cu * opr;
        o_ci = cu / g.ksz;
        o_u = cu - o_ci * g.ksz;
    }
    auto tap_dw = [&]() -> int { return (o_ci < g.Cin) ? (((o_ci * rows + o_u) * Wp + 8 * o_vo) >> 1) : 0; };
    auto tap_next = [&]() {                            // branch-free: +2 octets wrap at most twice (opr >= 1)
        o_vo += 2;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const bool wrap = o_vo >= opr;
            o_vo -= wrap ? opr : 0;
            o_u += wrap ? 1 : 0;
            const bool wrap_u = o_u == g.ksz;
            o_u = wrap_u ? 0 : o_u;
            o_ci += wrap_u ? 1 : 0;
        }
    };
    // the wave's DMA: rows [64w, 64w+64) of the six (part, octet half) cell rows of octets o, o+1 -> ring slot
    const long part_cells = (long)K8pad * Mpad;
    const uint4* a_src = A3 + m0 + 64 * wave + lane;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem_raw +
                              (unsigned)(wave * 3 * X6_FWD_SLOT_CELLS * 16);
    auto dma_a = [&](int slot, int o) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int oh = 0; oh < 2; ++oh) {
                const uint4* src = a_src + p * part_cells + (long)(o + oh) * Mpad;
                const unsigned dst = ring_lds + (unsigned)((slot * 6 + p * 2 + oh) * 64 * 16);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             :: "s"(dst), "v"(src) : "memory", "m0");
            }
    };
    auto read_a = [&](int slot, Cell16 (&a)[2][3]) {
        const uint4* as = ring + slot * X6_FWD_SLOT_CELLS + khalf * 64 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[i][p].u = as[p * 128 + i * 32];
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int olast = K8pad - 2;
    Cell16 bf[4][3];
    {
        const int kt = tap_dw();
        tap_next();
#pragma unroll
        for (int j = 0; j < 4; ++j) read_b_frag(imgdw, bdw[j] + kt, psd, bf[j]);
        dma_a(0, min(obeg, olast));
        dma_a(1, min(obeg + 2, olast));
    }
    int s_cur = 0, s_dma = 2;
    for (int t = 0; t < nk; ++t) {
        // stage t (issued two steps ago) has landed once at most the 6 DMAs of stage t+1 are still in flight
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        Cell16 af[2][3], bn[4][3];
        read_a(s_cur, af);
        // bookkeeping of the coming steps: B fragments of step t+1 from the static image, DMA of stage t+2 into the
        // slot step t-1 used
        const int ktn = tap_dw();
        tap_next();
#pragma unroll
        for (int j = 0; j < 4; ++j) read_b_frag(imgdw, bdw[j] + ktn, psd, bn[j]);
        dma_a(s_dma, min(obeg + 2 * (t + 2), olast));   // unconditional (clamped): uniform vmcnt bookkeeping
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma6(acc[i][j], af[i], bf[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = bn[j][p];
        // Issue order (one wave per SIMD issues in order, and an MFMA occupies the pipe for 32 cycles): the 6 A reads,
        // then every MFMA followed by at most one LDS read and two ALU instructions, so that the 24 B reads and the
        // address / DMA bookkeeping sit in the shadows of the MFMAs instead of in front of them (a wave can only have
        // 15 LDS operations outstanding: a burst of 24 reads stalls the in-order issue, MFMAs included).
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        s_cur = (s_cur == 2) ? 0 : s_cur + 1;
        s_dma = (s_dma == 2) ? 0 : s_dma + 1;
        __builtin_amdgcn_sched_barrier(0);
    }
    // Direct epilogue (nothing overlaps it with one workgroup per CU, so it must be short): the accumulator layout
    // already gives 32 consecutive positions (128 contiguous bytes) per row and instruction; no LDS staging.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail DMAs still target this wave's ring
    int pc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pc[j] = p0 + j * 32 + (lane & 31);
    const long imgoff = (long)b * g.R * g.P;
    const int rmask = (1 << ep.conv_shift) - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = m0 + row;
            const float bv = bsm[row];
            float* crow = ep.C + (long)(m >> ep.conv_shift) * ep.ldc + imgoff + (long)(m & rmask) * g.P;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[i][j][r] + bv;
                if (ep.act == ACT_LRELU) v = v > 0.f ? v : v * ep.slope;
                else if (ep.act == ACT_TANH) v = tanhf(v);
                if (m < M && pc[j] < g.P) crow[pc[j]] = v;
            }
        }
}


// ------------------------------------------------------------------------------------------
// Weight gradient.  Tile 128 (rows) x 256 (taps), 2x2 waves, wave 64 x 128 = 2 x 4 MFMA tiles; k-step = 2 position
// octets of one image; split over images as in the fp32 kernel (slab partials + deterministic finalize).
// 1-D grid, XCD-aware: the tilesN tap tiles of one (row tile, image slice) group share their dY cells in one L2.
//
// K order of one image (X6WgK): the Ho x Ho output positions are covered WITHOUT padding every row to whole octets:
//   row cells   q <  Ho*opwf           : 8 consecutive w of one output row h (w < 8*opwf = the whole octets of a row);
//   column cells q >= Ho*opwf          : the rem = Ho - 8*opwf leftover columns, 8 consecutive h of one column w.
// (Ho = 33: 132 + 5 cells instead of 165, i.e. 0.6 % instead of 21 % of the MFMAs multiply padding.)  A column cell
// needs 8 vertically adjacent pixels of the padded image per lane, so the columns that the leftover positions can
// touch (8*opwf .. 8*opwf+rem-1+ksz-1) are ALSO kept transposed in LDS, again in two parity copies; the parity of a
// column-cell start is the parity of the lane's tap row u (the transposed pitch and the cell starts are even).
// The per-step cell ids and image offsets come from a small LDS table built once per workgroup.
// LDS: [A stages 2 x 12 KiB][cell table][6 image arrays][6 transposed arrays].
// ------------------------------------------------------------------------------------------
struct X6WgK {
    int opwf;      // whole octets per output row
    int rem;       // leftover columns per row (0..7)
    int opc;       // octets per leftover column = ceil(Ho / 8)
    int row_cells; // Ho * opwf
    int cells;     // row_cells + rem * opc
    int QP;        // cells + 1 zero cell, rounded up to even (cell index `cells` is all zero)
};
static inline X6WgK x6_wg_k(int Ho) {
    X6WgK k;
    k.opwf = Ho / 8;
    k.rem = Ho - 8 * k.opwf;
    k.opc = (Ho + 7) / 8;
    k.row_cells = Ho * k.opwf;
    k.cells = k.row_cells + k.rem * k.opc;
    k.QP = (k.cells + 2) & ~1;
    return k;
}
constexpr int X6_WG_TAB_INTS = 2 * 256;            // (cell id, packed offset) per octet of the tile's k range, <= 256

static __global__ __launch_bounds__(GEMM_THREADS, 1)
void conv1_wgrad_x6_kernel(const uint4* __restrict__ D3, const float* __restrict__ y, ConvGeom g, int M, int Mpad,
                           int N, X6WgK kk, int imgs_per_split, float* ws, int tilesN, int rows, int Wp, int arr,
                           int PT, int arrT, int nsplits, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* As = reinterpret_cast<uint4*>(smem_raw);
    int* tab = reinterpret_cast<int*>(smem_raw + 2 * X6_STAGE_CELLS_WG * 16);
    unsigned short* img = reinterpret_cast<unsigned short*>(smem_raw + 2 * X6_STAGE_CELLS_WG * 16 + X6_WG_TAB_INTS * 4);
    unsigned short* imgT = img + 6 * arr;
    const unsigned* imgdw = reinterpret_cast<const unsigned*>(img);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    const int m0 = tile_m * 128, n0 = tile_n * 256;
    const int ib = split * imgs_per_split;
    const int ie = min(g.B, ib + imgs_per_split);
    const int Hp = g.n + 2 * g.pad;
    // kept image rows and zero skipping exactly as in conv1_wgrad_img_kernel (single channel)
    const int ulo = (rows == Hp) ? 0 : (n0 / g.ksz);
    int hlo = 0, hhi = g.Ho - 1;
    if (g.Cin == 1) {
        const int ua = n0 / g.ksz, ub = min(N - 1, n0 + 255) / g.ksz;
        hlo = max(0, g.pad - ub);
        hhi = min(g.Ho - 1, g.pad + g.n - 1 - ua);
    }
    // octet sequence of this tile: row cells of rows hlo..hhi, then column cells covering hlo..hhi of each leftover
    // column; padded to an even count with the all-zero cell.  tab[2s] = cell id, tab[2s+1] = dword offset | phase<<24
    const int nh = max(0, hhi - hlo + 1);
    const int n_row = nh * kk.opwf;
    const int ho_lo = hlo >> 3, ho_hi = hhi >> 3;
    const int n_colh = (nh > 0) ? (ho_hi - ho_lo + 1) : 0;
    const int n_oct = n_row + kk.rem * n_colh;
    const int nk = (n_oct + 1) >> 1;
    const int c0 = 8 * kk.opwf;                        // first leftover column (in output coordinates)
    for (int s = tid; s < 2 * nk; s += GEMM_THREADS) {
        int cell = kk.cells, packed = 0;               // padding octet: zero dY cell, any valid image offset
        if (s < n_row) {
            const int hr = s / kk.opwf, wo = s - hr * kk.opwf;
            const int h = hlo + hr;
            cell = h * kk.opwf + wo;
            packed = (h * Wp + 8 * wo) >> 1;
        } else if (s < n_oct) {
            const int sc = s - n_row;
            const int wr = sc / n_colh, ho = ho_lo + (sc - wr * n_colh);
            cell = kk.row_cells + wr * kk.opc + ho;
            packed = ((wr * PT + 8 * ho) >> 1) | (1 << 24);
        }
        tab[2 * s] = cell;
        tab[2 * s + 1] = packed;
    }

    const int psd = arr >> 1, psdT = arrT >> 1;
    const int tbase = (6 * arr) >> 1;                  // dword index of the transposed arrays
    int ndw[4], ndwT[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int nn = n0 + wn * 128 + j * 32 + (lane & 31);
        if (nn >= N) nn = N - 1;
        const int ci = nn / g.K2, rem = nn - ci * g.K2;
        const int u = rem / g.ksz, v = rem - u * g.ksz;
        const int noff = (ci * rows + (u - ulo)) * Wp + v;
        const int c = noff & 1;
        ndw[j] = ((c * 3) * arr + noff - c) >> 1;
        // transposed: element (column v [+ leftover index wr*PT from the table], row u - ulo [+ 8*ho + j])
        const int toff = (ci * (g.ksz + 7) + v) * PT + (u - ulo);
        const int ct = toff & 1;
        ndwT[j] = tbase + (((ct * 3) * arrT + toff - ct) >> 1);
    }
    const int khalf = lane >> 5;
    const long part_cells = (long)g.B * kk.QP * Mpad;
    auto dma_a = [&](uint4* stage, int b, int s2) {
        // 6 cell rows of 128 cells = 12 instructions of 64 cells: wave w takes (part, octet half, 64-row half) ids w, w+4, w+8
        const int q0 = __builtin_amdgcn_readfirstlane(tab[2 * s2]);
        const int q1 = __builtin_amdgcn_readfirstlane(tab[2 * s2 + 2]);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int id = wave + 4 * s;               // 0..11
            const int p = id >> 2, oh = (id >> 1) & 1, mh = id & 1;
            const uint4* src = D3 + p * part_cells + ((long)b * kk.QP + (oh ? q1 : q0)) * Mpad + m0 + 64 * mh + lane;
            uint4* dst = stage + (p * 2 + oh) * 128 + 64 * mh;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // B fragments of octet s (this lane half): row cells from the image copies, column cells from the transposed ones
    auto read_b_all = [&](int s, Cell16 (&bf)[4][3]) {
        const int packed = tab[2 * s + 1];
        const bool col = (packed >> 24) != 0;
        const int off = packed & 0xffffff;
        const int ps = col ? psdT : psd;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) read_b_frag(imgdw, (col ? ndwT[jj] : ndw[jj]) + off, ps, bf[jj]);
    };

    f32x16 acc[2][2][2];                                // [tap half hn][row tile i][tap tile j within the half]
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[hn][i][j][r] = 0.f;

    const int arow = wm * 64 + (lane & 31);
    for (int b = ib; b < ie; ++b) {
        __syncthreads();                                // previous image fully consumed (and the table is visible)
        load_split_image(img, arr, y, b, g, ulo, rows, Wp, imgT, arrT, PT, c0, g.ksz + 7);
        Cell16 bf[4][3];
        if (nk > 0) {
            dma_a(As, b, 0);
            read_b_all(khalf, bf);
        }
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            const bool more = (t + 1) < nk;
            // all LDS reads of this step first, then the DMA of the next stage (see conv1_fwd_x6_kernel)
            const uint4* as = As + cur * X6_STAGE_CELLS_WG + khalf * 128 + arow;
            Cell16 af[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[i][p].u = as[p * 256 + i * 32];
            Cell16 bn[4][3];
            read_b_all(more ? 2 * (t + 1) + khalf : khalf, bn);
            if (more) dma_a(As + (cur ^ 1) * X6_STAGE_CELLS_WG, b, 2 * (t + 1));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                mfma6(acc[jj >> 1][0][jj & 1], af[0], bf[jj]);
                mfma6(acc[jj >> 1][1][jj & 1], af[1], bf[jj]);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[jj][p] = bn[jj][p];
            __builtin_amdgcn_sched_barrier(0);          // see conv1_fwd_x6_kernel
            __syncthreads();
        }
    }
    __syncthreads();
    // wave (wm, wn) holds rows wm*64 + i*32 and taps wn*128 + hn*64 + j*32: stage each 128-tap half separately
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        float* ct = reinterpret_cast<float*>(smem_raw);
        const int ecol = tid & 127;
        const int n = n0 + half * 128 + ecol;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i) __syncthreads();
            if (wn == half) {
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            ct[rl * 128 + hn * 64 + j * 32 + (lane & 31)] = acc[hn][i][j][r];
                        }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int rl = (tid >> 7) + 2 * it;
                const int m = m0 + (rl >> 5) * 64 + i * 32 + (rl & 31);
                if (n < N && m < M) ws[((long)split * M + m) * N + n] = ct[rl * 128 + ecol];
            }
        }
    }
}


}  // namespace tvae
