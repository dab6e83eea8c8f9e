// Probe: are packed-fp32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) bitwise repeatable while ANOTHER
// kernel keeps the matrix pipe of the same SIMDs busy?  Round 2 twice saw sums kept in packed-FMA accumulators come out
// wrong in ONE 16-lane group (lanes 48-63) of kernels that mix MFMA phases with packed math, and round 3 saw the same
// signature in dft_spectra_kernel (packed complex arithmetic) whenever a second PROCESS shared the GPU.
//   hipcc -O2 --offload-arch=gfx950 profiles/tools/pk_mfma_probe.hip -o /tmp/pk_probe && /tmp/pk_probe [launches] [mode]
// One process, two streams: stream A runs `launches` back-to-back launches of a packed-fp32 dependent chain (or the same
// chain on scalar fp32 instructions: the control), stream B an endless train of MFMA-only kernels; grids are sized so that
// workgroups of both kernels share every CU.  Every launch of the chain is compared with the first one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void chain_pk(float* out, int iters) {
    __shared__ float2 tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = make_float2(1.0f + 1e-3f * threadIdx.x, 0.5f - 1e-3f * threadIdx.x);
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f2v a = {1e-3f * (gid & 1023), 2e-3f * (gid & 511)}, b = {0.f, 0.f};
    const f2v ca = {0.999f, 0.9985f}, cb = {0.998f, 0.9975f};
    int ph = gid % 44;
    for (int i = 0; i < iters; ++i) {
        const float2 t = tab[ph];
        const f2v tv = {t.x, t.y};
        a = __builtin_elementwise_fma(a, ca, tv * 1e-3f);
        b = __builtin_elementwise_fma(b, cb, a * tv);
        ph += 7;
        if (ph >= 44) ph -= 44;
    }
    out[gid] = a.x + a.y + b.x + b.y;
}
__global__ void chain_sc(float* out, int iters) {        // the same arithmetic, scalar fp32 instructions only
    __shared__ float2 tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = make_float2(1.0f + 1e-3f * threadIdx.x, 0.5f - 1e-3f * threadIdx.x);
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float ax = 1e-3f * (gid & 1023), ay = 2e-3f * (gid & 511), bx = 0.f, by = 0.f;
    int ph = gid % 44;
    for (int i = 0; i < iters; ++i) {
        const float2 t = tab[ph];
        ax = __builtin_fmaf(ax, 0.999f, t.x * 1e-3f);
        asm volatile("" : "+v"(ax));                     // (keeps the SLP vectoriser from pairing the two chains)
        ay = __builtin_fmaf(ay, 0.9985f, t.y * 1e-3f);
        bx = __builtin_fmaf(bx, 0.998f, ax * t.x);
        asm volatile("" : "+v"(bx));
        by = __builtin_fmaf(by, 0.9975f, ay * t.y);
        ph += 7;
        if (ph >= 44) ph -= 44;
    }
    out[gid] = ax + ay + bx + by;
}
__global__ void mfma_burn(float* sink, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.02f * (threadIdx.x - i)); }
    f32x16 acc0 = {0}, acc1 = {0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[3] == 12345.678f) sink[0] = acc0[1];
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 3000;
    const int blocks = 1024, threads = 256, n = blocks * threads, iters = 3000;
    float *d, *sink;
    if (hipMalloc(&d, (size_t)n * launches * sizeof(float)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    std::vector<float> all((size_t)n * launches);
    for (int mode = 0; mode < 4; ++mode) {                // 0: packed alone, 1: packed beside MFMA, 2: scalar beside MFMA, 3: packed beside MFMA again
        const bool burn = mode != 0, packed = mode != 2;
        for (int l = 0; l < launches; ++l) {
            if (packed) hipLaunchKernelGGL(chain_pk, dim3(blocks), dim3(threads), 0, sa, d + (size_t)l * n, iters);
            else hipLaunchKernelGGL(chain_sc, dim3(blocks), dim3(threads), 0, sa, d + (size_t)l * n, iters);
            if (burn) hipLaunchKernelGGL(mfma_burn, dim3(1024), dim3(256), 0, sb, sink, 4000);
        }
        hipStreamSynchronize(sa);
        hipStreamSynchronize(sb);
        if (hipMemcpy(all.data(), d, all.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        long hist[4] = {0, 0, 0, 0};
        int bad_launches = 0;
        for (int l = 1; l < launches; ++l) {
            int bad = 0;
            for (int i = 0; i < n; ++i)
                if (memcmp(&all[(size_t)l * n + i], &all[i], 4) != 0) { ++bad; ++hist[(i & 63) >> 4]; }
            if (bad) ++bad_launches;
        }
        printf("mode %d (%s fp32 chain%s): %d of %d launches deviate; deviating lanes by 16-lane group [0-15,16-31,32-47,48-63] = %ld %ld %ld %ld\n",
               mode, packed ? "PACKED" : "scalar", burn ? " beside an MFMA kernel on another stream" : ", alone", bad_launches,
               launches - 1, hist[0], hist[1], hist[2], hist[3]);
        fflush(stdout);
    }
    return 0;
}
