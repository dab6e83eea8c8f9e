// Platform probe, independent of libtvae_hip.so: does a plain kernel give bitwise-repeatable results while ANOTHER PROCESS
// uses the same GPU?  Build on the GPU box:  hipcc -O2 --offload-arch=gfx950 profiles/tools/cwsr_probe.hip -o /tmp/cwsr_probe
// Run two at once:  /tmp/cwsr_probe 400 & /tmp/cwsr_probe 400 & wait      (alone: /tmp/cwsr_probe 400)
// Every thread runs a long dependent chain of FMAs fed by per-lane LDS reads (mode 1) or by registers only (mode 0) and
// writes one float; the host compares every launch with the first one and histograms the deviating lanes (id & 63).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <unistd.h>

__global__ void chain(float* out, int iters, int mode) {
    __shared__ float2 tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = make_float2(1.0f + 1e-3f * threadIdx.x, 0.5f - 1e-3f * threadIdx.x);
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 1e-3f * (gid & 1023), b = 0.f;
    int ph = gid % 44;
    for (int i = 0; i < iters; ++i) {
        const float2 t = mode ? tab[ph] : make_float2(1.0f + 1e-3f * ph, 0.5f - 1e-3f * ph);
        a = fmaf(a, 0.999f, t.x * 1e-3f);
        b = fmaf(b, 0.998f, a * t.y);
        ph += 7;
        if (ph >= 44) ph -= 44;
    }
    out[gid] = a + b;
}

typedef float f2v __attribute__((ext_vector_type(2)));
// mode 2: the same chain on packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: what the compiler's SLP vectoriser
// emits for complex arithmetic, e.g. in dft_spectra_kernel), LDS-fed
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

int main(int argc, char** argv) {
    // back-to-back launches (no host synchronisation in between: both processes keep their queues full, so their
    // workgroups share CUs), each into its own slice of the output; compared with slice 0 at the end
    const int launches = argc > 1 ? atoi(argv[1]) : 2000, iters = argc > 2 ? atoi(argv[2]) : 3000;
    const int blocks = argc > 3 ? atoi(argv[3]) : 131, threads = 256, n = blocks * threads;
    float* d;
    if (hipMalloc(&d, (size_t)n * launches * sizeof(float)) != hipSuccess) return 1;
    std::vector<float> all((size_t)n * launches);
    for (int mode = 0; mode < 3; ++mode) {
        long hist[4] = {0, 0, 0, 0};
        int bad_launches = 0;
        for (int l = 0; l < launches; ++l) {
            if (mode == 2) hipLaunchKernelGGL(chain_pk, dim3(blocks), dim3(threads), 0, 0, d + (size_t)l * n, iters);
            else hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d + (size_t)l * n, iters, mode);
        }
        if (hipMemcpy(all.data(), d, all.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (int l = 1; l < launches; ++l) {
            int bad = 0;
            for (int i = 0; i < n; ++i)
                if (memcmp(&all[(size_t)l * n + i], &all[i], 4) != 0) { ++bad; ++hist[(i & 63) >> 4]; }
            if (bad) ++bad_launches;
        }
        printf("pid %d mode %d (%s): %d of %d launches deviate; deviating lanes by 16-lane group [0-15,16-31,32-47,48-63] = %ld %ld %ld %ld\n",
               (int)getpid(), mode, mode == 2 ? "packed fp32, LDS-fed" : (mode ? "LDS-fed" : "registers only"), bad_launches,
               launches - 1, hist[0], hist[1], hist[2], hist[3]);
    }
    return 0;
}
