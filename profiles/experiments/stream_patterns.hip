// Streaming-read patterns over a feature-major [128][N] fp32 tensor (N = 2 230 272: the encoder tail at the headline
// batch), one persistent 512-thread workgroup per CU limited to 2 waves / SIMD by a 100 KB LDS allocation -- the
// occupancy of the fused encoder-tail kernels.  Question: does the bytes-per-row-segment of a load instruction
// (128 B / 256 B / 1 KB) decide the achieved HBM rate at this occupancy?
//   hipcc -O3 --offload-arch=gfx950 stream_patterns.hip -o stream_patterns && ./stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROWS = 128;

// P0: B-fragment order, dword per lane: lanes 0-31 -> row r, 32 columns; lanes 32-63 -> row r + 8 (128 B per row segment)
__global__ __launch_bounds__(512, 2) void p0(const float* __restrict__ X, long ld, long N, float* out, int depth) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 31, kh = lane >> 5;
    const long nch = N / 64, gw = blockIdx.x * 8 + wave, gs = gridDim.x * 8;
    float acc = 0.f;
    for (long c = gw; c < nch; c += gs) {
        const float* p = X + c * 64 + nl + (long)(8 * kh) * ld;
#pragma unroll 2
        for (int t = 0; t < 8; ++t) {
            float v[16];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[jt * 8 + j] = p[(long)(16 * t + j) * ld + 32 * jt];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j];
        }
    }
    if (acc == 123.456f) out[0] = acc + lds[threadIdx.x];
}
// P1: dwordx4 per lane, 16 lanes per row: 4 rows x 256 B per instruction
__global__ __launch_bounds__(512, 2) void p1(const float* __restrict__ X, long ld, long N, float* out, int depth) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nch = N / 64, gw = blockIdx.x * 8 + wave, gs = gridDim.x * 8;
    float acc = 0.f;
    for (long c = gw; c < nch; c += gs) {
        const float* p = X + c * 64 + (lane & 15) * 4 + (long)(lane >> 4) * ld;
#pragma unroll 2
        for (int t = 0; t < 8; ++t) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(p + (long)(16 * t + 4 * q) * ld);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
        }
    }
    if (acc == 123.456f) out[0] = acc + lds[threadIdx.x];
}
// P2: dwordx4 per lane, 64 lanes on ONE row: 1 KB contiguous per instruction (a workgroup tile of 512 columns: wave w
// takes rows w, w + 8, ...; two instructions per row)
__global__ __launch_bounds__(512, 2) void p2(const float* __restrict__ X, long ld, long N, float* out, int depth) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nch = N / 512;
    float acc = 0.f;
    for (long c = blockIdx.x; c < nch; c += gridDim.x) {
        const float* p = X + c * 512 + lane * 4 + (long)wave * ld;
#pragma unroll 2
        for (int t = 0; t < 8; ++t) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[2 * q + h] = *reinterpret_cast<const float4*>(p + (long)(16 * t + 8 * q) * ld + 256 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
        }
    }
    if (acc == 123.456f) out[0] = acc + lds[threadIdx.x];
}
// P3: like P0 but the 128 rows are read as 64 row-PAIRS of one contiguous copy laid out [N/64][128][64] (chunk-major):
// what the pattern would cost if the tensor were stored tile-major (same dword loads, no page striding)
__global__ __launch_bounds__(512, 2) void p3(const float* __restrict__ X, long ld, long N, float* out, int depth) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 31, kh = lane >> 5;
    const long nch = N / 64, gw = blockIdx.x * 8 + wave, gs = gridDim.x * 8;
    float acc = 0.f;
    for (long c = gw; c < nch; c += gs) {
        const float* p = X + c * 64 * ROWS + nl + (long)(8 * kh) * 64;
#pragma unroll 2
        for (int t = 0; t < 8; ++t) {
            float v[16];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[jt * 8 + j] = p[(long)(16 * t + j) * 64 + 32 * jt];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j];
        }
    }
    if (acc == 123.456f) out[0] = acc + lds[threadIdx.x];
}

template <class K>
void run(const char* name, K kern, const float* X, long N, float* out, size_t lds, int grid) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, X, N, N, out, 0);
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, X, N, N, out, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    printf("%-44s grid %4d lds %6zu: %7.3f ms  %7.1f GB/s\n", name, grid, lds, ms, ROWS * N * 4.0 / ms / 1e6);
}

int main() {
    const long N = 2230272;
    float *X, *out;
    hipMalloc(&X, sizeof(float) * ROWS * N);
    hipMalloc(&out, 1024);
    hipMemset(X, 0, sizeof(float) * ROWS * N);
    for (size_t lds : {(size_t)100 * 1024, (size_t)40 * 1024, (size_t)1024}) {
        const int grid = lds > 80 * 1024 ? 256 : (lds > 20 * 1024 ? 512 : 1024);
        run("P0 dword, 2 rows x 128 B per instr", p0, X, N, out, lds, grid);
        run("P1 dwordx4, 4 rows x 256 B per instr", p1, X, N, out, lds, grid);
        run("P2 dwordx4, 1 row x 1 KB per instr", p2, X, N, out, lds, grid);
        run("P3 dword, tile-major copy (no row stride)", p3, X, N, out, lds, grid);
    }
    return 0;
}
