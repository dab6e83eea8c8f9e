// Ablation of enc_tail_fwd_x6_kernel at the headline shape (N = 2 230 272): which part bounds it?
//   for A in 0 1 2 4 8 3 12; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DET_ABL=$A enc_tail_ablate.hip -o enc_tail_ablate_$A; done
// ET_ABL bits: 1 no H stores, 2 no head FMAs, 4 no MFMAs, 8 no operand reloads (first chunk's registers reused)
#include "../../target-vae_amd/csrc/abi_dense_x6.hpp"
#include "../../target-vae_amd/csrc/enc_tail_x6_kernels.hpp"
#include <cstdio>
using namespace tvae;
int main() {
    const long N = 2230272;
    const int Rpad = 512;
    float *X, *H, *heads, *b2, *Wh, *bh;
    uint4* W3;
    hipMalloc(&X, 4 * 128 * N); hipMalloc(&H, 4 * 128 * N); hipMalloc(&heads, 4 * 8 * N);
    hipMalloc(&b2, 512); hipMalloc(&Wh, 4 * 7 * 128); hipMalloc(&bh, 64); hipMalloc(&W3, 3 * 16 * Rpad * 16);
    hipMemset(X, 0x3c, 4 * 128 * N); hipMemset(W3, 0x3c, 3 * 16 * Rpad * 16); hipMemset(b2, 0, 512);
    hipMemset(Wh, 0x3c, 4 * 7 * 128); hipMemset(bh, 0, 64);
    const size_t lds = 3 * 16 * 128 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tail_fwd_x6_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 13; ++r) {
        if (r == 3) hipEventRecord(a);
        hipLaunchKernelGGL((enc_tail_fwd_x6_kernel<3>), dim3(256), dim3(ET_THREADS), lds, 0, W3, Rpad, X, N, b2, Wh, bh, 7, H, N, heads, N, N, 1, 0.01f);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("ET_ABL=%d waves %d depth %d: %.3f ms\n", ET_ABL, ET_WAVES, ET_DEPTH, ms / 10);
    return 0;
}
