// Host-side helpers shared by the translation units of libtvae_hip.so (one .hip file per kernel family, compiled in
// parallel; every __global__ function in the kernel headers has internal linkage, so each unit carries exactly the
// kernels it launches).  The library keeps NO process-wide state: the arithmetic a call runs in is chosen by WHICH
// entry point the caller invokes (tvae_linear_fwd vs tvae_linear_fwd_x6, ...), never by a global mode.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/tvae_hip.h"
#include "gemm_f32_mfma.hpp"

#define TVAE_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

#define TVAE_INTERNAL __attribute__((visibility("hidden")))

namespace tvae {

static inline hipStream_t S(tvae_stream_t s) { return (hipStream_t)s; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
static inline int grid1d(long total, int block, int cap = 8192) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
template <class KernelT>
static hipError_t allow_big_lds(KernelT kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}
// number of split-K slices that brings a GEMM with `tiles` output tiles to ~4 workgroups per CU
static inline int pick_splits(int tiles, long K) {
    long want = (1024 + tiles - 1) / tiles;
    long maxs = K / (4 * BK);
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    return (int)want;
}
static inline int panels_of(long N, int width) { return (int)((N + width - 1) / width); }

static inline ConvGeom make_geom(int B, int Cin, int n, int ksz, int pad, int R) {
    ConvGeom g;
    g.B = B; g.Cin = Cin; g.n = n; g.ksz = ksz; g.pad = pad; g.R = R;
    g.Ho = n + 2 * pad - ksz + 1;
    g.P = g.Ho * g.Ho;
    g.K2 = ksz * ksz;
    return g;
}

static const size_t X6_LDS_MAX = 160 * 1024;       // whole LDS of a CU (one workgroup per CU by design)

}  // namespace tvae
