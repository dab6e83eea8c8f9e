// libtvae_hip.so: the gradient all-reduce of the data-parallel step as a C-ABI entry point (tvae_allreduce_flat).
// RCCL is resolved at RUN time (dlopen / dlsym): the library carries no link dependency on it, and inside a PyTorch process the
// handle is the RCCL instance torch has already loaded (RTLD_NOLOAD first) -- two copies of RCCL in one process would each keep
// their own communicators and proxy threads.  The caller owns the communicator; nothing here is process-wide except the handle.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include "abi_common.hpp"

namespace {
// the part of rccl.h this unit needs (ABI-stable since NCCL 2.x): opaque communicator, 128-byte unique id, enums by value
struct UniqueId { char internal[128]; };
using Comm = void*;
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;
using GetUniqueIdFn = int (*)(UniqueId*);
using CommInitRankFn = int (*)(Comm*, int, UniqueId, int);
using AllReduceFn = int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t);
using CommDestroyFn = int (*)(Comm);

struct Rccl {
    void* h = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn destroy = nullptr;
    bool ok = false;
    bool preloaded = false;      // the handle is the instance that was already in the process (RTLD_NOLOAD hit)
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {                    // the instance already in the process (torch's) ...
            r.h = dlopen(n, RTLD_LAZY | RTLD_NOLOAD);
            if (r.h) break;
        }
        r.preloaded = r.h != nullptr;
        for (int i = 0; !r.h && i < 4; ++i) r.h = dlopen(names[i], RTLD_LAZY | RTLD_LOCAL);      // ... else load one
        if (!r.h) return;
        r.get_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.h, "ncclGetUniqueId"));
        r.init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.h, "ncclCommInitRank"));
        r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.h, "ncclAllReduce"));
        r.destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.h, "ncclCommDestroy"));
        r.ok = r.get_id && r.init_rank && r.all_reduce && r.destroy;
    });
    return r;
}
constexpr int TVAE_ERR_NO_RCCL = 100001;                 // (outside the hipError_t range)
// an ncclResult_t r != 0 is reported as 100100 + r (ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
// ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7): the caller can tell them apart
constexpr int TVAE_ERR_NCCL_BASE = 100100;
inline int nccl_rc(int r) { return r ? TVAE_ERR_NCCL_BASE + r : 0; }
}  // namespace

extern "C" {

// 0: no RCCL library could be resolved; 1: the instance that was ALREADY loaded in this process (inside PyTorch: torch's own);
// 2: this library loaded a copy of its own -- a second RCCL instance if the process also uses another one (the host side
// refuses that combination: tvae/_lib.py RcclComm)
int tvae_rccl_available(void) { return rccl().ok ? (rccl().preloaded ? 1 : 2) : 0; }
// id128: 128 bytes, filled on ONE rank (ncclGetUniqueId) and distributed to the others by the caller
int tvae_rccl_unique_id(void* id128) {
    if (!rccl().ok) return TVAE_ERR_NO_RCCL;
    if (!id128) return (int)hipErrorInvalidValue;
    return nccl_rc(rccl().get_id(reinterpret_cast<UniqueId*>(id128)));
}
// collective over all ranks (ncclCommInitRank on the CURRENT device): *comm receives the communicator
int tvae_rccl_comm_init(void** comm, int nranks, const void* id128, int rank) {
    if (!rccl().ok) return TVAE_ERR_NO_RCCL;
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return (int)hipErrorInvalidValue;
    UniqueId id = *reinterpret_cast<const UniqueId*>(id128);
    return nccl_rc(rccl().init_rank(comm, nranks, id, rank));
}
// in-place sum all-reduce of count floats on `stream` (asynchronous, like every other entry point): the flat gradient buffer
// of tvae/optim.py, or a leading / trailing segment of it (the two buckets of the data-parallel step)
int tvae_allreduce_flat(void* comm, float* buf, long count, tvae_stream_t stream) {
    if (!rccl().ok) return TVAE_ERR_NO_RCCL;
    if (!comm || (count > 0 && !buf) || count < 0) return (int)hipErrorInvalidValue;
    if (count == 0) return 0;
    return nccl_rc(rccl().all_reduce(buf, buf, (size_t)count, NCCL_FLOAT32, NCCL_SUM, comm, tvae::S(stream)));
}
int tvae_rccl_comm_destroy(void* comm) {
    if (!rccl().ok) return TVAE_ERR_NO_RCCL;
    if (!comm) return 0;
    return nccl_rc(rccl().destroy(comm));
}

}  // extern "C"
