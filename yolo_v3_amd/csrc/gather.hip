// The path's ONE exchange as a C entry point: all-gather of the per-rank final-box payloads over RCCL (SURVEY 8b:
// "yv3_gather_boxes(comm, ...)").  The reference is single-GPU (no collective anywhere); this is the final step of the sharded
// detect (yolo_v3_amd/dist.py, `detect_sharded`), for hosts that own their ncclComm_t.  The Python product issues the same
// collective through torch.distributed, whose communicator is not reachable from outside torch.
//
// libyv3.so does NOT link librccl: the symbol is resolved at the first call -- in the running process if RCCL is already loaded
// (torch's copy), otherwise from librccl.so.1 / librccl.so -- and the resolved pointer is the only process-wide state (write-once).
#include <dlfcn.h>
#include <atomic>
#include "yv3_common.h"

namespace {

typedef int (*nccl_allgather_fn)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream);
constexpr int kNcclFloat32 = 7;                  // rccl.h: ncclFloat32 = 7

std::atomic<nccl_allgather_fn> g_allgather{nullptr};

nccl_allgather_fn resolve_allgather() {
    nccl_allgather_fn fn = g_allgather.load(std::memory_order_acquire);
    if (fn) return fn;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");               // already in the process (e.g. torch's librccl.so)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (sym) break;
        if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) sym = dlsym(h, "ncclAllGather");
    }
    fn = reinterpret_cast<nccl_allgather_fn>(sym);
    if (fn) g_allgather.store(fn, std::memory_order_release);
    return fn;
}

}  // namespace

extern "C" int yv3_gather_boxes(const float* payload, float* gathered, int b_local, int rows, void* rccl_comm, void* stream) {
    if (!payload || !gathered || !rccl_comm || b_local <= 0 || rows <= 0) return YV3_EINVAL;
    nccl_allgather_fn fn = resolve_allgather();
    if (!fn) return YV3_ERCCL;
    const int rc = fn(payload, gathered, (size_t)b_local * rows * 7, kNcclFloat32, rccl_comm, (hipStream_t)stream);
    return rc == 0 ? 0 : YV3_ERCCL;
}
