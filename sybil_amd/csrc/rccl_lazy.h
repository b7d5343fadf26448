// rccl_lazy.h -- RCCL bound when the first communicator is made, not when the library is loaded.
//
// librccl.so.1 is a 570 MB shared object; as a link-time dependency it was mapped, relocated and had its code objects
// registered by every process that loaded libsybilgpu.so -- every one-GPU `sybil-gpu-query`, every test process -- whether or
// not it ever made a communicator (round 6: the cold CLI's time before main()).  The nine entry points the engine uses are
// looked up on the first sybl_comm_unique_id / sybl_comm_init instead: first among what the process already has
// (dlsym(RTLD_DEFAULT): a host that linked RCCL itself, torch's bundled copy, or -- in the test suite -- the shared-memory
// stand-in in LD_PRELOAD), then by dlopen("librccl.so.1").
#pragma once
#include <rccl/rccl.h>

namespace sybl {
struct RcclApi {
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    bool ok = false;
    const char *why = "";
};
const RcclApi &rccl();  // (rccl.cpp) resolved on first use; ok == false: no RCCL in this process and none to be loaded
}  // namespace sybl

#define ncclGetUniqueId (sybl::rccl().GetUniqueId)
#define ncclCommInitRank (sybl::rccl().CommInitRank)
#define ncclCommDestroy (sybl::rccl().CommDestroy)
#define ncclGetErrorString (sybl::rccl().GetErrorString)
#define ncclAllReduce (sybl::rccl().AllReduce)
#define ncclAllGather (sybl::rccl().AllGather)
#define ncclReduceScatter (sybl::rccl().ReduceScatter)
#define ncclGroupStart (sybl::rccl().GroupStart)
#define ncclGroupEnd (sybl::rccl().GroupEnd)
