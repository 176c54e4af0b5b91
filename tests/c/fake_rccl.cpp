// tests/c/fake_rccl.cpp -- TEST INFRASTRUCTURE, never shipped, never loaded by the product on its own.
//
// A stand-in for librccl.so that lets the N > 1 code of rt_gang_* (rust-tracer_amd/csrc/rt_capi.hip: ncclCommInitAll, one ncclGather
// per frame inside ncclGroupStart / ncclGroupEnd, shards double-buffered under the next frame's render) EXECUTE on a box with one GPU:
// every "rank" is a communicator on the same device, and the gather is what RCCL's is on the wire -- rank r's `count` elements land at
// recvbuff + r * count on the root -- done with hipMemcpyAsync on the ranks' own streams, the root's stream waiting for every copy.
// Only the entry points rt_capi.hip binds exist.  Selected with rt_debug_rccl_library(<this file's .so>) (csrc/rt_debug.h); the real
// library needs one GPU per rank.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

namespace {

struct Comm { int rank, nranks, device; struct World *world; };
struct Pending { const void *send; void *recv; size_t bytes; int root; Comm *comm; hipStream_t stream; };
struct World { std::vector<Comm *> comms; };

std::mutex g_mu;
int g_group_depth = 0;
std::vector<Pending> g_pending;

ncclResult_t run_gathers(std::vector<Pending> &calls)
{
    // one collective = one call per rank of a world; process world by world, in the order the calls came
    while (!calls.empty()) {
        World *w = calls.front().comm->world;
        std::vector<Pending> mine;
        for (auto it = calls.begin(); it != calls.end();)
            if (it->comm->world == w && mine.size() < w->comms.size()) { mine.push_back(*it); it = calls.erase(it); } else ++it;
        if (mine.size() != w->comms.size()) return ncclInvalidUsage;            // a rank is missing from the group
        const int root = mine[0].root;
        const Pending *rp = nullptr;
        for (const Pending &p : mine) {
            if (p.root != root || p.bytes != mine[0].bytes) return ncclInvalidArgument;
            if (p.comm->rank == root) rp = &p;
        }
        if (!rp || !rp->recv) return ncclInvalidArgument;
        for (const Pending &p : mine) {
            if (hipSetDevice(p.comm->device) != hipSuccess) return ncclUnhandledCudaError;
            char *dst = static_cast<char *>(rp->recv) + (size_t)p.comm->rank * p.bytes;
            if (hipMemcpyAsync(dst, p.send, p.bytes, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (&p == rp) continue;
            // the root's stream completes the collective only when every rank's piece has landed
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            hipError_t e = hipEventRecord(ev, p.stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(rp->stream, ev, 0);
            (void)hipEventDestroy(ev);                                          // released once it has completed
            if (e != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) { if (!version) return ncclInvalidArgument; *version = 0; return ncclSuccess; }      // 0: not a real RCCL

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl: error"; }

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World *w = new World();
    for (int r = 0; r < ndev; ++r) {
        Comm *c = new Comm{ r, ndev, devlist ? devlist[r] : r, w };
        w->comms.push_back(c);
        comms[r] = reinterpret_cast<ncclComm_t>(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    World *w = c->world;
    for (auto &p : w->comms) if (p == c) p = nullptr;
    delete c;
    bool any = false;
    for (auto p : w->comms) any = any || p != nullptr;
    if (!any) delete w;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { std::lock_guard<std::mutex> lk(g_mu); ++g_group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<Pending> calls;
    calls.swap(g_pending);
    return run_gathers(calls);
}

ncclResult_t ncclGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t esz = type_bytes(datatype);
    if (!c || !sendbuff || esz == 0 || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.push_back(Pending{ sendbuff, recvbuff, sendcount * esz, root, c, stream });
    if (g_group_depth > 0) return ncclSuccess;
    // outside a group a one-rank world can run at once; more ranks need the group (one thread drives them all)
    if (c->nranks != 1) { g_pending.pop_back(); return ncclInvalidUsage; }
    std::vector<Pending> calls;
    calls.swap(g_pending);
    return run_gathers(calls);
}

}  // extern "C"
