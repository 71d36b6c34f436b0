// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so.1 that lets the n > 1 code paths of the device-set
// entries (cpi_group_create -> ncclCommInitAll, cpi_group_gather -> ncclGroupStart / ncclRecv / ncclSend / ncclGroupEnd)
// execute on a box with ONE GPU.  Real RCCL refuses two ranks on one device; this library accepts them and carries out
// every matched (send, recv) pair as a stream-ordered device copy:
//     recv stream waits for the send stream's position at the call  ->  hipMemcpyAsync(recv buffer <- send buffer) on
//     the recv stream  ->  the send stream waits for the copy,
// i.e. the ordering contract of the real calls (rccl/rccl.h:687-722: both are "blocking for the GPU" on their streams).
// Matching follows the real rules: inside one ncclGroupStart / ncclGroupEnd, a recv on rank R from peer P pairs with the
// oldest unmatched send on rank P to peer R, and both must carry the SAME count and datatype -- a mismatch is an error
// here where real RCCL would hang or corrupt.  The definitions are compiled against the real <rccl/rccl.h>, so their
// signatures are the real prototypes.  cpi_amd binds it through CPI_AMD_RCCL_LIB (tests/test_gpu_group.py);
// fake_rccl_stats reports what crossed it.  Nothing in the product references this file.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <mutex>
#include <vector>

struct ncclComm {
    int rank, nranks, dev;
    void *world;   // identity of the ncclCommInitAll call that made it
};

namespace {
struct Op {
    bool send;
    const void *sbuf;
    void *rbuf;
    size_t count;
    ncclDataType_t dt;
    int peer;
    ncclComm *comm;
    hipStream_t stream;
    bool matched;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
std::mutex g_mu;
long long g_groups = 0, g_pairs = 0, g_bytes = 0, g_comms = 0, g_max_pairs_per_peer = 0;

size_t dt_size(ncclDataType_t dt) {
    switch (dt) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    int prev = -1;
    (void)hipGetDevice(&prev);
    ncclResult_t res = ncclSuccess;
    long long pairs = 0, bytes = 0;
    std::vector<long long> per_peer(64 * 64, 0);
    for (Op &r : ops) {
        if (r.send) continue;
        Op *s = nullptr;
        for (Op &c : ops)
            if (c.send && !c.matched && c.comm->world == r.comm->world && c.comm->rank == r.peer && c.peer == r.comm->rank) { s = &c; break; }
        if (!s) { std::fprintf(stderr, "fake_rccl: recv on rank %d from %d has no matching send in this group\n", r.comm->rank, r.peer); res = ncclInvalidUsage; break; }
        if (s->count != r.count || s->dt != r.dt) {
            std::fprintf(stderr, "fake_rccl: send %d -> %d carries %zu x type %d, the recv expects %zu x type %d\n", s->comm->rank, r.comm->rank,
                         s->count, (int)s->dt, r.count, (int)r.dt);
            res = ncclInvalidArgument;
            break;
        }
        s->matched = r.matched = true;
        const size_t nbytes = r.count * dt_size(r.dt);
        hipEvent_t e1 = nullptr, e2 = nullptr;
        bool ok = hipSetDevice(s->comm->dev) == hipSuccess && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
                  hipEventRecord(e1, s->stream) == hipSuccess;
        ok = ok && hipSetDevice(r.comm->dev) == hipSuccess && hipStreamWaitEvent(r.stream, e1, 0) == hipSuccess &&
             hipMemcpyAsync(r.rbuf, s->sbuf, nbytes, hipMemcpyDefault, r.stream) == hipSuccess &&
             hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess && hipEventRecord(e2, r.stream) == hipSuccess;
        ok = ok && hipSetDevice(s->comm->dev) == hipSuccess && hipStreamWaitEvent(s->stream, e2, 0) == hipSuccess;
        if (e1) (void)hipEventDestroy(e1);   // destruction is deferred by the runtime until the recorded work has passed
        if (e2) (void)hipEventDestroy(e2);
        if (!ok) { res = ncclUnhandledCudaError; break; }
        pairs++; bytes += (long long)nbytes;
        per_peer[(s->comm->rank & 63) * 64 + (r.comm->rank & 63)]++;
    }
    if (res == ncclSuccess)
        for (Op &c : ops) if (!c.matched) { std::fprintf(stderr, "fake_rccl: unmatched %s on rank %d\n", c.send ? "send" : "recv", c.comm->rank); res = ncclInvalidUsage; break; }
    if (prev >= 0) (void)hipSetDevice(prev);
    std::lock_guard<std::mutex> lock(g_mu);
    g_groups++; g_pairs += pairs; g_bytes += bytes;
    g_max_pairs_per_peer = 0;
    for (long long v : per_peer) if (v > g_max_pairs_per_peer) g_max_pairs_per_peer = v;
    return res;
}
}  // namespace

extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
    if (!comm || ndev <= 0) return ncclInvalidArgument;
    void *world = new int(ndev);
    for (int r = 0; r < ndev; r++) comm[r] = new ncclComm{r, ndev, devlist ? devlist[r] : r, world};
    std::lock_guard<std::mutex> lock(g_mu);
    g_comms += ndev;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl: count / datatype of a send and its recv differ)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl: unmatched send / recv, or a call outside ncclGroupStart/End)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        default: return "error (fake_rccl)";
    }
}
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return flush();
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || peer < 0 || peer >= comm->nranks || dt_size(datatype) == 0) return ncclInvalidArgument;
    if (g_depth <= 0) return ncclInvalidUsage;   // a lone send would block until its recv: the library always groups them
    g_ops.push_back(Op{true, sendbuff, nullptr, count, datatype, peer, comm, stream, false});
    return ncclSuccess;
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !recvbuff || peer < 0 || peer >= comm->nranks || dt_size(datatype) == 0) return ncclInvalidArgument;
    if (g_depth <= 0) return ncclInvalidUsage;
    g_ops.push_back(Op{false, nullptr, recvbuff, count, datatype, peer, comm, stream, false});
    return ncclSuccess;
}
// what crossed the stand-in so far: {groups flushed, matched pairs, bytes copied, communicators made, most pairs between
// one (sender, receiver) couple in the LAST group}
void fake_rccl_stats(long long out[5]) {
    std::lock_guard<std::mutex> lock(g_mu);
    out[0] = g_groups; out[1] = g_pairs; out[2] = g_bytes; out[3] = g_comms; out[4] = g_max_pairs_per_peer;
}
}
