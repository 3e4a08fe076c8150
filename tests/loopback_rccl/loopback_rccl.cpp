// TEST INFRASTRUCTURE -- not part of the product and never loaded unless GPF_RCCL_LIBRARY names it.
//
// A loopback stand-in for the nine RCCL entry points libgpf's sharded engine calls (csrc/libgpf.hip: struct Rccl), so that
// gpf_shard_resample -- the all-gathers, the grouped ncclSend / ncclRecv exchange with its counts and offsets -- can run with
// SEVERAL ranks on ONE GPU (real RCCL refuses two ranks on one device, and the build environment has 1-GPU boxes only).
// Transport: files under /dev/shm, one per message, named by (communicator id, source, destination, sequence number); device
// buffers are staged through the host.  Semantics kept: a group's operations complete together at ncclGroupEnd (sends are
// posted before any receive is waited for, so pairwise exchanges cannot deadlock); operations outside a group complete at the
// call.  Everything is synchronous with respect to the stream -- correct, merely slow.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
struct Comm {
    int rank, nranks;
    char id[64];
    std::vector<unsigned long long> sent, received;      // per peer sequence numbers
};
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;                                   // ncclInt64, ncclUint64, ncclFloat64
    }
}
std::string path_of(const Comm* c, int src, int dst, unsigned long long seq)
{
    char b[256];
    snprintf(b, sizeof b, "/dev/shm/%s_s%d_d%d_q%llu", c->id, src, dst, seq);
    return b;
}
ncclResult_t post(const Op& op)
{
    Comm* c = op.comm;
    std::vector<char> host(op.bytes);
    if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string p = path_of(c, c->rank, op.peer, c->sent[op.peer]++), tmp = p + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return ncclSystemError;
    const unsigned long long n = op.bytes;
    bool ok = fwrite(&n, sizeof n, 1, f) == 1 && (op.bytes == 0 || fwrite(host.data(), 1, op.bytes, f) == op.bytes);
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), p.c_str()) != 0) return ncclSystemError;      // rename: the message appears complete or not at all
    return ncclSuccess;
}
ncclResult_t take(const Op& op)
{
    Comm* c = op.comm;
    const std::string p = path_of(c, op.peer, c->rank, c->received[op.peer]++);
    const auto t0 = std::chrono::steady_clock::now();
    FILE* f = nullptr;
    while (!(f = fopen(p.c_str(), "rb"))) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "loopback_rccl: rank %d timed out waiting for %s\n", c->rank, p.c_str()); return ncclSystemError; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    unsigned long long n = 0;
    std::vector<char> host(op.bytes);
    bool ok = fread(&n, sizeof n, 1, f) == 1 && n == op.bytes && (op.bytes == 0 || fread(host.data(), 1, op.bytes, f) == op.bytes);
    fclose(f);
    unlink(p.c_str());
    if (!ok) { fprintf(stderr, "loopback_rccl: rank %d expected %zu bytes from rank %d, the message holds %llu\n", c->rank, op.bytes, op.peer, n); return ncclInvalidArgument; }
    if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
ncclResult_t run(std::vector<Op>& ops)
{
    for (const Op& op : ops) if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (const Op& op : ops) if (op.send) { ncclResult_t r = post(op); if (r != ncclSuccess) return r; }
    for (const Op& op : ops) if (!op.send) { ncclResult_t r = take(op); if (r != ncclSuccess) return r; }
    ops.clear();
    return ncclSuccess;
}
ncclResult_t enqueue(const Op& op)
{
    if (!op.comm || op.peer < 0 || op.peer >= op.comm->nranks) return ncclInvalidArgument;
    g_ops.push_back(op);
    return g_depth > 0 ? ncclSuccess : run(g_ops);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "gpf_lb_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->nranks = nranks;
    strncpy(c->id, id.internal, sizeof c->id - 1);
    c->sent.assign(nranks, 0); c->received.assign(nranks, 0);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm*>(comm); return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? run(g_ops) : ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(Op{true, const_cast<void*>(buf), count * dtype_size(dt), peer, reinterpret_cast<Comm*>(comm), stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(Op{false, buf, count * dtype_size(dt), peer, reinterpret_cast<Comm*>(comm), stream});
}
ncclResult_t ncclAllGather(const void* src, void* dst, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    const size_t bytes = count * dtype_size(dt);
    ncclGroupStart();
    for (int r = 0; r < c->nranks; ++r) {
        ncclSend(src, count, dt, r, comm, stream);
        ncclRecv(static_cast<char*>(dst) + (size_t)r * bytes, count, dt, r, comm, stream);
    }
    return ncclGroupEnd();
}
const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "loopback: HIP error";
        case ncclSystemError: return "loopback: system error (file transport / timeout)";
        case ncclInvalidArgument: return "loopback: invalid argument (or message size mismatch)";
        case ncclInvalidUsage: return "loopback: invalid usage";
        default: return "loopback: error";
    }
}
}
