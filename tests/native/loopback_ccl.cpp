// Loopback communicator for tests: the eight nccl* entry points libpicles_hip.so binds (picles_hip.hip, RcclApi), with the
// ranks being THREADS of one process that share one GPU.  RCCL itself refuses two ranks on one device, so on a one-GPU box the
// native slab ring (picles_slab_run_steps) can only be run with world = 1 through the real library; with this stand-in
// (selected through PICLES_CCL_LIB) its neighbour arithmetic, its send/recv pairing and its stream ordering run with world = 2, 3,
// 4.  Semantics kept from NCCL: operations are issued at ncclGroupEnd; sends and receives between a pair of ranks match in issue
// order; a receive is stream-ordered after the sender's stream at the time of the send, and the sender's stream is ordered after
// the copy (the send buffer may be overwritten by later work on that stream).  The copy itself is a device-to-device
// hipMemcpyAsync on the receiver's stream.  Test infrastructure only: nothing in picles_amd/ links or loads it by default.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace {
struct Msg {
    const void *src;
    size_t bytes;
    hipEvent_t ready, done;
    bool consumed = false;
};
struct Group {
    int n = 0, joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Msg *>> box;   // (from, to) -> messages in issue order
};
struct Comm {
    Group *g;
    int rank, n;
};
struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
std::mutex g_m;
std::map<std::string, Group *> g_groups;
unsigned long long g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    struct Sent { Msg *msg; hipStream_t stream; Group *g; };
    std::vector<Sent> mine;
    for (Op &o : ops) {
        if (!o.send) continue;
        Msg *msg = new Msg{o.buf, o.bytes, nullptr, nullptr};
        if (hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&msg->done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(msg->ready, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        Group *g = o.comm->g;
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->box[{o.comm->rank, o.peer}].push_back(msg);
        }
        g->cv.notify_all();
        mine.push_back(Sent{msg, o.stream, g});
    }
    for (Op &o : ops) {
        if (o.send) continue;
        Group *g = o.comm->g;
        Msg *msg;
        {
            std::unique_lock<std::mutex> lk(g->m);
            auto &q = g->box[{o.peer, o.comm->rank}];
            g->cv.wait(lk, [&] { return !q.empty(); });
            msg = q.front();
            q.pop_front();
        }
        if (msg->bytes != o.bytes) return ncclInvalidArgument;
        if (hipStreamWaitEvent(o.stream, msg->ready, 0) != hipSuccess ||
            hipMemcpyAsync(o.buf, msg->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
            hipEventRecord(msg->done, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g->m);
            msg->consumed = true;
        }
        g->cv.notify_all();
    }
    for (Sent &ms : mine) {
        Msg *msg = ms.msg;
        {
            std::unique_lock<std::mutex> lk(ms.g->m);
            ms.g->cv.wait(lk, [&] { return msg->consumed; });
        }
        if (hipStreamWaitEvent(ms.stream, msg->done, 0) != hipSuccess) return ncclUnhandledCudaError;
        (void)hipEventDestroy(msg->ready);
        (void)hipEventDestroy(msg->done);
        delete msg;
    }
    ops.clear();
    return ncclSuccess;
}
}   // namespace

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_m);
    memset(id, 0, sizeof(*id));
    unsigned long long v = g_next_id++;
    memcpy(id->internal, "loopback", 8);
    memcpy(id->internal + 8, &v, sizeof(v));
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (memcmp(id.internal, "loopback", 8) != 0 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Group *g;
    {
        std::lock_guard<std::mutex> lk(g_m);
        std::string key(id.internal, sizeof(id.internal));
        auto it = g_groups.find(key);
        if (it == g_groups.end()) {
            g = new Group();
            g->n = nranks;
            g_groups[key] = g;
        } else {
            g = it->second;
        }
    }
    if (g->n != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->joined++;
        g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->joined >= g->n; });      // the rendezvous of the real ncclCommInitRank
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{g, rank, nranks});
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<Comm *>(comm);      // the group object stays (another rank may still be inside it)
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart(void)
{
    t_depth++;
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run_ops(t_ops);
}
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                                                             hipStream_t stream)
{
    t_ops.push_back(Op{true, const_cast<void *>(buf), count * type_size(type), peer, reinterpret_cast<Comm *>(comm), stream});
    return t_depth > 0 ? ncclSuccess : run_ops(t_ops);
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                                                             hipStream_t stream)
{
    t_ops.push_back(Op{false, buf, count * type_size(type), peer, reinterpret_cast<Comm *>(comm), stream});
    return t_depth > 0 ? ncclSuccess : run_ops(t_ops);
}
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : "loopback communicator error";
}
}
