// shm_ccl.cpp — a MULTI-PROCESS stand-in for RCCL, for tests: the eight nccl* entry points libpicles_hip.so binds (picles_hip.hip,
// RcclApi), with the ranks being PROCESSES of one host that may all sit on the same GPU.  RCCL refuses two ranks on one device,
// so on a one-GPU box `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` cannot reach the native slab ring;
// with this library bound through PICLES_CCL_LIB it can: torch.distributed (gloo) carries the 128-byte id exactly as it carries
// RCCL's, every rank calls picles_slab_comm_init / picles_slab_run_steps, and the halo blocks travel through a POSIX
// shared-memory mailbox (device -> host -> device, blocking: correctness and plumbing, not speed).
// Semantics kept from NCCL: operations are issued at ncclGroupEnd; sends and receives between a pair of ranks match in issue
// order (a FIFO of four messages per ordered pair, so that the two sends of a two-rank ring do not wait for each other).
// Test infrastructure only: nothing in picles_amd/ links or loads it by default.  (tests/native/loopback_ccl.cpp is the
// in-process, thread-rank sibling.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
constexpr int MAXR = 8, DEPTH = 4;
constexpr size_t SLOT_BYTES = 2u << 20;      // a halo block of a 4096-wide slab with 4 ghost rows is 0.8 MB

struct Msg { std::atomic<unsigned long long> posted, consumed; size_t bytes[DEPTH]; };
struct Header {
    std::atomic<int> joined, left;
    int n;
    Msg box[MAXR][MAXR];                      // [from][to]
};
struct Comm { Header *h; char *data; size_t map_bytes; int rank, n; char name[64]; };
struct Op { bool send; void *buf; size_t bytes; int peer; Comm *comm; hipStream_t stream; };

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

char *slot(Comm *c, int from, int to, unsigned long long seq)
{
    return c->data + (((size_t)from * MAXR + to) * DEPTH + (size_t)(seq % DEPTH)) * SLOT_BYTES;
}
size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}
template <class F> bool wait_for(F ok)
{
    for (long spins = 0; !ok(); spins++) {
        if (spins > 4000000) return false;              // ~ two minutes: a peer died
        if (spins > 1000) std::this_thread::sleep_for(std::chrono::microseconds(30));
    }
    return true;
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    for (Op &o : ops) {
        if (!o.send) continue;
        if (o.bytes > SLOT_BYTES) return ncclInvalidArgument;
        Comm *c = o.comm;
        Msg &m = c->h->box[c->rank][o.peer];
        const unsigned long long seq = m.posted.load(std::memory_order_relaxed);
        if (!wait_for([&] { return seq - m.consumed.load(std::memory_order_acquire) < DEPTH; })) return ncclSystemError;
        if (hipMemcpyAsync(slot(c, c->rank, o.peer, seq), o.buf, o.bytes, hipMemcpyDeviceToHost, o.stream) != hipSuccess ||
            hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        m.bytes[seq % DEPTH] = o.bytes;
        m.posted.store(seq + 1, std::memory_order_release);
    }
    for (Op &o : ops) {
        if (o.send) continue;
        Comm *c = o.comm;
        Msg &m = c->h->box[o.peer][c->rank];
        const unsigned long long seq = m.consumed.load(std::memory_order_relaxed);
        if (!wait_for([&] { return m.posted.load(std::memory_order_acquire) > seq; })) return ncclSystemError;
        if (m.bytes[seq % DEPTH] != o.bytes) return ncclInvalidArgument;
        if (hipMemcpyAsync(o.buf, slot(c, o.peer, c->rank, seq), o.bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess ||
            hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        m.consumed.store(seq + 1, std::memory_order_release);
    }
    ops.clear();
    return ncclSuccess;
}
}   // namespace

extern "C" {
#define CCL_API __attribute__((visibility("default")))
CCL_API ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<int> counter{0};
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/picles_shm_ccl_%d_%d", (int)getpid(), counter++);
    return ncclSuccess;
}
CCL_API ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank)
{
    if (n < 1 || n > MAXR || rank < 0 || rank >= n) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank; c->n = n;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = sizeof(Header) + (size_t)MAXR * MAXR * DEPTH * SLOT_BYTES;      // sparse: only the slots in use are ever touched
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->h = (Header *)p;                           // a fresh shm object is zero-filled: counters start at 0
    c->data = (char *)p + sizeof(Header);
    c->h->n = n;
    c->h->joined.fetch_add(1);
    if (!wait_for([&] { return c->h->joined.load() >= n; })) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}
CCL_API ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    if (c->h->left.fetch_add(1) + 1 == c->n) shm_unlink(c->name);      // the last rank out removes the name
    munmap((void *)c->h, c->map_bytes);
    delete c;
    return ncclSuccess;
}
CCL_API ncclResult_t ncclGroupStart(void) { t_depth++; return ncclSuccess; }
CCL_API ncclResult_t ncclGroupEnd(void)
{
    if (--t_depth > 0) return ncclSuccess;
    return run_ops(t_ops);
}
CCL_API ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    t_ops.push_back(Op{true, (void *)buf, count * type_size(t), peer, (Comm *)comm, s});
    return t_depth ? ncclSuccess : run_ops(t_ops);
}
CCL_API ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    t_ops.push_back(Op{false, buf, count * type_size(t), peer, (Comm *)comm, s});
    return t_depth ? ncclSuccess : run_ops(t_ops);
}
CCL_API const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "shm_ccl error (a peer died, a message too large for its slot, or a HIP call failed)"; }
}
