// TEST INFRASTRUCTURE — a stand-in for the eight librccl entry points the library resolves (polystokes_amd/csrc/ps_dist.hpp: rccl()),
// so that the ASYNCHRONOUS transport branch of the distributed solve (Dist::transport / allreduce with useRccl) can run with N > 1
// ranks on a box with ONE GPU: real RCCL refuses several ranks on one device, the TCP transport synchronises the host with the device
// twice per exchange and the in-process group shares one stream — both hide a missing ordering between the solver and the comm stream.
//
// Ranks = processes sharing GPU 0 — or several ranks per process, one THREAD each (r05: eight ranks as four processes of two, the box
// admits six GPU processes): a peer of the same process is reached through its plain device pointer instead of an IPC mapping, nothing
// else differs.  Every message is stream-ordered on the caller's stream, as in RCCL, and NOTHING in it involves the
// host after the communicator is built (no hipStreamSynchronize / hipEventSynchronize / hipDeviceSynchronize, no host-side handshake):
//   send k:  [stream waits until the receiver has taken message k - 2, whose mailbox slot this one reuses]
//            device copy into the receiver's mailbox (its device memory, mapped with hipIpcOpenMemHandle)
//            -> hipStreamWriteValue32: the receiver's "ready" counter of this pair := k + 1
//   recv k:  hipStreamWaitValue32 on the own "ready" counter >= k + 1  ->  device copy out of the own mailbox
//            -> hipStreamWriteValue32: the sender's "consumed" counter of this pair := k + 1
// The counters live in the same IPC-mapped allocations as the mailboxes; the stream memory operations order the copies of the two
// processes on the device (measured with scripts/t_ipc_event.hip: a value wait enqueued 0.8 s before the other process' write
// releases exactly then, the data behind it visible).
// Why not hipIpcEventHandle events (what this was first built on): HIP's inter-process event carries a ring of 32 signals, and the
// 33rd hipEventRecord / hipStreamWaitEvent pair on one event fails with hipErrorInvalidValue (seen here at message 64 of a pair with
// two events) — a solve exchanges thousands of messages per pair.  The value waits have no such limit and need no "record before
// wait" handshake on the host.
// All-reduce = every rank sends its values to every other rank through the same mailboxes, then one kernel adds the `world`
// contributions in rank order (the same order on every rank: identical results everywhere).
// Loaded by tests only, through PS_RCCL_LIB (ps_dist.hpp).  Not part of the product.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MAXW = 8;             // ranks
constexpr int SLOTS = 2;            // mailbox slots per ordered pair and channel (message k uses slot k % SLOTS)
constexpr int CH = 2;               // channel 0: send / recv payloads, channel 1: all-reduce contributions
constexpr size_t AR_BYTES = 64 * 8; // an all-reduce carries <= 64 doubles
constexpr int ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5;

struct RankCtl {
    std::atomic<int> ready;                          // 1: the handle below is valid; 3: this rank has left
    hipIpcMemHandle_t mailbox;                       // this rank's mailbox allocation (flags page + message slots)
    int pid;                                         // the process holding it, and its address there: a rank of the SAME process cannot
    char* base;                                      // open its own process' handle — it uses the pointer
};
struct Shm { RankCtl rank[MAXW]; };                  // only the start-up handshake goes through the host

struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; hipStream_t stream; };

// A rank's allocation: [flags: 4 KB][per source rank: SLOTS payload slots + SLOTS all-reduce slots].
// Flags (uint32 counters), all written by the PEER with hipStreamWriteValue32 and waited for locally:
//   ready[src][ch]    at word  (src * CH + ch)            : messages src has delivered into my mailbox
//   consumed[dst][ch] at word  256 + (dst * CH + ch)      : messages dst has taken out of ITS mailbox that came from me
constexpr size_t FLAG_BYTES = 4096;
struct Comm {
    int rank = 0, world = 1;
    Shm* shm = nullptr;
    char shmName[80] = {0};
    size_t slotBytes = 0;
    char* mailbox = nullptr;
    char* peerMailbox[MAXW] = {nullptr};
    bool peerLocal[MAXW] = {};                       // peerMailbox[q] is a pointer of this process, not an IPC mapping
    uint32_t sendSeq[MAXW][CH] = {}, recvSeq[MAXW][CH] = {};
    double* arStage = nullptr;                       // world x 64 doubles: the contributions lined up for the sum kernel
    size_t chBytes(int ch) const { return ch == 0 ? slotBytes : AR_BYTES; }
    size_t regionOff(int src, int ch, int slot) const {    // inside a rank's allocation
        const size_t perSrc = SLOTS * (slotBytes + AR_BYTES);
        return FLAG_BYTES + (size_t)src * perSrc + (ch == 0 ? 0 : SLOTS * slotBytes) + (size_t)slot * chBytes(ch);
    }
    static uint32_t* readyFlag(char* base, int src, int ch) { return (uint32_t*)base + (src * CH + ch); }
    static uint32_t* consFlag(char* base, int dst, int ch) { return (uint32_t*)base + 256 + (dst * CH + ch); }
};

thread_local int g_groupDepth = 0;
thread_local std::vector<Op>* g_ops = nullptr;
thread_local Comm* g_groupComm = nullptr;

#define STUB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "[stub rccl] %s -> %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return ncclUnhandledCudaError; } } while (0)

const bool g_dbg = getenv("PS_STUB_DEBUG") != nullptr;
// Payloads move by a KERNEL on the caller's stream, not by hipMemcpyAsync: a device-to-device copy goes through the HSA runtime's copy
// queue (SDMA ring or blit queue), ONE per process and agent, filled under a mutex by every stream of the process.  With two ranks of a
// communicator in one process (threads) a copy of rank A that waits for A's value-wait sits at the head of that queue, the ring fills
// with the 25 iterations A's host enqueues ahead, A's thread spins for space holding the mutex — and rank B's thread, whose copy would
// release A, blocks on the mutex (native stacks: scripts/dbg/btdump.c).  A kernel uses the stream's own queue.  (All messages are doubles.)
__global__ void k_copy_words(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int copyOn(hipStream_t st, void* dst, const void* src, size_t bytes) {
    if ((bytes & 7) || ((uintptr_t)dst & 7) || ((uintptr_t)src & 7)) return ncclInvalidArgument;
    const size_t n = bytes / 8;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_copy_words, dim3(grid ? grid : 1), dim3(256), 0, st, (unsigned long long*)dst, (const unsigned long long*)src, n);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
int doSend(Comm* c, int ch, const void* buf, size_t bytes, int peer, hipStream_t st) {
    if (bytes > c->chBytes(ch)) { std::fprintf(stderr, "[stub rccl] message of %zu bytes exceeds the mailbox slot (%zu): raise PS_STUB_MAILBOX_MB\n", bytes, c->chBytes(ch)); return ncclInvalidArgument; }
    const uint32_t k = c->sendSeq[peer][ch]++;
    const int slot = (int)(k % SLOTS);
    if (g_dbg) std::fprintf(stderr, "[stub %d] send ch %d -> %d, %zu B, k %u, stream %p\n", c->rank, ch, peer, bytes, k, (void*)st);
    if (k >= (uint32_t)SLOTS)   // the slot's previous message (k - SLOTS) must have been copied out by the receiver
        STUB_HIP(hipStreamWaitValue32(st, Comm::consFlag(c->mailbox, peer, ch), k - SLOTS + 1, hipStreamWaitValueGte, 0xffffffffu));
    if (bytes) { const int rc = copyOn(st, c->peerMailbox[peer] + c->regionOff(c->rank, ch, slot), buf, bytes); if (rc) return rc; }
    STUB_HIP(hipStreamWriteValue32(st, Comm::readyFlag(c->peerMailbox[peer], c->rank, ch), k + 1, 0));
    return ncclSuccess;
}
int doRecv(Comm* c, int ch, void* buf, size_t bytes, int peer, hipStream_t st) {
    if (bytes > c->chBytes(ch)) return ncclInvalidArgument;
    const uint32_t k = c->recvSeq[peer][ch]++;
    const int slot = (int)(k % SLOTS);
    if (g_dbg) std::fprintf(stderr, "[stub %d] recv ch %d <- %d, %zu B, k %u, stream %p\n", c->rank, ch, peer, bytes, k, (void*)st);
    STUB_HIP(hipStreamWaitValue32(st, Comm::readyFlag(c->mailbox, peer, ch), k + 1, hipStreamWaitValueGte, 0xffffffffu));
    if (bytes) { const int rc = copyOn(st, buf, c->mailbox + c->regionOff(peer, ch, slot), bytes); if (rc) return rc; }
    STUB_HIP(hipStreamWriteValue32(st, Comm::consFlag(c->peerMailbox[peer], c->rank, ch), k + 1, 0));
    return ncclSuccess;
}
// all sends of a group first, then its receives: a stream never waits for a message of the SAME group before its own have been
// enqueued — the exchange pattern of Dist::transport cannot deadlock whatever the order of the calls inside the group
int flush(Comm* c, std::vector<Op>& ops) {
    {   // messages to self (ps_comm_selftest): the k-th send pairs with the k-th receive as one device copy
        std::vector<const Op*> ss, rr;
        for (const Op& o : ops) if (o.peer == c->rank) (o.send ? ss : rr).push_back(&o);
        if (ss.size() != rr.size()) return ncclInvalidUsage;
        for (size_t k = 0; k < ss.size(); ++k) {
            if (ss[k]->bytes != rr[k]->bytes) return ncclInvalidArgument;
            if (ss[k]->bytes) { const int rc = copyOn(rr[k]->stream, rr[k]->rbuf, ss[k]->sbuf, ss[k]->bytes); if (rc) return rc; }
        }
        std::vector<Op> rest;
        for (const Op& o : ops) if (o.peer != c->rank) rest.push_back(o);
        ops.swap(rest);
    }
    for (const Op& o : ops) if (o.send) { const int rc = doSend(c, 0, o.sbuf, o.bytes, o.peer, o.stream); if (rc) return rc; }
    for (const Op& o : ops) if (!o.send) { const int rc = doRecv(c, 0, o.rbuf, o.bytes, o.peer, o.stream); if (rc) return rc; }
    return ncclSuccess;
}
__global__ void k_sum_ranks(const double* __restrict__ stage, int world, int count, double* __restrict__ out) {
    const int i = threadIdx.x;
    if (i >= count) return;
    double s = 0.;
    for (int r = 0; r < world; ++r) s += stage[r * 64 + i];     // rank order: the same sum on every rank
    out[i] = s;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* id128) {
    if (!id128) return ncclInvalidArgument;
    std::memset(id128, 0, 128);
    unsigned long long r = (unsigned long long)std::chrono::high_resolution_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 32);
    std::snprintf((char*)id128, 64, "/ps_stub_rccl_%016llx", r);
    return ncclSuccess;
}

struct StubUid { char internal[128]; };
int ncclCommInitRank(void** comm, int world, StubUid id, int rank) {
    if (!comm || world < 1 || world > MAXW || rank < 0 || rank >= world) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    std::snprintf(c->shmName, sizeof(c->shmName), "%s", id.internal);
    const char* mb = getenv("PS_STUB_MAILBOX_MB");
    c->slotBytes = (size_t)(mb ? atoi(mb) : 8) << 20;
    const int fd = shm_open(c->shmName, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); delete c; return ncclSystemError; }     // (new pages read as zero: every counter starts at 0)
    c->shm = (Shm*)mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->shm == MAP_FAILED) { delete c; return ncclSystemError; }
    const size_t perSrc = SLOTS * (c->slotBytes + AR_BYTES);
    STUB_HIP(hipMalloc((void**)&c->mailbox, FLAG_BYTES + perSrc * (size_t)world));
    STUB_HIP(hipMemset(c->mailbox, 0, FLAG_BYTES));                                   // every counter starts at 0
    STUB_HIP(hipMalloc((void**)&c->arStage, (size_t)MAXW * 64 * sizeof(double)));
    STUB_HIP(hipMemset(c->arStage, 0, (size_t)MAXW * 64 * sizeof(double)));
    STUB_HIP(hipDeviceSynchronize());                                                 // (start-up only: the zeros are there before a peer can write)
    RankCtl& me = c->shm->rank[rank];
    if (world > 1) STUB_HIP(hipIpcGetMemHandle(&me.mailbox, c->mailbox));
    me.pid = (int)getpid(); me.base = c->mailbox;
    me.ready.store(1, std::memory_order_release);
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        RankCtl& pr = c->shm->rank[q];
        const auto t0 = std::chrono::steady_clock::now();
        while (pr.ready.load(std::memory_order_acquire) < 1) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { std::fprintf(stderr, "[stub rccl] rank %d never arrived\n", q); return ncclSystemError; }
        }
        if (pr.pid == (int)getpid()) { c->peerMailbox[q] = pr.base; c->peerLocal[q] = true; }
        else STUB_HIP(hipIpcOpenMemHandle((void**)&c->peerMailbox[q], pr.mailbox, hipIpcMemLazyEnablePeerAccess));
    }
    *comm = c;
    return ncclSuccess;
}

int ncclGroupStart() {
    if (g_groupDepth++ == 0) { if (!g_ops) g_ops = new std::vector<Op>(); g_ops->clear(); g_groupComm = nullptr; }
    return ncclSuccess;
}
int ncclGroupEnd() {
    if (g_groupDepth <= 0) return ncclInvalidUsage;
    if (--g_groupDepth > 0) return ncclSuccess;
    if (!g_groupComm || g_ops->empty()) return ncclSuccess;
    return flush(g_groupComm, *g_ops);
}
static int p2p(bool send, const void* sbuf, void* rbuf, size_t count, int dtype, int peer, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 8 || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    Op o{send, sbuf, rbuf, count * 8, peer, st};
    if (peer == c->rank) {   // to self (ps_comm_selftest): a send and a receive of one group pair up as a device copy
        if (g_groupDepth == 0) return ncclInvalidUsage;
    }
    if (g_groupDepth > 0) { g_groupComm = c; g_ops->push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return flush(c, one);
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) { return p2p(true, buf, nullptr, count, dtype, peer, comm, st); }
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) { return p2p(false, nullptr, buf, count, dtype, peer, comm, st); }

int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 8 || op != 0 || count > 64) return ncclInvalidArgument;
    { const int rc = copyOn(st, c->arStage + (size_t)c->rank * 64, sendbuf, count * 8); if (rc) return rc; }
    for (int q = 0; q < c->world; ++q) if (q != c->rank) { const int rc = doSend(c, 1, sendbuf, count * 8, q, st); if (rc) return rc; }
    for (int q = 0; q < c->world; ++q) if (q != c->rank) { const int rc = doRecv(c, 1, c->arStage + (size_t)q * 64, count * 8, q, st); if (rc) return rc; }
    hipLaunchKernelGGL(k_sum_ranks, dim3(1), dim3(64), 0, st, (const double*)c->arStage, c->world, (int)count, (double*)recvbuf);
    return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();                    // teardown only: the mailboxes go away
    for (int q = 0; q < c->world; ++q) if (q != c->rank && c->peerMailbox[q] && !c->peerLocal[q]) (void)hipIpcCloseMemHandle(c->peerMailbox[q]);
    // the last rank out removes the shared-memory name (the others may still hold their mappings)
    c->shm->rank[c->rank].ready.store(3, std::memory_order_release);
    bool last = true;
    for (int q = 0; q < c->world; ++q) if (c->shm->rank[q].ready.load(std::memory_order_acquire) != 3) last = false;
    if (last || c->world == 1) shm_unlink(c->shmName);
    munmap(c->shm, sizeof(Shm));
    if (c->mailbox) (void)hipFree(c->mailbox);
    if (c->arStage) (void)hipFree(c->arStage);
    delete c;
    return ncclSuccess;
}

}  // extern "C"
