// TEST INFRASTRUCTURE — a stand-in for the eight librccl entry points the library resolves (polystokes_amd/csrc/ps_dist.hpp: rccl()),
// so that the ASYNCHRONOUS transport branch of the distributed solve (Dist::transport / allreduce with useRccl) can run with N > 1
// ranks on a box with ONE GPU: real RCCL refuses several ranks on one device, the TCP transport synchronises the host with the device
// twice per exchange and the in-process group shares one stream — both hide a missing ordering between the solver and the comm stream.
//
// Ranks = processes sharing GPU 0.  Every message is stream-ordered on the caller's stream, as in RCCL:
//   send:  [wait for the receiver's "consumed" event of the slot's previous use]  copy into the receiver's mailbox (device memory of
//          the receiver, mapped with hipIpcOpenMemHandle)  ->  record the slot's "ready" event (hipEventInterprocess)
//   recv:  wait for the sender's "ready" event (hipIpcOpenEventHandle)  ->  copy out of the own mailbox  ->  record "consumed"
// The host never waits for the DEVICE (no hipStreamSynchronize / hipEventSynchronize / hipDeviceSynchronize anywhere): it only waits,
// through counters in a POSIX shared-memory block, until the peer has ENQUEUED the matching record — an inter-process event wait
// binds to the records issued before it, so the record call has to come first in host time.  All-reduce = every rank sends its
// values to every other rank through the same mailboxes, then one kernel adds the `world` contributions in rank order (the same
// order on every rank: identical results everywhere, like ncclAllReduce's guarantee for a fixed communicator).
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

struct PairCtl {                    // ordered pair (src -> dst), one channel
    std::atomic<uint64_t> sentEnq;  // messages whose copy + "ready" record the sender has enqueued
    std::atomic<uint64_t> consEnq;  // messages whose wait + copy-out + "consumed" record the receiver has enqueued
};
struct RankCtl {
    std::atomic<int> ready;                          // 1: handles below are valid; 2: peers opened
    hipIpcMemHandle_t mailbox;                       // this rank's mailbox allocation
    hipIpcEventHandle_t evReady[MAXW][CH][SLOTS];    // recorded by THIS rank as sender to [dst]
    hipIpcEventHandle_t evCons[MAXW][CH][SLOTS];     // recorded by THIS rank as receiver from [src]
};
struct Shm {
    std::atomic<int> magic;
    size_t slotBytes;
    RankCtl rank[MAXW];
    PairCtl pair[MAXW][MAXW][CH];                    // [src][dst][channel]
};

struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
    int rank = 0, world = 1;
    Shm* shm = nullptr;
    char shmName[80] = {0};
    size_t slotBytes = 0;
    char* mailbox = nullptr;                         // own: [src][channel][slot] regions
    char* peerMailbox[MAXW] = {nullptr};
    hipEvent_t myReady[MAXW][CH][SLOTS] = {}, myCons[MAXW][CH][SLOTS] = {};        // created here
    hipEvent_t peerReady[MAXW][CH][SLOTS] = {}, peerCons[MAXW][CH][SLOTS] = {};    // opened: peer's ready (as sender to me) / consumed (as receiver from me)
    uint64_t sendSeq[MAXW][CH] = {}, recvSeq[MAXW][CH] = {};
    double* arStage = nullptr;                       // world x 64 doubles: the contributions lined up for the sum kernel
    size_t chBytes(int ch) const { return ch == 0 ? slotBytes : AR_BYTES; }
    size_t regionOff(int src, int ch, int slot) const {    // inside a rank's mailbox
        const size_t perSrc = SLOTS * (slotBytes + AR_BYTES);
        return (size_t)src * perSrc + (ch == 0 ? 0 : SLOTS * slotBytes) + (size_t)slot * chBytes(ch);
    }
};

thread_local int g_groupDepth = 0;
thread_local std::vector<Op>* g_ops = nullptr;
thread_local Comm* g_groupComm = nullptr;

bool waitCounter(std::atomic<uint64_t>& c, uint64_t atLeast, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (c.load(std::memory_order_acquire) < atLeast) {
        if (++spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { std::fprintf(stderr, "[stub rccl] timed out waiting for %s\n", what); return false; }
    }
    return true;
}
#define STUB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "[stub rccl] %s -> %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return ncclUnhandledCudaError; } } while (0)

int doSend(Comm* c, int ch, const void* buf, size_t bytes, int peer, hipStream_t st) {
    if (bytes > c->chBytes(ch)) { std::fprintf(stderr, "[stub rccl] message of %zu bytes exceeds the mailbox slot (%zu): raise PS_STUB_MAILBOX_MB\n", bytes, c->chBytes(ch)); return ncclInvalidArgument; }
    const uint64_t k = c->sendSeq[peer][ch]++;
    const int slot = (int)(k % SLOTS);
    PairCtl& pc = c->shm->pair[c->rank][peer][ch];
    if (k >= SLOTS) {   // the slot's previous message (k - SLOTS) must have been copied out: the receiver has enqueued that -> wait for its event on the stream
        if (!waitCounter(pc.consEnq, k - SLOTS + 1, "the receiver to take an earlier message")) return ncclSystemError;
        STUB_HIP(hipStreamWaitEvent(st, c->peerCons[peer][ch][slot], 0));
    }
    if (bytes) STUB_HIP(hipMemcpyAsync(c->peerMailbox[peer] + c->regionOff(c->rank, ch, slot), buf, bytes, hipMemcpyDeviceToDevice, st));
    STUB_HIP(hipEventRecord(c->myReady[peer][ch][slot], st));
    pc.sentEnq.store(k + 1, std::memory_order_release);
    return ncclSuccess;
}
int doRecv(Comm* c, int ch, void* buf, size_t bytes, int peer, hipStream_t st) {
    if (bytes > c->chBytes(ch)) return ncclInvalidArgument;
    const uint64_t k = c->recvSeq[peer][ch]++;
    const int slot = (int)(k % SLOTS);
    PairCtl& pc = c->shm->pair[peer][c->rank][ch];
    if (!waitCounter(pc.sentEnq, k + 1, "the sender to enqueue its message")) return ncclSystemError;
    STUB_HIP(hipStreamWaitEvent(st, c->peerReady[peer][ch][slot], 0));
    if (bytes) STUB_HIP(hipMemcpyAsync(buf, c->mailbox + c->regionOff(peer, ch, slot), bytes, hipMemcpyDeviceToDevice, st));
    STUB_HIP(hipEventRecord(c->myCons[peer][ch][slot], st));
    pc.consEnq.store(k + 1, std::memory_order_release);
    return ncclSuccess;
}
// all sends of a group first, then its receives: no rank waits (on the host) for a message of the SAME group before having
// enqueued its own — the exchange pattern of Dist::transport cannot deadlock whatever the order of the calls inside the group
int flush(Comm* c, std::vector<Op>& ops) {
    {   // messages to self (ps_comm_selftest): the k-th send pairs with the k-th receive as one device copy
        std::vector<const Op*> ss, rr;
        for (const Op& o : ops) if (o.peer == c->rank) (o.send ? ss : rr).push_back(&o);
        if (ss.size() != rr.size()) return ncclInvalidUsage;
        for (size_t k = 0; k < ss.size(); ++k) {
            if (ss[k]->bytes != rr[k]->bytes) return ncclInvalidArgument;
            if (ss[k]->bytes) STUB_HIP(hipMemcpyAsync(rr[k]->rbuf, ss[k]->sbuf, ss[k]->bytes, hipMemcpyDeviceToDevice, rr[k]->stream));
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
    STUB_HIP(hipMalloc((void**)&c->mailbox, perSrc * (size_t)world));
    STUB_HIP(hipMalloc((void**)&c->arStage, (size_t)MAXW * 64 * sizeof(double)));
    STUB_HIP(hipMemset(c->arStage, 0, (size_t)MAXW * 64 * sizeof(double)));
    RankCtl& me = c->shm->rank[rank];
    if (world > 1) STUB_HIP(hipIpcGetMemHandle(&me.mailbox, c->mailbox));
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        for (int ch = 0; ch < CH; ++ch)
            for (int s = 0; s < SLOTS; ++s) {
                STUB_HIP(hipEventCreateWithFlags(&c->myReady[q][ch][s], hipEventDisableTiming | hipEventInterprocess));
                STUB_HIP(hipEventCreateWithFlags(&c->myCons[q][ch][s], hipEventDisableTiming | hipEventInterprocess));
                STUB_HIP(hipIpcGetEventHandle(&me.evReady[q][ch][s], c->myReady[q][ch][s]));
                STUB_HIP(hipIpcGetEventHandle(&me.evCons[q][ch][s], c->myCons[q][ch][s]));
            }
    }
    me.ready.store(1, std::memory_order_release);
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        RankCtl& pr = c->shm->rank[q];
        const auto t0 = std::chrono::steady_clock::now();
        while (pr.ready.load(std::memory_order_acquire) < 1) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { std::fprintf(stderr, "[stub rccl] rank %d never arrived\n", q); return ncclSystemError; }
        }
        STUB_HIP(hipIpcOpenMemHandle((void**)&c->peerMailbox[q], pr.mailbox, hipIpcMemLazyEnablePeerAccess));
        for (int ch = 0; ch < CH; ++ch)
            for (int s = 0; s < SLOTS; ++s) {
                STUB_HIP(hipIpcOpenEventHandle(&c->peerReady[q][ch][s], pr.evReady[rank][ch][s]));
                STUB_HIP(hipIpcOpenEventHandle(&c->peerCons[q][ch][s], pr.evCons[rank][ch][s]));
            }
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
    STUB_HIP(hipMemcpyAsync(c->arStage + (size_t)c->rank * 64, sendbuf, count * 8, hipMemcpyDeviceToDevice, st));
    for (int q = 0; q < c->world; ++q) if (q != c->rank) { const int rc = doSend(c, 1, sendbuf, count * 8, q, st); if (rc) return rc; }
    for (int q = 0; q < c->world; ++q) if (q != c->rank) { const int rc = doRecv(c, 1, c->arStage + (size_t)q * 64, count * 8, q, st); if (rc) return rc; }
    hipLaunchKernelGGL(k_sum_ranks, dim3(1), dim3(64), 0, st, (const double*)c->arStage, c->world, (int)count, (double*)recvbuf);
    return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();                    // teardown only: the mailboxes go away
    for (int q = 0; q < c->world; ++q) {
        if (q == c->rank) continue;
        if (c->peerMailbox[q]) (void)hipIpcCloseMemHandle(c->peerMailbox[q]);
        for (int ch = 0; ch < CH; ++ch)
            for (int s = 0; s < SLOTS; ++s) {
                if (c->myReady[q][ch][s]) (void)hipEventDestroy(c->myReady[q][ch][s]);
                if (c->myCons[q][ch][s]) (void)hipEventDestroy(c->myCons[q][ch][s]);
                if (c->peerReady[q][ch][s]) (void)hipEventDestroy(c->peerReady[q][ch][s]);
                if (c->peerCons[q][ch][s]) (void)hipEventDestroy(c->peerCons[q][ch][s]);
            }
    }
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
