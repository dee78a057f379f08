// Multi-GPU: distributed PCG over bricks / z-slabs (RCCL, host-staged TCP or in-process ranks) and its C ABI.
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once

// =====================================================================================================
// Multi-GPU: distributed PCG over a brick decomposition — z-slabs are its 1 x 1 x N case.  Not in the reference
// (single process).
// The same kernels as above run on every rank over its local rows / owned DOF range; what is added is
//   * pack / unpack of the one-layer exchange lists,
//   * a transport (RCCL send/recv + all-reduce on the solver stream, or device copies between ranks that
//     live in one process), and
//   * two-phase scalar kernels so that the all-reduce sits between "local sum" and "use".
// =====================================================================================================
#include <dlfcn.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>
#include <array>
#include <cstring>
#include <cstdio>

struct PsNcclUid { char internal[128]; };   // layout of ncclUniqueId

namespace {

// both cut planes of ONE axis in one launch: entries [0, nA) use list A / buffer A, entries [nA, nA + nB) list B / buffer B.
// (Along an axis a DOF lies next to at most one cut — a brick is at least one 16-cell block thick — so the two lists are disjoint.)
__global__ void k_pack2(const int32_t* __restrict__ listA, int64_t nA, double* __restrict__ bufA, const int32_t* __restrict__ listB, int64_t nB,
                        double* __restrict__ bufB, const double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nA) bufA[i] = v[listA[i]];
    else if (i < nA + nB) bufB[i - nA] = v[listB[i - nA]];
}
template <bool ADD>
__global__ void k_unpack2(const int32_t* __restrict__ listA, int64_t nA, const double* __restrict__ bufA, const int32_t* __restrict__ listB, int64_t nB,
                          const double* __restrict__ bufB, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nA) { if (ADD) v[listA[i]] += bufA[i]; else v[listA[i]] = bufA[i]; }
    else if (i < nA + nB) { if (ADD) v[listB[i - nA]] += bufB[i - nA]; else v[listB[i - nA]] = bufB[i - nA]; }
}
// one round: the lists of all three axes in one launch (entries [end[q - 1], end[q]) use list q / buffer q).  Values only: the halo samples of
// different axes are different samples (no corner copies in this mode), so the scatter has no conflicts; the ADDING unpack of the contributions
// stays one launch per axis — a DOF next to two cuts receives from both, in a fixed order.
constexpr int NLIST = 2 * ps_context::NLINK;
struct Lists6 { const int32_t* list[NLIST]; double* buf[NLIST]; int64_t end[NLIST]; };   // (two lists per link: below, above)
__global__ void k_pack6(Lists6 L, const double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.end[NLIST - 1]) return;
    int q = 0;
#pragma unroll
    for (int k = 0; k < NLIST - 1; ++k) q += i >= L.end[k] ? 1 : 0;
    const int64_t j = i - (q > 0 ? L.end[q - 1] : 0);
    L.buf[q][j] = v[L.list[q][j]];
}
__global__ void k_unpack6(Lists6 L, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.end[NLIST - 1]) return;
    int q = 0;
#pragma unroll
    for (int k = 0; k < NLIST - 1; ++k) q += i >= L.end[k] ? 1 : 0;
    const int64_t j = i - (q > 0 ? L.end[q - 1] : 0);
    v[L.list[q][j]] = L.buf[q][j];
}
// In-process groups: one exchange = ONE launch for all ranks and links.  Segment k moves n entries: dst[dstList[i]] (dstList null: dst[i]) = src[srcList[i]];
// the two lists of a cut hold the same samples in the same order (Dist::checkLists).  blk0 = the segment's first workgroup.
struct XSeg { const double* src; const int32_t* srcList; double* dst; const int32_t* dstList; int32_t n, blk0; };
__global__ void __launch_bounds__(BS) k_xchg_direct(const XSeg* __restrict__ segs, int nSeg) {
    __shared__ int sIdx;
    if ((int)threadIdx.x < nSeg) {
        const int b0 = segs[threadIdx.x].blk0, b1 = (int)threadIdx.x + 1 < nSeg ? segs[threadIdx.x + 1].blk0 : 0x7fffffff;
        if (b0 <= (int)blockIdx.x && (int)blockIdx.x < b1) sIdx = (int)threadIdx.x;
    }
    __syncthreads();
    const XSeg s = segs[sIdx];
    const int i = ((int)blockIdx.x - s.blk0) * BS + (int)threadIdx.x;
    if (i < s.n) {
        const double v = s.src[s.srcList[i]];
        if (s.dstList) s.dst[s.dstList[i]] = v; else s.dst[i] = v;
    }
}
// the received contributions whose DOF is not this rank's own (it sits on an earlier axis's upper halo plane: see buildHaloLists) are added
// to the rank's copy, which the exchange along that earlier axis passes on
__global__ void k_relay2(const int32_t* __restrict__ listA, int64_t nA, const double* __restrict__ bufA, const int32_t* __restrict__ listB, int64_t nB,
                         const double* __restrict__ bufB, double* __restrict__ v, int ownHi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nA + nB) return;
    const int j = i < nA ? listA[i] : listB[i - nA];
    if (j >= ownHi) v[j] += i < nA ? bufA[i] : bufB[i - nA];
}
// Exchange mode (Dist::decideExchangeMode): mark the halo samples the one-round lists deliver, then count the columns of this rank's rows that
// lie outside its owned range and are NOT marked — samples only a forwarded copy (a diagonal neighbour's) could bring
__global__ void k_mark_list(const int32_t* __restrict__ list, int64_t n, unsigned char* __restrict__ mark) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mark[list[i]] = 1;
}
__global__ void k_count_unmarked_halo(const int32_t* __restrict__ col, int64_t nnz, int ownLo, int ownHi, const unsigned char* __restrict__ mark, int32_t* __restrict__ misses) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int c = col[e];
    if ((c < ownLo || c >= ownHi) && !mark[c]) atomicAdd(misses, 1);
}
// Which chunks of S gather a halo value (a column outside the owned DOF range)?  One wave per chunk.
__global__ void __launch_bounds__(64) k_chunk_flags_S(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const int4* __restrict__ chunkInfo,
                                                      int ownLo, int ownHi, int32_t* __restrict__ flag) {
    const int4 ci = chunkInfo[blockIdx.x];
    const int p0 = ptr[ci.z], p1 = ptr[ci.z + (int)((unsigned)ci.y >> 16)];
    bool out = false;
    for (int e = p0 + (int)threadIdx.x; e < p1; e += 64) { const int c = col[e]; out |= c < ownLo || c >= ownHi; }
    if (__any(out) && threadIdx.x == 0) flag[blockIdx.x] = 1;
    else if (threadIdx.x == 0) flag[blockIdx.x] = 0;
}
// Chunks of St: 0 = halo rows without entries (nothing to do), 1 = owned rows ONLY (every row of the chunk inside [ownLo, ownHi): the
// launch compiled for that treats all of its rows as owned), 2 = holds halo rows — with entries (this rank's share of the neighbour's
// A p), or empty ones next to owned rows (a chunk that straddles ownHi: short lattice-block ranges are packed with their neighbours)
__global__ void k_chunk_flags_St(const int32_t* __restrict__ ptr, const int4* __restrict__ chunkInfo, int nChunks, int ownLo, int ownHi, int32_t* __restrict__ flag) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nChunks) return;
    const int4 ci = chunkInfo[c];
    const int r0 = ci.z, r1 = ci.z + (int)((unsigned)ci.y >> 16);
    const int lo = max(r0, ownLo), hi = min(r1, ownHi);
    const int all = ptr[r1] - ptr[r0], owned = hi > lo ? ptr[hi] - ptr[lo] : 0;
    const bool inside = r0 >= ownLo && r1 <= ownHi;
    flag[c] = (all - owned) > 0 ? 2 : (hi > lo ? (inside ? 1 : 2) : 0);
}
// A halo row that RECEIVES relayed contributions (k_relay2: v[j] += ... for j >= ownHi) must be rewritten by the St launch of every
// iteration, else the sums of earlier iterations stay in it: its chunk goes to the list whose launch stores the halo rows' y
__global__ void k_chunk_flags_relay(const int32_t* __restrict__ listA, int64_t nA, const int32_t* __restrict__ listB, int64_t nB, int ownHi,
                                    const int4* __restrict__ chunkInfo, int nChunks, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nA + nB) return;
    const int j = i < nA ? listA[i] : listB[i - nA];
    if (j < ownHi) return;
    int lo = 0, hi = nChunks - 1;                         // the last chunk whose first row is <= j (chunks are consecutive row ranges)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (chunkInfo[mid].z <= j) lo = mid; else hi = mid - 1; }
    const int4 ci = chunkInfo[lo];
    if (j >= ci.z && j < ci.z + (int)((unsigned)ci.y >> 16)) flag[lo] = 2;
}
// out[q] = sum of partial[q*stride .. q*stride+count)   (q < nq), one block
__global__ void __launch_bounds__(BS) k_sumq(const CGScalars* __restrict__ sc, const double* __restrict__ partial, int count, int stride, int nq,
                                             double* __restrict__ out) {
    if (sc && sc->done) return;
    for (int q = 0; q < nq; ++q) {
        const double s = sumPartials(partial + (int64_t)q * stride, count);
        if (threadIdx.x == 0) out[q] = s;
        __syncthreads();
    }
}
__global__ void k_dscal0(CGScalars* sc, const double* __restrict__ red, double tol, int maxit, int vecNT) {
    const double s = red[0];
    sc->rsold = s; sc->rsold2[0] = s; sc->rsold2[1] = 0.; sc->rre = 0.; sc->iter = maxit; sc->maxit = maxit; sc->tol2 = tol * tol;
    sc->done = (s == 0.) ? 1 : 0;
    if (s == 0.) sc->iter = 0;
    sc->alpha = sc->beta = sc->pAp = sc->rr = sc->xx = sc->rz = 0.;
    sc->pend = 0; sc->pendIter = 0; sc->vecNT = vecNT;
}
__global__ void k_invert_diag(double* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = d[i];
        d[i] = v != 0. ? 1. / v : 1.;
    }
}
// ---- the cell labels of a rank's halo blocks, taken from their owners (Dist::exchangeLabels) ----------------------------------------
// Layers [q0, q0 + h) along axis a over the FULL cross-section of the local grid (the halo parts of the other axes included: the
// axes are exchanged x, y, z, each forwarding what the previous brought, as the values of the solve are); entry j = layer * cs + u2 * d1 + u1
// with (u1, u2) the two other coordinates in ascending axis order.  Two neighbours along a have the same cross-section.
__device__ inline int64_t labelCell(const int3 d, int a, int q0, int64_t j) {
    const int d1 = a == 0 ? d.y : d.x, d2 = a == 2 ? d.y : d.z;
    const int64_t cs = (int64_t)d1 * d2;
    const int layer = (int)(j / cs);
    const int64_t rem = j - (int64_t)layer * cs;
    const int u1 = (int)(rem % d1), u2 = (int)(rem / d1);
    const int q = q0 + layer;
    return a == 0 ? lin3(d, q, u1, u2) : (a == 1 ? lin3(d, u1, q, u2) : lin3(d, u1, u2, q));
}
__global__ void k_labels_pack(const int32_t* __restrict__ lab, int3 d, int a, int qLo, int64_t nLo, int32_t* __restrict__ outLo, int qUp, int64_t nUp,
                              int32_t* __restrict__ outUp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nLo + nUp; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nLo) outLo[i] = lab[labelCell(d, a, qLo, i)];
        else outUp[i - nLo] = lab[labelCell(d, a, qUp, i - nLo)];
    }
}
// reg != null (after fixReducedRegionBoundaries): a cell the owner kept REDUCED has the component it had before the fix (reg0), any
// other cell none.  flags[0] += labels changed, flags[1] += REDUCED cells without a component (the views disagree before the fix: refused)
// pass 0 (reg == null): nbrMask bit 2b / 2b+1 = a neighbour below / above along axis b; a label that changes `reach` or more cells
// away from every end of the view that is a cut cannot come from the classification's reach: the ranks were handed different
// fields for the same cells (flags[2])
__global__ void k_labels_unpack(int32_t* __restrict__ lab, int32_t* __restrict__ reg, const int32_t* __restrict__ reg0, int3 d, int a, int qLo, int64_t nLo,
                                const int32_t* __restrict__ inLo, int qUp, int64_t nUp, const int32_t* __restrict__ inUp, int32_t* __restrict__ flags,
                                int nbrMask, int reach) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nLo + nUp; i += (int64_t)gridDim.x * blockDim.x) {
        const bool low = i < nLo;
        const int64_t c = low ? labelCell(d, a, qLo, i) : labelCell(d, a, qUp, i - nLo);
        const int32_t v = low ? inLo[i] : inUp[i - nLo];
        if (lab[c] != v) {
            atomicAdd(&flags[0], 1);
            lab[c] = v;
            if (!reg) {
                const int3 q = unlin3(d, c);
                const int qq[3] = {q.x, q.y, q.z}, dd[3] = {d.x, d.y, d.z};
                int dist = INT_MAX;
                for (int b = 0; b < 3; ++b) {
                    if (nbrMask & (1 << (2 * b))) dist = min(dist, qq[b]);
                    if (nbrMask & (2 << (2 * b))) dist = min(dist, dd[b] - 1 - qq[b]);
                }
                if (dist >= reach) atomicAdd(&flags[2], 1);
            }
        }
        if (reg) {
            const int32_t r = v == PS_REDUCED ? reg0[c] : (int32_t)PS_UNASSIGNED;
            if (v == PS_REDUCED && r < 0) atomicAdd(&flags[1], 1);
            reg[c] = r;
        }
    }
}

struct SumPtrs { double* p[16]; int n; };
__global__ void k_sum_across(SumPtrs P, int count) {
    const int i = threadIdx.x;
    if (i >= count) return;
    double s = 0.;
    for (int q = 0; q < P.n; ++q) s += P.p[q][i];
    for (int q = 0; q < P.n; ++q) P.p[q][i] = s;
}
// faces this rank is responsible for in the output fields
__global__ void k_owned_faces(Grid g, int axis, Own own, const int32_t* __restrict__ faceRow, const int32_t* __restrict__ reg,
                              const int32_t* __restrict__ regionOwned, float* __restrict__ out) {
    const int3 d = g.dims(1 + axis);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    bool mine;
    const int r = reg[c];
    if (faceRow[c] >= 0) mine = true;
    else if (r >= 0 && regionOwned) mine = regionOwned[r] != 0;
    else mine = own.sample(1 + axis, q.x, q.y, q.z);
    out[c] = mine ? 1.f : 0.f;
}

// ---- RCCL through dlopen (no link-time dependency; torch.distributed only bootstraps the unique id) ----
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /*ncclUniqueId by value*/ PsNcclUid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void*) = nullptr;
};
Rccl& rccl() {
    // loaded once, by the first caller, under the compiler's guard of a function-local static: several contexts of one process may reach
    // this from different threads at the same time (a throwing first attempt is retried by the next caller)
    static Rccl loaded = [] {
        Rccl R;
        // PS_RCCL_LIB (tests only): another library exporting the same eight entry points — tests/stub_rccl, the stand-in that lets
        // this asynchronous branch run with several ranks on ONE GPU (real RCCL refuses duplicate devices)
        if (const char* alt = PS_ENV_LOUD("PS_RCCL_LIB")) {
            R.h = dlopen(alt, RTLD_NOW | RTLD_LOCAL);
            if (!R.h) throw Error(std::string("cannot load PS_RCCL_LIB: ") + dlerror());
        }
        // a copy the process has already mapped (torch links its own) must be THE copy: two RCCLs on one device abort at exit
        if (!R.h) R.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!R.h) R.h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!R.h) R.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!R.h) R.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!R.h) throw Error(std::string("cannot load librccl: ") + dlerror());
        auto sym = [&](const char* n) { void* p = dlsym(R.h, n); if (!p) throw Error(std::string("librccl lacks ") + n); return p; };
        R.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        R.CommInitRank = (int (*)(void**, int, PsNcclUid, int))sym("ncclCommInitRank");
        R.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        R.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        R.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        R.GroupStart = (int (*)())sym("ncclGroupStart");
        R.GroupEnd = (int (*)())sym("ncclGroupEnd");
        R.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        return R;
    }();
    return loaded;
}
// ---- host-staged transport: TCP sockets between the ranks' processes (pack -> D2H -> socket -> H2D -> unpack) ----
// The same distributed algorithm without RCCL: for boxes where RCCL cannot run (several ranks on one GPU: RCCL refuses
// duplicate devices) and as the fallback when librccl is absent.  Rank r listens on port base + r; it connects to rank r-1
// (halo exchange) and, for the scalar all-reduce, to rank 0 (star: rank 0 adds the contributions in rank order).
struct HostComm {
    int rank = 0, world = 1;
    int fdListen = -1, fdRoot = -1;
    static constexpr int NL = ps_context::NLINK;
    int fdLo[NL] = {-1, -1, -1, -1, -1, -1}, fdUp[NL] = {-1, -1, -1, -1, -1, -1};   // the neighbour below / above of every link (faces 0..2, diagonals 3..5; connected once the brick is known)
    int nbrLo[NL] = {-1, -1, -1, -1, -1, -1}, nbrUp[NL] = {-1, -1, -1, -1, -1, -1};
    std::string host; int basePort = 0;
    std::vector<int> fdLeaf;            // rank 0: connection of every other rank (index = rank)
    std::vector<std::array<int, 3>> pending;   // accepted connections nobody has asked for yet: (fd, rank, kind)
    std::vector<double> hs[2 * NL], hr[2 * NL];
    ~HostComm() {
        for (int fd : {fdListen, fdRoot}) if (fd >= 0) ::close(fd);
        for (int a = 0; a < NL; ++a) { if (fdLo[a] >= 0) ::close(fdLo[a]); if (fdUp[a] >= 0) ::close(fdUp[a]); }
        for (int fd : fdLeaf) if (fd >= 0) ::close(fd);
        for (auto& q : pending) ::close(q[0]);
    }
    static void sendAll(int fd, const void* p, size_t n) {
        const char* b = (const char*)p;
        while (n) { const ssize_t k = ::send(fd, b, n, MSG_NOSIGNAL); if (k <= 0) throw Error("TCP transport: send failed (peer gone?)"); b += k; n -= (size_t)k; }
    }
    static void recvAll(int fd, void* p, size_t n) {
        char* b = (char*)p;
        while (n) {
            pollfd pf{fd, POLLIN, 0};
            const int pr = ::poll(&pf, 1, 120000);   // a dead peer must not block this rank for ever
            if (pr <= 0) throw Error("TCP transport: timed out waiting for a peer");
            const ssize_t k = ::recv(fd, b, n, 0);
            if (k <= 0) throw Error("TCP transport: receive failed (peer gone?)");
            b += k; n -= (size_t)k;
        }
    }
    static int connectTo(const char* host, int port, int rank, int kind) {
        for (int attempt = 0; attempt < 1200; ++attempt) {   // the peer may not listen yet: retry for up to 60 s
            const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd < 0) throw Error("TCP transport: socket() failed");
            sockaddr_in a{};
            a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
            if (::inet_pton(AF_INET, host, &a.sin_addr) != 1) { ::close(fd); throw Error("TCP transport: bad host address (dotted IPv4 expected)"); }
            if (::connect(fd, (sockaddr*)&a, sizeof(a)) == 0) {
                const int one = 1;
                ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                const int32_t hello[2] = {rank, kind};
                sendAll(fd, hello, sizeof(hello));
                return fd;
            }
            ::close(fd);
            ::usleep(50000);
        }
        throw Error("TCP transport: cannot connect to port " + std::to_string(port));
    }
    // the connection (from, kind): kind 1 = a rank's link to rank 0 (scalar all-reduce), 10 + a = the upper neighbour along axis a
    // calling.  Connections arrive in any order: the others wait in `pending` until their turn.
    int acceptFrom(int from, int kind) {
        for (size_t q = 0; q < pending.size(); ++q)
            if (pending[q][1] == from && pending[q][2] == kind) { const int fd = pending[q][0]; pending.erase(pending.begin() + (long)q); return fd; }
        for (;;) {
            pollfd pf{fdListen, POLLIN, 0};
            if (::poll(&pf, 1, 120000) <= 0) throw Error("TCP transport: timed out waiting for the other ranks to connect");
            const int fd = ::accept(fdListen, nullptr, nullptr);
            if (fd < 0) throw Error("TCP transport: accept() failed");
            const int one = 1;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            int32_t hello[2];
            recvAll(fd, hello, sizeof(hello));
            if (hello[0] == from && hello[1] == kind) return fd;
            if (hello[0] < 0 || hello[0] >= world || pending.size() > 256) { ::close(fd); throw Error("TCP transport: unexpected connection"); }
            pending.push_back({fd, hello[0], hello[1]});
        }
    }
    void init(int r, int w, const char* h, int port) {
        rank = r; world = w; host = h; basePort = port;
        fdLeaf.assign((size_t)w, -1);
        fdListen = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fdListen < 0) throw Error("TCP transport: socket() failed");
        const int one = 1;
        ::setsockopt(fdListen, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{};
        a.sin_family = AF_INET; a.sin_port = htons((uint16_t)(basePort + r)); a.sin_addr.s_addr = htonl(INADDR_ANY);
        if (::bind(fdListen, (sockaddr*)&a, sizeof(a)) != 0 || ::listen(fdListen, 4 * w + 8) != 0)
            throw Error("TCP transport: cannot listen on port " + std::to_string(basePort + r));
        if (r > 0) fdRoot = connectTo(host.c_str(), basePort, r, 1);
        else for (int q = 1; q < w; ++q) fdLeaf[(size_t)q] = acceptFrom(q, 1);
    }
    // links to the face neighbours (lo[a] / up[a]: their ranks, -1 = none): a rank connects to its lower neighbours and accepts its upper ones
    void connectNeighbours(const int* lo, const int* up) {
        bool same = true;
        for (int a = 0; a < NL; ++a) same = same && lo[a] == nbrLo[a] && up[a] == nbrUp[a];
        if (same) return;
        for (int a = 0; a < NL; ++a) {
            if (fdLo[a] >= 0) { ::close(fdLo[a]); fdLo[a] = -1; }
            if (fdUp[a] >= 0) { ::close(fdUp[a]); fdUp[a] = -1; }
            nbrLo[a] = lo[a]; nbrUp[a] = up[a];
        }
        for (int a = 0; a < NL; ++a) if (lo[a] >= 0) fdLo[a] = connectTo(host.c_str(), basePort + lo[a], rank, 10 + a);
        for (int a = 0; a < NL; ++a) if (up[a] >= 0) fdUp[a] = acceptFrom(up[a], 10 + a);
    }
    // neighbour exchange along one axis, ordered so that the chain of ranks along it cannot deadlock: with the lower neighbour receive
    // first, with the upper send first
    void exchange(int a, const double* sLo, size_t nsLo, double* rLo, size_t nrLo, const double* sUp, size_t nsUp, double* rUp, size_t nrUp) {
        if (fdLo[a] >= 0) { if (nrLo) recvAll(fdLo[a], rLo, nrLo * 8); if (nsLo) sendAll(fdLo[a], sLo, nsLo * 8); }
        if (fdUp[a] >= 0) { if (nsUp) sendAll(fdUp[a], sUp, nsUp * 8); if (nrUp) recvAll(fdUp[a], rUp, nrUp * 8); }
    }
    void allreduceSum(double* v, int count) {
        if (world == 1) return;
        if (rank > 0) { sendAll(fdRoot, v, (size_t)count * 8); recvAll(fdRoot, v, (size_t)count * 8); return; }
        std::vector<double> t((size_t)count);
        for (int q = 1; q < world; ++q) { recvAll(fdLeaf[(size_t)q], t.data(), (size_t)count * 8); for (int i = 0; i < count; ++i) v[i] += t[(size_t)i]; }
        for (int q = 1; q < world; ++q) sendAll(fdLeaf[(size_t)q], v, (size_t)count * 8);
    }
};

constexpr int NCCL_DOUBLE = 8;   // ncclFloat64
constexpr int NCCL_SUM = 0;
void ncclCheck(int rc, const char* what) { if (rc != 0) throw Error(std::string("RCCL failure in ") + what + " (code " + std::to_string(rc) + ")"); }

struct Dist {
    std::vector<ps_context*> R;   // the ranks living in this process (1 with RCCL / TCP, `world` for an in-process group)
    bool useRccl = false;         // one process per GPU, RCCL
    bool useTcp = false;          // one process per rank, host-staged TCP (HostComm)
    HostComm* hc() const { return (HostComm*)R[0]->hostComm; }

    // Streams.  One process per rank (RCCL / TCP): the transports run on the rank's second stream (commStream), ordered against the
    // solver stream by events, so that the rows that do not need a halo value are computed while it travels.  In-process ranks share one
    // stream: commStream == stream, the same sequence without anything to overlap.
    void ensureStreams() {
        for (ps_context* c : R) {
            if (!(useRccl || useTcp)) { c->commStream = c->stream; continue; }
            if (!c->commStream || c->commStream == c->stream) {
                // highest priority: the transport's kernels must find CUs while the persistent S / St workgroups of the solver stream hold
                // all of them — they are dispatched as those retire, ahead of the solver stream's own next round
                int lo = 0, hi = 0;
                HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                HIP_CHECK(hipStreamCreateWithPriority(&c->commStream, hipStreamNonBlocking, hi));
            }
            for (int e = 0; e < 8; ++e) if (!c->distEv[e]) HIP_CHECK(hipEventCreate(&c->distEv[e]));   // (timing events among them: default flags)
        }
    }
    hipStream_t cs(ps_context* c, bool onComm) const { return (onComm && c->commStream) ? c->commStream : c->stream; }
    // everything queued on `from` so far happens before what is queued on `to` from now on
    void order(ps_context* c, int ev, bool mainToComm) {
        if (!c->commStream || c->commStream == c->stream) return;
        static const int skipEv = [] { const char* e = PS_ENV_LOUD("PS_DIST_SKIP_ORDER"); return e ? atoi(e) : -1; }();   // tests only: drop one ordering edge
        if (ev == skipEv) return;                                                                             // (a transport that hides the race proves nothing)
        hipStream_t from = mainToComm ? c->stream : c->commStream, to = mainToComm ? c->commStream : c->stream;
        HIP_CHECK(hipEventRecord(c->distEv[ev], from));
        HIP_CHECK(hipStreamWaitEvent(to, c->distEv[ev], 0));
    }
    // sizes: kind 0 = x exchange (send own layers, receive halo), kind 1 = y exchange (send halo contributions, receive for own); every
    // axis with a neighbour, all of them in ONE group of sends and receives (each face neighbour is its own xGMI link)
    static int64_t nSendLo(const ps_context* c, int kind, int a) { return c->linkLower(a) ? (kind == 0 ? c->nLowOwn[a] : c->nLowHalo[a]) : 0; }
    static int64_t nSendUp(const ps_context* c, int kind, int a) { return c->linkUpper(a) ? (kind == 0 ? c->nUpOwn[a] : c->nUpHalo[a]) : 0; }
    static int64_t nRecvLo(const ps_context* c, int kind, int a) { return c->linkLower(a) ? (kind == 0 ? c->nLowHalo[a] : c->nLowOwn[a]) : 0; }
    static int64_t nRecvUp(const ps_context* c, int kind, int a) { return c->linkUpper(a) ? (kind == 0 ? c->nUpHalo[a] : c->nUpOwn[a]) : 0; }
    void transport(int kind, bool onComm = false, int only = -1) {   // only >= 0: that axis alone
        if (useRccl) {
            ps_context* c = R[0];
            hipStream_t st = cs(c, onComm);
            Rccl& L = rccl();
            ncclCheck(L.GroupStart(), "ncclGroupStart");
            for (int a = 0; a < ps_context::NLINK; ++a) {
                if (only >= 0 && a != only) continue;
                const int64_t sLo = nSendLo(c, kind, a), sUp = nSendUp(c, kind, a), rLo = nRecvLo(c, kind, a), rUp = nRecvUp(c, kind, a);
                if (sLo) ncclCheck(L.Send(c->sendLo[a].p, (size_t)sLo, NCCL_DOUBLE, c->nbrLo(a), c->rcclComm, st), "ncclSend");
                if (rLo) ncclCheck(L.Recv(c->recvLo[a].p, (size_t)rLo, NCCL_DOUBLE, c->nbrLo(a), c->rcclComm, st), "ncclRecv");
                if (sUp) ncclCheck(L.Send(c->sendUp[a].p, (size_t)sUp, NCCL_DOUBLE, c->nbrUp(a), c->rcclComm, st), "ncclSend");
                if (rUp) ncclCheck(L.Recv(c->recvUp[a].p, (size_t)rUp, NCCL_DOUBLE, c->nbrUp(a), c->rcclComm, st), "ncclRecv");
            }
            ncclCheck(L.GroupEnd(), "ncclGroupEnd");
            return;
        }
        if (useTcp) {
            ps_context* c = R[0];
            HostComm& H = *hc();
            hipStream_t st = cs(c, onComm);
            for (int a = 0; a < ps_context::NLINK; ++a) {
                if (only >= 0 && a != only) continue;
                const size_t sLo = (size_t)nSendLo(c, kind, a), sUp = (size_t)nSendUp(c, kind, a);
                H.hs[2 * a].resize(sLo + 1); H.hs[2 * a + 1].resize(sUp + 1);
                H.hr[2 * a].resize((size_t)nRecvLo(c, kind, a) + 1); H.hr[2 * a + 1].resize((size_t)nRecvUp(c, kind, a) + 1);
                if (sLo) HIP_CHECK(hipMemcpyAsync(H.hs[2 * a].data(), c->sendLo[a].p, sLo * 8, hipMemcpyDeviceToHost, st));
                if (sUp) HIP_CHECK(hipMemcpyAsync(H.hs[2 * a + 1].data(), c->sendUp[a].p, sUp * 8, hipMemcpyDeviceToHost, st));
            }
            HIP_CHECK(hipStreamSynchronize(st));
            for (int a = 0; a < ps_context::NLINK; ++a)
                if (only < 0 || a == only) H.exchange(a, H.hs[2 * a].data(), (size_t)nSendLo(c, kind, a), H.hr[2 * a].data(), (size_t)nRecvLo(c, kind, a), H.hs[2 * a + 1].data(),
                           (size_t)nSendUp(c, kind, a), H.hr[2 * a + 1].data(), (size_t)nRecvUp(c, kind, a));
            for (int a = 0; a < ps_context::NLINK; ++a) {
                if (only >= 0 && a != only) continue;
                const size_t rLo = (size_t)nRecvLo(c, kind, a), rUp = (size_t)nRecvUp(c, kind, a);
                if (rLo) HIP_CHECK(hipMemcpyAsync(c->recvLo[a].p, H.hr[2 * a].data(), rLo * 8, hipMemcpyHostToDevice, st));
                if (rUp) HIP_CHECK(hipMemcpyAsync(c->recvUp[a].p, H.hr[2 * a + 1].data(), rUp * 8, hipMemcpyHostToDevice, st));
            }
            HIP_CHECK(hipStreamSynchronize(st));   // the host buffers are reused by the next exchange
            return;
        }
        for (size_t q = 0; q < R.size(); ++q) {   // in-process ranks share one stream: plain device copies
            ps_context* c = R[q];
            for (int a = 0; a < ps_context::NLINK; ++a) {
                if (only >= 0 && a != only) continue;
                const int64_t sLo = nSendLo(c, kind, a), sUp = nSendUp(c, kind, a);
                if (sLo) HIP_CHECK(hipMemcpyAsync(R[(size_t)c->nbrLo(a)]->recvUp[a].p, c->sendLo[a].p, (size_t)sLo * 8, hipMemcpyDeviceToDevice, c->stream));
                if (sUp) HIP_CHECK(hipMemcpyAsync(R[(size_t)c->nbrUp(a)]->recvLo[a].p, c->sendUp[a].p, (size_t)sUp * 8, hipMemcpyDeviceToDevice, c->stream));
            }
        }
    }
    // pack / unpack of one rank's lists along axis a (the two cuts of an axis in one launch: their lists are disjoint — a brick is at
    // least one 16-cell block thick).  own = true: the layers of mine the neighbours' rows touch; false: theirs my rows touch.
    void pack(ps_context* c, bool own, const double* v, hipStream_t st, int a) {
        const int64_t nA = own ? c->nLowOwn[a] : c->nLowHalo[a], nB = own ? c->nUpOwn[a] : c->nUpHalo[a];
        if (nA + nB > 0)
            hipLaunchKernelGGL(k_pack2, dim3(gridFor(nA + nB, BS)), dim3(BS), 0, st, (own ? c->listLowOwn[a] : c->listLowHalo[a]).p, nA, c->sendLo[a].p,
                               (own ? c->listUpOwn[a] : c->listUpHalo[a]).p, nB, c->sendUp[a].p, v);
    }
    template <bool ADD>
    void unpack(ps_context* c, bool own, double* v, hipStream_t st, int a) {
        const int64_t nA = own ? c->nLowOwn[a] : c->nLowHalo[a], nB = own ? c->nUpOwn[a] : c->nUpHalo[a];
        if (nA + nB > 0)
            hipLaunchKernelGGL(k_unpack2<ADD>, dim3(gridFor(nA + nB, BS)), dim3(BS), 0, st, (own ? c->listLowOwn[a] : c->listLowHalo[a]).p, nA, c->recvLo[a].p,
                               (own ? c->listUpOwn[a] : c->listUpHalo[a]).p, nB, c->recvUp[a].p, v);
    }
    // all axes of one rank at once (the one-round mode): own = true: my cut layers -> send buffers (pack) / false: my halo samples
    Lists6 lists6(ps_context* c, bool own, bool send) const {
        Lists6 L;
        int64_t run = 0;
        for (int a = 0; a < ps_context::NLINK; ++a) {
            L.list[2 * a] = (own ? c->listLowOwn[a] : c->listLowHalo[a]).p; L.list[2 * a + 1] = (own ? c->listUpOwn[a] : c->listUpHalo[a]).p;
            L.buf[2 * a] = send ? c->sendLo[a].p : c->recvLo[a].p; L.buf[2 * a + 1] = send ? c->sendUp[a].p : c->recvUp[a].p;
            run += own ? c->nLowOwn[a] : c->nLowHalo[a]; L.end[2 * a] = run;
            run += own ? c->nUpOwn[a] : c->nUpHalo[a]; L.end[2 * a + 1] = run;
        }
        return L;
    }
    void packAll(ps_context* c, bool own, const double* v, hipStream_t st) {
        const Lists6 L = lists6(c, own, true);
        if (L.end[NLIST - 1] > 0) hipLaunchKernelGGL(k_pack6, dim3(gridFor(L.end[NLIST - 1], BS)), dim3(BS), 0, st, L, v);
    }
    void unpackAllValues(ps_context* c, double* v, hipStream_t st) {
        const Lists6 L = lists6(c, false, false);
        if (L.end[NLIST - 1] > 0) hipLaunchKernelGGL(k_unpack6, dim3(gridFor(L.end[NLIST - 1], BS)), dim3(BS), 0, st, L, v);
    }
    bool axisUsed(int a) const { for (const ps_context* c : R) if (c->linkLower(a) || c->linkUpper(a)) return true; return false; }   // (a: a link, 0 .. NLINK - 1)
    // The exchanges run axis after axis, each one forwarding what the previous ones brought (ps_grid.hip: buildHaloLists): values x, y, z;
    // contributions z, y, x.  With cuts along one axis only (slabs) that is one pack / transport / unpack, as before.
    // values of the cut layers -> the neighbours' halo copies, on the stream `onComm` selects
    // ONE round (r05, the default: ps_context::haloForward false): every list holds samples of the sender's own, so the three axes are packed,
    // sent and unpacked together — with bricks 2 transports per iteration instead of 6, no relay kernels.  Legal whenever no row of any
    // rank reaches a sample of a DIAGONAL neighbour (decideExchangeMode checks it on the matrices; it happens only when a tile's skin rows lie on a
    // cut plane, i.e. with tilePadding 1): then the forwarding rounds below run as before.
    bool forwarding() const { return R[0]->haloForward; }
    // in-process group, one-round mode: the tables of k_xchg_direct for the two exchanges of the PCG iteration (values of p out, contributions of A p
    // back into the receive buffers the fix-up reads).  Rebuilt per setup (buildLists): the lists and the vectors may have been re-allocated.
    bool inProcess() const { return !useRccl && !useTcp && R.size() > 1; }
    void buildDirectExchange() {
        ps_context* c0 = R[0];
        c0->nXseg[0] = c0->nXseg[1] = 0;
        if (!inProcess() || forwarding() || R.size() * 2 * ps_context::NLINK > (size_t)BS) return;
        for (int kind = 0; kind < 2; ++kind) {
            std::vector<XSeg> segs;
            int blk = 0;
            auto add = [&](const double* src, const int32_t* sl, double* dst, const int32_t* dl, int64_t n) {
                if (n <= 0) return;
                segs.push_back(XSeg{src, sl, dst, dl, (int32_t)n, blk});
                blk += gridFor(n, BS);
            };
            for (ps_context* c : R)
                for (int a = 0; a < ps_context::NLINK; ++a) {
                    if (c->linkLower(a)) {
                        ps_context* nb = R[(size_t)c->nbrLo(a)];
                        if (kind == 0) add(c->pvec.p, c->listLowOwn[a].p, nb->pvec.p, nb->listUpHalo[a].p, c->nLowOwn[a]);
                        else add(c->Ap.p, c->listLowHalo[a].p, nb->recvUp[a].p, nullptr, c->nLowHalo[a]);
                    }
                    if (c->linkUpper(a)) {
                        ps_context* nb = R[(size_t)c->nbrUp(a)];
                        if (kind == 0) add(c->pvec.p, c->listUpOwn[a].p, nb->pvec.p, nb->listLowHalo[a].p, c->nUpOwn[a]);
                        else add(c->Ap.p, c->listUpHalo[a].p, nb->recvLo[a].p, nullptr, c->nUpHalo[a]);
                    }
                }
            if (segs.empty() || segs.size() > (size_t)BS) continue;
            c0->xsegTab[kind].alloc(segs.size() * sizeof(XSeg));
            HIP_CHECK(hipMemcpyAsync(c0->xsegTab[kind].p, segs.data(), segs.size() * sizeof(XSeg), hipMemcpyHostToDevice, c0->stream));
            HIP_CHECK(hipStreamSynchronize(c0->stream));     // (the host vector goes out of scope)
            c0->nXseg[kind] = (int)segs.size(); c0->xsegBlocks[kind] = blk;
        }
    }
    bool directExchange(int kind) {
        ps_context* c0 = R[0];
        if (c0->nXseg[kind] <= 0) return false;
        hipLaunchKernelGGL(k_xchg_direct, dim3((unsigned)c0->xsegBlocks[kind]), dim3(BS), 0, c0->stream, (const XSeg*)c0->xsegTab[kind].p, c0->nXseg[kind]);
        return true;
    }
    void valuesOut(DevBuf<double> ps_context::*vec, bool onComm) {
        if (!forwarding()) {
            if (inProcess() && vec == &ps_context::pvec && directExchange(0)) return;
            for (ps_context* c : R) packAll(c, true, (c->*vec).p, cs(c, onComm));
            transport(0, onComm);
            for (ps_context* c : R) unpackAllValues(c, (c->*vec).p, cs(c, onComm));
            return;
        }
        for (int a = 0; a < 3; ++a) {
            if (!axisUsed(a)) continue;
            for (ps_context* c : R) pack(c, true, (c->*vec).p, cs(c, onComm), a);
            transport(0, onComm, a);
            for (ps_context* c : R) unpack<false>(c, false, (c->*vec).p, cs(c, onComm), a);
        }
    }
    // the halo rows' contributions -> their owners.  addOwned: the owners add them into vec (b, the Jacobi diagonal, A p of the plain step);
    // else only the copies on the way are updated (the owners of the fused step correct r from the receive buffers: fixup)
    void contributionsBack(DevBuf<double> ps_context::*vec, bool onComm, bool addOwned) {
        if (!forwarding()) {   // one round: what arrives is for DOFs of this rank's own (added here in the order z, y, x of the forwarding rounds: the same sums)
            if (inProcess() && !addOwned && vec == &ps_context::Ap && directExchange(1)) return;
            for (ps_context* c : R) packAll(c, false, (c->*vec).p, cs(c, onComm));
            transport(1, onComm);
            if (addOwned)
                for (int a = ps_context::NLINK - 1; a >= 0; --a) if (axisUsed(a)) for (ps_context* c : R) unpack<true>(c, true, (c->*vec).p, cs(c, onComm), a);   // (diagonals first, then z, y, x)
            return;
        }
        for (int a = 2; a >= 0; --a) {
            if (!axisUsed(a)) continue;
            for (ps_context* c : R) pack(c, false, (c->*vec).p, cs(c, onComm), a);
            transport(1, onComm, a);
            for (ps_context* c : R) {
                if (addOwned) unpack<true>(c, true, (c->*vec).p, cs(c, onComm), a);
                else if (c->nLowOwn[a] + c->nUpOwn[a] > 0)
                    hipLaunchKernelGGL(k_relay2, dim3(gridFor(c->nLowOwn[a] + c->nUpOwn[a], BS)), dim3(BS), 0, cs(c, onComm), (const int32_t*)c->listLowOwn[a].p, c->nLowOwn[a],
                                       (const double*)c->recvLo[a].p, (const int32_t*)c->listUpOwn[a].p, c->nUpOwn[a], (const double*)c->recvUp[a].p, (c->*vec).p, (int)c->ownHi);
            }
        }
    }
    // r_j -= alpha * (the neighbours' share of (A p)_j) on the OWNED DOFs next to a cut, from the receive buffers of every axis (a DOF next
    // to two cuts is corrected twice, each launch sees the r the previous one left); partials of the changes of r.r / r.z: [axis][2][gFix] (zeroed once per solve: an axis
    // without lists leaves its part alone)
    // the merged fix-up list of a rank (k_dist_fixup_merged): every owned DOF that receives contributions, with its sources in link order
    // (diagonals last; a halo copy of the forwarding mode — index >= ownHi — is not a DOF of this rank and is left out, as k_dist_fixup skips it)
    void buildFixup(ps_context* c) {
        c->nFix = 0;
        std::vector<std::pair<int32_t, int32_t>> ent;                       // (DOF, (list index << 4) | buffer), in link order
        for (int a = 0; a < ps_context::NLINK; ++a)
            for (int side = 0; side < 2; ++side) {
                const std::vector<int32_t>& L = c->hostOwnList[2 * a + side];
                if (L.size() >= (size_t)(1 << 27)) return;                  // (would not fit the encoding: the per-link launches stay)
                for (size_t i = 0; i < L.size(); ++i) if (L[i] < (int32_t)c->ownHi) ent.push_back({L[i], (int32_t)((i << 4) | (size_t)(2 * a + side))});
            }
        if (ent.empty()) return;
        std::stable_sort(ent.begin(), ent.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first < y.first; });
        std::vector<int32_t> dof, cnt;
        for (size_t i = 0; i < ent.size(); ++i) { if (dof.empty() || dof.back() != ent[i].first) { dof.push_back(ent[i].first); cnt.push_back(0); } ++cnt.back(); }
        for (int32_t k : cnt) if (k > FIX_MAXSRC) return;                   // (cannot happen with six links; the per-link launches stay if it does)
        const size_t n = dof.size();
        std::vector<int32_t> src((size_t)FIX_MAXSRC * n, -1);
        size_t e = 0;
        for (size_t i = 0; i < n; ++i) for (int32_t k = 0; k < cnt[i]; ++k) src[(size_t)k * n + i] = ent[e++].second;
        std::vector<const double*> bufs(12);
        for (int a = 0; a < ps_context::NLINK; ++a) { bufs[(size_t)2 * a] = c->recvLo[a].p; bufs[(size_t)2 * a + 1] = c->recvUp[a].p; }
        c->fixDof.alloc(n); c->fixSrc.alloc(src.size()); c->fixBufs.alloc(12);
        HIP_CHECK(hipMemcpyAsync(c->fixDof.p, dof.data(), n * 4, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->fixSrc.p, src.data(), src.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->fixBufs.p, bufs.data(), 12 * sizeof(const double*), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));                          // (the host vectors go out of scope)
        c->nFix = (int64_t)n;
    }
    void fixup(ps_context* c, const CGScalars* sc, bool jac, double* fX, int gFix) {
        if (c->nFix > 0) {                                                   // one launch; its partials are the first of the sets k_sum_rr adds up (the others stay zero)
            hipLaunchKernelGGL(k_dist_fixup_merged, dim3(gFix), dim3(BS), 0, c->stream, sc, (const int32_t*)c->fixDof.p, (const int32_t*)c->fixSrc.p, c->nFix,
                               (const double* const*)c->fixBufs.p, c->r.p, jac ? (const diag_t*)c->dinvF.p : (const diag_t*)nullptr, fX);
            return;
        }
        for (int a = 0; a < ps_context::NLINK; ++a) {
            double* part = fX + (size_t)a * 2 * (size_t)gFix;
            if (c->nLowOwn[a] + c->nUpOwn[a] > 0)
                hipLaunchKernelGGL(k_dist_fixup, dim3(gFix), dim3(BS), 0, c->stream, sc, (const int32_t*)c->listLowOwn[a].p, c->nLowOwn[a], (const double*)c->recvLo[a].p,
                                   (const int32_t*)c->listUpOwn[a].p, c->nUpOwn[a], (const double*)c->recvUp[a].p, c->r.p,
                                   jac ? (const diag_t*)c->dinvF.p : (const diag_t*)nullptr, part, (int)c->ownHi);
        }
    }
    void exchangeX(DevBuf<double> ps_context::*vec) {
        for (ps_context* c : R) order(c, 0, true);
        valuesOut(vec, true);
        for (ps_context* c : R) order(c, 1, false);
    }
    void exchangeAddY(DevBuf<double> ps_context::*vec) {
        for (ps_context* c : R) order(c, 0, true);
        contributionsBack(vec, true, true);
        for (ps_context* c : R) order(c, 1, false);
    }
    void allreduce(int count) {
        if (useRccl) {   // every RCCL call of the communicator goes to ONE stream (the comm stream once it exists)
            ps_context* c = R[0];
            order(c, 6, true);
            ncclCheck(rccl().AllReduce(c->redbuf.p, c->redbuf.p, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->rcclComm, cs(c, true)), "ncclAllReduce");
            order(c, 7, false);
            return;
        }
        if (useTcp) {
            ps_context* c = R[0];
            double v[8];
            HIP_CHECK(hipMemcpyAsync(v, c->redbuf.p, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            hc()->allreduceSum(v, count);
            HIP_CHECK(hipMemcpyAsync(c->redbuf.p, v, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            return;
        }
        if (R.size() == 1) return;
        SumPtrs P;
        P.n = (int)R.size();
        for (size_t q = 0; q < R.size(); ++q) P.p[q] = R[q]->redbuf.p;
        hipLaunchKernelGGL(k_sum_across, dim3(1), dim3(64), 0, R[0]->stream, P, count);
    }
    void syncAll() { for (ps_context* c : R) HIP_CHECK(hipStreamSynchronize(c->stream)); }

    // a flag summed over all ranks (setup failures, interrupts): every rank learns that some rank wants to stop
    double sumFlag(double mine) {
        for (ps_context* c : R) HIP_CHECK(hipMemcpyAsync(c->redbuf.p, &mine, 8, hipMemcpyHostToDevice, c->stream));
        if (!useRccl && !useTcp) return mine * (double)R.size();   // in-process ranks: the caller already knows
        allreduce(1);
        double out = 0.;
        HIP_CHECK(hipMemcpyAsync(&out, R[0]->redbuf.p, 8, hipMemcpyDeviceToHost, R[0]->stream));
        HIP_CHECK(hipStreamSynchronize(R[0]->stream));
        return out;
    }
    // The cell labels of every rank's halo blocks := the owners' labels.  The reference's classification is not local: the boundary layers
    // reach up to three cells, fixReducedRegionBoundaries (Classifier.cpp:1073-1172) one cell beyond a region's tile — beyond the halo
    // block — and fixSmallReducedRegions (:1174-1262) turns a cell demoted THERE into a whole region kept or dropped next to the cut
    // (tilePadding 1: scripts/brick_diag.py).  A rank's OWNED cells are at least a halo block away from where its view ends, so the
    // owner is always right.  pass 0: after classifyCells / constructReducedRegions (labels only); pass 1: after the boundary fix
    // (labels + the cell's component from before the fix).  Both go through the transport of the solve's x-exchange.
    // Before the first label transport: the two ranks of every cut agree on what they are about to exchange.  The message sizes of
    // exchangeLabels are derived on each side from its own brick (halo layers x cross-section) with no length in the message: ranks
    // configured with different tile sizes or cross-sections would wait in a receive for ever or read each other's labels at the wrong
    // cells.  One double per side and axis — (cross-section, halo layers, tileSize), exact in a double — through the same transport,
    // compared with what this rank expects from that side; any mismatch fails EVERY rank with a message.
    static double cutKey(const ps_context* c, int a, int layers) {
        const int3 d = c->g.dims(0);
        const int b = a == 0 ? d.y : d.x, e = a == 2 ? d.y : d.z;
        return (((double)b * 4096. + (double)e) * 64. + (double)layers) * 64. + (double)c->P.tileSize;      // < 2^42
    }
    void handshakeCuts() {
        bool bad = false;
        for (int a = 0; a < 3; ++a) {
            if (!axisUsed(a)) continue;
            std::vector<int64_t> keep(R.size() * 4);
            std::vector<double> want(R.size() * 2);
            for (size_t q = 0; q < R.size(); ++q) {
                ps_context* c = R[q];
                const int3 d = c->g.dims(0);
                const int da = a == 0 ? d.x : (a == 1 ? d.y : d.z);
                c->sendLo[a].alloc(8); c->sendUp[a].alloc(8); c->recvLo[a].alloc(8); c->recvUp[a].alloc(8);
                want[2 * q] = c->brick.hasLower[a] ? cutKey(c, a, c->brick.lo[a]) : 0.;
                want[2 * q + 1] = c->brick.hasUpper[a] ? cutKey(c, a, da - c->brick.hi[a]) : 0.;
                HIP_CHECK(hipMemcpyAsync(c->sendLo[a].p, &want[2 * q], 8, hipMemcpyHostToDevice, cs(c, true)));
                HIP_CHECK(hipMemcpyAsync(c->sendUp[a].p, &want[2 * q + 1], 8, hipMemcpyHostToDevice, cs(c, true)));
                keep[4 * q] = c->nLowOwn[a]; keep[4 * q + 1] = c->nLowHalo[a]; keep[4 * q + 2] = c->nUpOwn[a]; keep[4 * q + 3] = c->nUpHalo[a];
                c->nLowOwn[a] = c->nLowHalo[a] = c->nUpOwn[a] = c->nUpHalo[a] = 1;
            }
            auto restore = [&]() { for (size_t q = 0; q < R.size(); ++q) { ps_context* c = R[q]; c->nLowOwn[a] = keep[4 * q]; c->nLowHalo[a] = keep[4 * q + 1]; c->nUpOwn[a] = keep[4 * q + 2]; c->nUpHalo[a] = keep[4 * q + 3]; } };
            try { transport(0, true, a); } catch (...) { restore(); throw; }
            restore();
            for (size_t q = 0; q < R.size(); ++q) {
                ps_context* c = R[q];
                double got[2] = {0., 0.};
                if (c->brick.hasLower[a]) HIP_CHECK(hipMemcpyAsync(&got[0], c->recvLo[a].p, 8, hipMemcpyDeviceToHost, cs(c, true)));
                if (c->brick.hasUpper[a]) HIP_CHECK(hipMemcpyAsync(&got[1], c->recvUp[a].p, 8, hipMemcpyDeviceToHost, cs(c, true)));
                HIP_CHECK(hipStreamSynchronize(cs(c, true)));
                bad = bad || (c->brick.hasLower[a] && got[0] != want[2 * q]) || (c->brick.hasUpper[a] && got[1] != want[2 * q + 1]);
            }
        }
        if (sumFlag(bad ? 1. : 0.) > 0.)
            throw Error(bad ? "the rank across a cut was configured differently (cross-section, halo layers or tileSize of the two bricks disagree): refusing to exchange labels"
                            : "another pair of ranks disagrees about the cut between their bricks (cross-section, halo layers or tileSize)");
    }
    static int nbrMask(const ps_context* c) { int m = 0; for (int b = 0; b < 3; ++b) m |= (c->brick.hasLower[b] ? 1 : 0) << (2 * b) | (c->brick.hasUpper[b] ? 2 : 0) << (2 * b); return m; }
    void exchangeLabels(int pass) {
        bool any = false;
        for (int a = 0; a < 3; ++a) any = any || axisUsed(a);
        if (!any) return;
        if (useTcp) { ps_context* c = R[0]; int lo[ps_context::NLINK], up[ps_context::NLINK]; for (int a = 0; a < ps_context::NLINK; ++a) { lo[a] = c->nbrLo(a); up[a] = c->nbrUp(a); } hc()->connectNeighbours(lo, up); }
        for (ps_context* c : R) {
            c->labelFlags.alloc(4);
            HIP_CHECK(hipMemsetAsync(c->labelFlags.p, 0, 4 * sizeof(int32_t), c->stream));
            order(c, 0, true);
        }
        if (pass == 0) handshakeCuts();
        for (int a = 0; a < 3; ++a) {
            if (!axisUsed(a)) continue;
            struct Geo { int3 d; int hLo, hUp; int64_t nLo, nUp, keep[4]; };
            std::vector<Geo> G(R.size());
            for (size_t q = 0; q < R.size(); ++q) {
                ps_context* c = R[q];
                Geo& z = G[q];
                z.d = c->g.dims(0);
                const int da = a == 0 ? z.d.x : (a == 1 ? z.d.y : z.d.z);
                const int64_t cross = (int64_t)z.d.x * z.d.y * z.d.z / da;
                z.hLo = c->brick.hasLower[a] ? c->brick.lo[a] : 0;
                z.hUp = c->brick.hasUpper[a] ? da - c->brick.hi[a] : 0;
                z.nLo = (int64_t)z.hLo * cross; z.nUp = (int64_t)z.hUp * cross;
                z.keep[0] = c->nLowOwn[a]; z.keep[1] = c->nLowHalo[a]; z.keep[2] = c->nUpOwn[a]; z.keep[3] = c->nUpHalo[a];
                const size_t mx = (size_t)(std::max(z.nLo, z.nUp) + 1) / 2 + 8;   // int32 labels in the double buffers of the exchanges
                c->sendLo[a].alloc(mx); c->sendUp[a].alloc(mx); c->recvLo[a].alloc(mx); c->recvUp[a].alloc(mx);
                c->nLowOwn[a] = c->nLowHalo[a] = (z.nLo + 1) / 2; c->nUpOwn[a] = c->nUpHalo[a] = (z.nUp + 1) / 2;
                // what goes down: my first hLo owned layers (the lower rank's upper halo block); up: my last hUp owned layers
                if (z.nLo + z.nUp > 0)
                    hipLaunchKernelGGL(k_labels_pack, dim3(gridFor(z.nLo + z.nUp, BS)), dim3(BS), 0, cs(c, true), (const int32_t*)c->labels[0].p, z.d, a, c->brick.lo[a], z.nLo,
                                       (int32_t*)c->sendLo[a].p, c->brick.hi[a] - z.hUp, z.nUp, (int32_t*)c->sendUp[a].p);
            }
            auto restore = [&]() { for (size_t q = 0; q < R.size(); ++q) { ps_context* c = R[q]; c->nLowOwn[a] = G[q].keep[0]; c->nLowHalo[a] = G[q].keep[1]; c->nUpOwn[a] = G[q].keep[2]; c->nUpHalo[a] = G[q].keep[3]; } };
            try { transport(0, true, a); } catch (...) { restore(); throw; }
            restore();
            for (size_t q = 0; q < R.size(); ++q) {
                ps_context* c = R[q];
                const Geo& z = G[q];
                if (z.nLo + z.nUp > 0)
                    hipLaunchKernelGGL(k_labels_unpack, dim3(gridFor(z.nLo + z.nUp, BS)), dim3(BS), 0, cs(c, true), c->labels[0].p, pass ? c->reducedIdx[0].p : (int32_t*)nullptr,
                                       (const int32_t*)c->cellScratch[2].p, z.d, a, 0, z.nLo, (const int32_t*)c->recvLo[a].p, c->brick.hi[a], z.nUp,
                                       (const int32_t*)c->recvUp[a].p, c->labelFlags.p, nbrMask(c),
                                       c->P.activeLiquidBoundaryLayerSize + c->P.activeSolidBoundaryLayerSize + c->P.tilePadding + 2);   // (ps_set_brick: <= 16)
            }
        }
        bool bad = false, differ = false;
        for (ps_context* c : R) {
            order(c, 1, false);
            int32_t fl[3] = {0, 0, 0};
            HIP_CHECK(hipMemcpyAsync(fl, c->labelFlags.p, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (pass == 0) c->haloLabelChanges = 0;
            c->haloLabelChanges += fl[0];
            bad = bad || fl[1] != 0;
            differ = differ || fl[2] != 0;
        }
        if (sumFlag(bad || differ ? 1. : 0.) > 0.)
            throw Error(differ ? "the labels of a halo block differ from their owner's deeper inside the block than the classification reaches: the ranks were handed different fields for the same cells"
                        : (bad ? "the labels of a halo block cannot be reconciled with their owner's (REDUCED cells the rank's own classification never had)"
                               : "another rank found the labels of a halo block in disagreement with their owner's"));
    }
    // neighbours must agree on the exchange lists: same lengths AND the same keys (position in the cut's cross-section, kind) in the
    // same order (ps_context::buildHaloLists hashes them) — equal counts of different DOF sets would otherwise pair the wrong entries.
    void checkLists() {
        constexpr int NL = ps_context::NLINK;   // the three face links and the three diagonal links alike
        auto enc = [](int64_t n, uint64_t h, double* o) { o[0] = (double)n; o[1] = (double)(h & 0xffffffu); o[2] = (double)((h >> 24) & 0xffffffu); o[3] = (double)((h >> 48) & 0xffffu); };
        auto same = [](const double* a, const double* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3]; };
        if (!useRccl && !useTcp) {
            for (size_t q = 0; q < R.size(); ++q)
                for (int a = 0; a < NL; ++a) {
                    const ps_context* lo = R[q];
                    if (!lo->linkUpper(a)) continue;
                    const ps_context* up = R[(size_t)lo->nbrUp(a)];
                    if (lo->nUpHalo[a] != up->nLowOwn[a] || lo->nUpOwn[a] != up->nLowHalo[a] || lo->hashUpHalo[a] != up->hashLowOwn[a] || lo->hashUpOwn[a] != up->hashLowHalo[a])
                        throw Error("exchange lists disagree across the cut between ranks " + std::to_string(lo->brick.rank) + " and " + std::to_string(up->brick.rank));
                }
            return;
        }
        ps_context* c = R[0];
        if (useTcp) { int lo[NL], up[NL]; for (int a = 0; a < NL; ++a) { lo[a] = c->nbrLo(a); up[a] = c->nbrUp(a); } hc()->connectNeighbours(lo, up); }
        double mineLo[NL][8], mineUp[NL][8];
        int64_t keep[NL][4];
        for (int a = 0; a < NL; ++a) {
            enc(c->nLowOwn[a], c->hashLowOwn[a], mineLo[a]); enc(c->nLowHalo[a], c->hashLowHalo[a], mineLo[a] + 4);   // what I send down
            enc(c->nUpOwn[a], c->hashUpOwn[a], mineUp[a]); enc(c->nUpHalo[a], c->hashUpHalo[a], mineUp[a] + 4);       // what I send up
            HIP_CHECK(hipMemcpyAsync(c->sendLo[a].p, mineLo[a], 64, hipMemcpyHostToDevice, c->stream));
            HIP_CHECK(hipMemcpyAsync(c->sendUp[a].p, mineUp[a], 64, hipMemcpyHostToDevice, c->stream));
            keep[a][0] = c->nLowOwn[a]; keep[a][1] = c->nLowHalo[a]; keep[a][2] = c->nUpOwn[a]; keep[a][3] = c->nUpHalo[a];
            c->nLowOwn[a] = c->nLowHalo[a] = c->linkLower(a) ? 8 : 0; c->nUpOwn[a] = c->nUpHalo[a] = c->linkUpper(a) ? 8 : 0;   // ship 8 doubles each way through the x-exchange path
        }
        auto restore = [&]() { for (int a = 0; a < NL; ++a) { c->nLowOwn[a] = keep[a][0]; c->nLowHalo[a] = keep[a][1]; c->nUpOwn[a] = keep[a][2]; c->nUpHalo[a] = keep[a][3]; } };
        order(c, 0, true);
        try { transport(0, true); order(c, 1, false); } catch (...) { restore(); throw; }
        restore();
        // the lower rank's (UpHalo, UpOwn) must equal my (LowOwn, LowHalo); the upper rank's (LowHalo, LowOwn) my (UpOwn, UpHalo):
        // received from below: its (UpOwn, UpHalo); from above: its (LowOwn, LowHalo)
        bool bad = false;
        for (int a = 0; a < NL; ++a) {
            double lo[8] = {0}, up[8] = {0};
            if (c->linkLower(a)) HIP_CHECK(hipMemcpyAsync(lo, c->recvLo[a].p, 64, hipMemcpyDeviceToHost, c->stream));
            if (c->linkUpper(a)) HIP_CHECK(hipMemcpyAsync(up, c->recvUp[a].p, 64, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->linkLower(a) && !(same(lo, mineLo[a] + 4) && same(lo + 4, mineLo[a]))) bad = true;
            if (c->linkUpper(a) && !(same(up, mineUp[a] + 4) && same(up + 4, mineUp[a]))) bad = true;
        }
        if (sumFlag(bad ? 1. : 0.) > 0.) throw Error(bad ? "exchange lists disagree with a neighbour (labels differ across the cut: halo too thin for the layer sizes?)"
                                                          : "another rank found its exchange lists in disagreement");
    }

    using Vec = DevBuf<double> ps_context::*;
    // out = A in on the owned DOFs of every rank: halo values of `in` fetched, local products, halo contributions of `out` returned
    void applyDist(Vec in, Vec out) {
        exchangeX(in);
        for (ps_context* c : R) {
            Launch L = mk(c, nullptr);
            L.spmvS(0, (c->*in).p, c->ts.p);
            L.tiles(0, c->ts.p);
            L.spmvSt(0, c->ts.p, (c->*in).p, nullptr, (c->*out).p, c->dotPartials.p);
        }
        exchangeAddY(out);
    }
    // lambda_max(D^-1 A) for the Chebyshev interval, as ps_context::estimateLambdaMax but with the distributed operator and the
    // two sums reduced over the ranks: every rank ends with the same value
    void estimateLambdaMaxDist() {
        Vec V = &ps_context::tmp1, W = &ps_context::tmp2;
        const Vec AV = &ps_context::tmp3;
        for (ps_context* c : R) {
            const size_t nl = (size_t)std::max<int64_t>(c->nSystem, 1);
            c->tmp1.alloc(nl); c->tmp2.alloc(nl); c->tmp3.alloc(nl);
            c->chebPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor(std::max<int64_t>(c->nSystem, 1), BS)) + 16);
            c->dotPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor(std::max<int64_t>(c->nSystem, 1), BS)) + 16);
            hipLaunchKernelGGL(k_fill_f64, dim3(dotBlocks((int64_t)nl)), dim3(BS), 0, c->stream, c->tmp1.p, 1., (int64_t)nl);
            HIP_CHECK(hipMemsetAsync(c->tmp2.p, 0, nl * 8, c->stream));
        }
        for (int it = 0; it < 10; ++it) {
            applyDist(V, AV);
            for (ps_context* c : R) {
                const int64_t n = c->ownHi - c->ownLo, lo = c->ownLo;
                const int vb = dotBlocks(std::max<int64_t>(n, 1));
                hipLaunchKernelGGL(k_power_step, dim3(vb), dim3(BS), 0, c->stream, (const double*)(c->*V).p + lo, (const double*)c->tmp3.p + lo,
                                   (const double*)c->dinv.p + lo, (c->*W).p + lo, n, c->chebPartials.p);
            }
            std::swap(V, W);
        }
        for (ps_context* c : R) {
            const int vb = dotBlocks(std::max<int64_t>(c->ownHi - c->ownLo, 1));
            hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)nullptr, (const double*)c->chebPartials.p, vb, vb, 2, c->redbuf.p);
        }
        allreduce(2);
        double h[2] = {0., 0.};
        HIP_CHECK(hipMemcpyAsync(h, R[0]->redbuf.p, sizeof(h), hipMemcpyDeviceToHost, R[0]->stream));
        syncAll();
        const double lam = (h[0] > 0. && std::isfinite(h[1] / h[0])) ? h[1] / h[0] : 0.;
        for (ps_context* c : R) c->chebLmax = std::max(8.4, 1.25 * lam);
    }
    // z = q(D^-1 A) D^-1 r on the owned DOFs (ps_context::chebyshevApply with the distributed operator; the update of a term is
    // its own kernel here: A z is complete only after the halo contributions have come back).  Returns the vector holding z;
    // the partials of r.z of the final z are in chebPartials (dotBlocks(owned) of them per rank).
    Vec chebyshevDist(Vec Rv) {
        ps_context* c0 = R[0];
        const int k = c0->P.preconditionerDegree > 0 ? c0->P.preconditionerDegree : 4;
        const double lmax = c0->chebLmax, lmin = lmax / PS_CHEB_INTERVAL_RATIO;
        const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
        double rho = 1. / sigma;
        Vec cur = &ps_context::tmp1, other = &ps_context::tmp2;
        const Vec AZ = &ps_context::tmp5;
        for (ps_context* c : R) {
            const int64_t n = c->ownHi - c->ownLo, lo = c->ownLo;
            hipLaunchKernelGGL(k_cheb_first<double>, dim3(dotBlocks(std::max<int64_t>(n, 1))), dim3(BS), 0, c->stream, (const CGScalars*)c->scal.p, (const double*)(c->*Rv).p + lo,
                               (const diag_t*)c->dinvF.p + lo, 1. / theta, c->tmp1.p + lo, n, c->chebPartials.p);
        }
        for (int j = 1; j < k; ++j) {
            const double rhoN = 1. / (2. * sigma - rho);
            const double c1 = rhoN * rho, c2 = 2. * rhoN / delta;
            applyDist(cur, AZ);
            for (ps_context* c : R) {
                const int64_t n = c->ownHi - c->ownLo, lo = c->ownLo;
                hipLaunchKernelGGL(k_cheb_step, dim3(dotBlocks(std::max<int64_t>(n, 1))), dim3(BS), 0, c->stream, (const CGScalars*)c->scal.p, (const double*)(c->*Rv).p + lo,
                                   (const diag_t*)c->dinvF.p + lo, (const double*)c->tmp5.p + lo, c1, c2, (const double*)(c->*cur).p + lo,
                                   j == 1 ? (const double*)nullptr : (const double*)(c->*other).p + lo, (c->*other).p + lo, n, c->chebPartials.p);
            }
            std::swap(cur, other);
            rho = rhoN;
        }
        return cur;
    }

    // everything after the per-rank local setup: finish b and the Jacobi diagonal across the cuts; the Chebyshev interval
    // Chunk lists of the row-per-lane kernels (ps_context::distList): which chunks can run before the halo values have arrived /
    // while this rank's contributions to its neighbours travel.  Built from two flag kernels and a host pass per setup.
    void buildLists() {
        { int64_t all = 0; for (ps_context* c : R) all += c->nSystem; for (ps_context* c : R) c->deviceShareRows = inProcess() ? all : 0; }
        static const bool mergedFix = !(PS_ENV("PS_DIST_FIXUP_MERGED") && atoi(PS_ENV("PS_DIST_FIXUP_MERGED")) == 0);   // A/B: 0 = one k_dist_fixup launch per link
        for (ps_context* c : R) { c->nFix = 0; if (mergedFix) buildFixup(c); }
        for (ps_context* c : R) {
            c->distListsOk = false;
            for (int q = 0; q < 5; ++q) c->nDistList[q] = 0;
            Launch L = mk(c, nullptr);
            static const bool off = PS_ENV("PS_DIST_OVERLAP") && atoi(PS_ENV("PS_DIST_OVERLAP")) == 0;   // A/B: the sequential exchange
            if (off || !L.listsOk() || c->S.nChunks == 0 || c->St.nChunks == 0) continue;
            const int nS = c->S.nChunks, nT = c->St.nChunks;
            DevBuf<int32_t>& flags = c->scrVals;               // setup scratch
            flags.alloc((size_t)std::max(nS, nT));
            std::vector<int32_t> h((size_t)std::max(nS, nT)), lists[5];
            hipLaunchKernelGGL(k_chunk_flags_S, dim3((unsigned)nS), dim3(64), 0, c->stream, (const int32_t*)c->S.ptr.p, (const int32_t*)c->S.col.p, (const int4*)c->S.chunkInfo.p,
                               (int)c->ownLo, (int)c->ownHi, flags.p);
            HIP_CHECK(hipMemcpyAsync(h.data(), flags.p, (size_t)nS * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < nS; ++i) lists[h[(size_t)i] ? 1 : 0].push_back(i);
            hipLaunchKernelGGL(k_chunk_flags_St, dim3(gridFor(nT, BS)), dim3(BS), 0, c->stream, (const int32_t*)c->St.ptr.p, (const int4*)c->St.chunkInfo.p, nT,
                               (int)c->ownLo, (int)c->ownHi, flags.p);
            for (int a = 0; a < 3; ++a)
                if (forwarding() && c->nLowOwn[a] + c->nUpOwn[a] > 0)
                    hipLaunchKernelGGL(k_chunk_flags_relay, dim3(gridFor(c->nLowOwn[a] + c->nUpOwn[a], BS)), dim3(BS), 0, c->stream, (const int32_t*)c->listLowOwn[a].p, c->nLowOwn[a],
                                       (const int32_t*)c->listUpOwn[a].p, c->nUpOwn[a], (int)c->ownHi, (const int4*)c->St.chunkInfo.p, nT, flags.p);
            HIP_CHECK(hipMemcpyAsync(h.data(), flags.p, (size_t)nT * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < nT; ++i) { if (h[(size_t)i] == 2) lists[2].push_back(i); else if (h[(size_t)i] == 1) lists[3].push_back(i); if (h[(size_t)i] != 0) lists[4].push_back(i); }
            for (int q = 0; q < 5; ++q) {
                c->nDistList[q] = (int)lists[q].size();
                c->distList[q].alloc(lists[q].size());
                if (!lists[q].empty()) HIP_CHECK(hipMemcpyAsync(c->distList[q].p, lists[q].data(), lists[q].size() * 4, hipMemcpyHostToDevice, c->stream));
            }
            HIP_CHECK(hipStreamSynchronize(c->stream));        // (the host vectors go out of scope)
            c->distListsOk = true;
        }
    }
    // One exchange round or three forwarding rounds?  The lists were built without forwarded copies (ps_context::constructMatrixBlocks).  Every
    // rank checks on its own S — all of whose rows are its own — that each column outside its owned DOF range is a sample those lists deliver;
    // if ANY rank finds one that is not (agreed through the scalar all-reduce), every rank rebuilds its lists with the forwarded copies and the
    // exchanges run axis after axis as in r03 / r04.  PS_DIST_FORWARD=1 (lab build) forces the forwarding rounds.
    void decideExchangeMode() {
        static const bool force = PS_ENV("PS_DIST_FORWARD") && atoi(PS_ENV("PS_DIST_FORWARD")) != 0;
        bool need = force;
        for (ps_context* c : R) {
            if (need || c->nSystem == 0 || c->S.nnz == 0) continue;
            DevBuf<unsigned char>& mark = c->scrMark;
            mark.alloc((size_t)c->nSystem);
            HIP_CHECK(hipMemsetAsync(mark.p, 0, (size_t)c->nSystem, c->stream));
            HIP_CHECK(hipMemsetAsync(c->counters.p + 41, 0, sizeof(int32_t), c->stream));
            for (int a = 0; a < ps_context::NLINK; ++a) {
                if (c->nLowHalo[a] > 0) hipLaunchKernelGGL(k_mark_list, dim3(gridFor(c->nLowHalo[a], BS)), dim3(BS), 0, c->stream, (const int32_t*)c->listLowHalo[a].p, c->nLowHalo[a], mark.p);
                if (c->nUpHalo[a] > 0) hipLaunchKernelGGL(k_mark_list, dim3(gridFor(c->nUpHalo[a], BS)), dim3(BS), 0, c->stream, (const int32_t*)c->listUpHalo[a].p, c->nUpHalo[a], mark.p);
            }
            hipLaunchKernelGGL(k_count_unmarked_halo, dim3(gridFor(c->S.nnz, BS)), dim3(BS), 0, c->stream, (const int32_t*)c->S.col.p, (int64_t)c->S.nnz, (int)c->ownLo, (int)c->ownHi,
                               (const unsigned char*)mark.p, c->counters.p + 41);
            if (c->readCounter(41) != 0) need = true;
        }
        const bool fwd = sumFlag(need ? 1. : 0.) > 0.;
        for (ps_context* c : R)
            if (fwd != c->haloForward) { c->haloForward = fwd; c->buildHaloLists(); }
        if (PS_ENV_VERBOSE()) std::fprintf(stderr, "[polystokes] exchanges of the solve: %s\n", fwd ? "three forwarding rounds (a row reaches a diagonal neighbour's sample)" : "one round");
    }
    void finishSetup() {
        for (ps_context* c : R) c->redbuf.alloc(8);
        ensureStreams();
        decideExchangeMode();
        checkLists();
        buildLists();
        buildDirectExchange();
        exchangeAddY(&ps_context::b);
        const bool jac = R[0]->P.preconditioner == PS_PRE_DIAGONAL, cheb = R[0]->P.preconditioner == PS_PRE_CHEBYSHEV;
        if (jac || cheb) {
            exchangeAddY(&ps_context::dinv);
            for (ps_context* c : R) {
                const int64_t n = c->ownHi - c->ownLo;
                if (c->dinvF.p && c->nSystem > 0)       // (a rank without any DOF has no diagonal: constructPreconditioner returns before allocating it)
                    HIP_CHECK(hipMemsetAsync(c->dinvF.p, 0, (size_t)c->nSystem * sizeof(diag_t), c->stream));   // halo rows: never read as a diagonal
                if (n > 0) {
                    hipLaunchKernelGGL(k_invert_diag, dim3(dotBlocks(n)), dim3(BS), 0, c->stream, c->dinv.p + c->ownLo, n);
                    hipLaunchKernelGGL(k_to_diag, dim3(dotBlocks(n)), dim3(BS), 0, c->stream, c->dinv.p + c->ownLo, c->dinvF.p + c->ownLo, n);
                }
            }
        }
        if (cheb) estimateLambdaMaxDist();
    }

    int solve() {
        ps_context* c0 = R[0];
        const int maxit = c0->P.maxSolverIterations;
        const double tol = c0->P.tolerance;
        const bool jac = c0->P.preconditioner == PS_PRE_DIAGONAL, cheb = c0->P.preconditioner == PS_PRE_CHEBYSHEV;
        if (c0->P.solverType != PS_PCG_MATRIX_VECTOR_PRODUCTS) { c0->err = "Unsupported Solver."; return PS_UNSUPPORTED_SOLVER; }
        struct Loc { int64_t n, lo; int vb, stBlocks; const diag_t* dv; CGScalars* sc; Launch L; };
        std::vector<Loc> loc(R.size());
        for (size_t q = 0; q < R.size(); ++q) {
            ps_context* c = R[q];
            Loc& l = loc[q];
            l.lo = c->ownLo; l.n = c->ownHi - c->ownLo;
            l.vb = dotBlocks(std::max<int64_t>(l.n, 1));
            l.dv = jac ? c->dinvF.p + l.lo : nullptr;
            l.sc = c->scal.p;
            l.L = mk(c, &l.sc->done);
            l.stBlocks = l.L.stBlocks();
            c->usedBiCGStab = 0;
        }
        // r = b, x = 0, p = z on the owned range; rsold = sum over ranks of r.z
        for (size_t q = 0; q < R.size(); ++q) {
            ps_context* c = R[q];
            Loc& l = loc[q];
            HIP_CHECK(hipMemsetAsync(c->pvec.p, 0, (size_t)std::max<int64_t>(c->nSystem, 1) * 8, c->stream));
            HIP_CHECK(hipMemsetAsync(c->dotPartials3.p, 0, VGRID * sizeof(double), c->stream));
            // halo rows: A p there is this rank's share of a neighbour's row (or nothing: rows no launch of the overlapped step
            // touches are still packed by contributionsBack), r is never this rank's — neither may hold what an earlier step left
            if (c->nSystem > c->ownHi) {
                HIP_CHECK(hipMemsetAsync(c->Ap.p + c->ownHi, 0, (size_t)(c->nSystem - c->ownHi) * 8, c->stream));
                HIP_CHECK(hipMemsetAsync(c->r.p + c->ownHi, 0, (size_t)(c->nSystem - c->ownHi) * 8, c->stream));
            }
            if (c->ownLo > 0) {
                HIP_CHECK(hipMemsetAsync(c->Ap.p, 0, (size_t)c->ownLo * 8, c->stream));
                HIP_CHECK(hipMemsetAsync(c->r.p, 0, (size_t)c->ownLo * 8, c->stream));
            }
            hipLaunchKernelGGL(k_cg_init_f, dim3(l.vb), dim3(BS), 0, c->stream, c->b.p + l.lo, l.dv, c->x.p + l.lo, c->r.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials.p);
            hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)nullptr, c->dotPartials.p, l.vb, 0, 1, c->redbuf.p);
        }
        if (cheb) {   // z = M^-1 r, p = z, rsold = r.z
            for (ps_context* c : R) {
                const size_t nl = (size_t)std::max<int64_t>(c->nSystem, 1);
                c->tmp1.alloc(nl); c->tmp2.alloc(nl); c->tmp5.alloc(nl);
                c->chebPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor((int64_t)nl, BS)) + 16);
                HIP_CHECK(hipMemsetAsync(c->scal.p, 0, sizeof(CGScalars), c->stream));   // `done` must read 0 inside the polynomial's kernels
                HIP_CHECK(hipMemsetAsync(c->tmp1.p, 0, nl * 8, c->stream));
                HIP_CHECK(hipMemsetAsync(c->tmp2.p, 0, nl * 8, c->stream));
            }
            const Vec z0 = chebyshevDist(&ps_context::r);
            for (size_t q = 0; q < R.size(); ++q) {
                ps_context* c = R[q];
                Loc& l = loc[q];
                if (l.n > 0) HIP_CHECK(hipMemcpyAsync(c->pvec.p + l.lo, (c->*z0).p + l.lo, (size_t)l.n * 8, hipMemcpyDeviceToDevice, c->stream));
                hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)nullptr, (const double*)c->chebPartials.p, l.vb, 0, 1, c->redbuf.p);
            }
        }
        allreduce(1);
        for (size_t q = 0; q < R.size(); ++q)
            hipLaunchKernelGGL(k_dscal0, dim3(1), dim3(1), 0, R[q]->stream, loc[q].sc, R[q]->redbuf.p, tol, maxit, R[q]->ntLevel() >= 2 ? 1 : 0);
        // Four-kernel step across slabs (ps_solve.hip: solve, FusedR): every rank's share of p.Ap in its factored form is known after
        // the tile kernel — one all-reduce, then the St kernel updates r on the owned DOFs and hands the halo rows' (A p) to the
        // neighbours, who subtract alpha times it (k_dist_fixup).  Same rule as the single-domain solve: coded streams on every
        // rank and >= FUSED_STEP_MIN_ROWS owned rows on the largest (PS_FUSED_R = 0 / 1 forces); decided from values every rank
        // knows or agrees on, so all ranks take the same branch.
        static const int fusedEnv = PS_ENV("PS_FUSED_R") ? atoi(PS_ENV("PS_FUSED_R")) : -1;
        bool fused = !cheb && fusedEnv != 0;
        for (size_t q = 0; q < R.size(); ++q) fused = fused && loc[q].L.fusedOk() && loc[q].n > 0;
        {
            double mine[2] = {fused ? 0. : 1., 0.};
            for (size_t q = 0; q < R.size(); ++q) mine[1] = std::max(mine[1], (double)loc[q].n);
            if (useRccl || useTcp) {   // a rank without the coded stream vetoes; the size rule looks at the SUM of the owned rows (identical everywhere)
                for (ps_context* c : R) HIP_CHECK(hipMemcpyAsync(c->redbuf.p, mine, 16, hipMemcpyHostToDevice, c->stream));
                allreduce(2);
                HIP_CHECK(hipMemcpyAsync(mine, R[0]->redbuf.p, 16, hipMemcpyDeviceToHost, R[0]->stream));
                syncAll();
                mine[1] /= std::max(1, R[0]->slab.world);   // mean owned rows per rank
            }
            fused = fused && mine[0] == 0. && (fusedEnv > 0 || mine[1] >= (double)FUSED_STEP_MIN_ROWS);
        }
        struct FBuf { double *fS, *fT, *fU, *fR, *fX; int sBlocks, stBF, gFix, sI, tB; };   // sI / tB: workgroups of the first of the two S / St launches
        std::vector<FBuf> fb(R.size());
        // the exchanges overlap with the rows that do not need them when every rank has its chunk lists (row-per-lane kernels)
        // (the ranks of an in-process group share ONE stream: nothing runs beside anything, and splitting S and St into the chunks next to a cut and
        // the rest only doubles their launches — one launch each there; PS_DIST_OVERLAP=1 forces the split for the tests that walk that path on one GPU)
        static const bool forceSplit = PS_ENV("PS_DIST_OVERLAP") && atoi(PS_ENV("PS_DIST_OVERLAP")) == 1;
        bool overlap = fused && (useRccl || useTcp || forceSplit);
        for (ps_context* c : R) overlap = overlap && c->distListsOk;
        if (useRccl || useTcp) {   // all ranks take the same branch (the kernels differ, not the messages — but keep the ranks alike)
            double mine = overlap ? 0. : 1.;
            for (ps_context* c : R) HIP_CHECK(hipMemcpyAsync(c->redbuf.p, &mine, 8, hipMemcpyHostToDevice, c->stream));
            allreduce(1);
            HIP_CHECK(hipMemcpyAsync(&mine, R[0]->redbuf.p, 8, hipMemcpyDeviceToHost, R[0]->stream));
            syncAll();
            overlap = overlap && mine == 0.;
        }
        const bool timed = c0->commStream && c0->commStream != c0->stream;
        for (ps_context* c : R) {
            for (int q = 0; q < 8; ++q) c->distStats[q] = 0.;
            c->distStats[0] = 8. * (double)c->exchangeEntries();   // bytes this rank sends per iteration (x layers + A p contributions)
            c->distStats[1] = (double)(c->ownHi - c->ownLo);                                        // owned DOFs
            c->distStats[2] = overlap ? 1. : 0.;
            c->distStats[7] = (double)c->haloLabelChanges;
        }
        if (fused) {
            for (size_t q = 0; q < R.size(); ++q) {
                ps_context* c = R[q];
                Loc& l = loc[q];
                FBuf& f = fb[q];
                f.sBlocks = l.L.sBlocks(); f.stBF = (c->distListsOk && !overlap) ? l.L.stBlocksFor(c->nDistList[4], 3) : l.L.stBlocks(3); f.sI = 0; f.tB = 0;
                if (overlap) {
                    f.sI = l.L.sBlocksFor(c->nDistList[0]); f.sBlocks = f.sI + l.L.sBlocksFor(c->nDistList[1]);
                    f.tB = l.L.stBlocksFor(c->nDistList[2], 3); f.stBF = f.tB + l.L.stBlocksFor(c->nDistList[3], 3);
                }
                int64_t mostOwn = 1;
                for (int a = 0; a < ps_context::NLINK; ++a) mostOwn = std::max(mostOwn, c->nLowOwn[a] + c->nUpOwn[a]);
                if (c->nFix > 0) mostOwn = std::max<int64_t>(mostOwn, c->nFix);             // the merged fix-up: one thread per receiving DOF, up to 1024 workgroups
                f.gFix = (int)std::min<int64_t>(c->nFix > 0 ? 1024 : 256, (mostOwn + BS - 1) / BS);   // workgroups of one axis's k_dist_fixup; its partials: [axis][2][gFix]
                c->fusedPart.alloc((size_t)f.sBlocks + (size_t)c->regionCount + VGRID + 2 * (size_t)f.stBF + 2 * ps_context::NLINK * (size_t)f.gFix + 16);
                f.fS = c->fusedPart.p; f.fT = f.fS + f.sBlocks; f.fU = f.fT + c->regionCount; f.fR = f.fU + VGRID; f.fX = f.fR + 2 * f.stBF;
                HIP_CHECK(hipMemsetAsync(f.fX, 0, 2 * ps_context::NLINK * (size_t)f.gFix * sizeof(double), c->stream));
                l.L.sPart = f.fS; l.L.wvPart = f.fT;
                c->fusedStepHost = 1;
                const uint8_t* ucode = c->uCoded ? c->uCode.p + l.lo : nullptr;
                hipLaunchKernelGGL(k_uinv_pp, dim3(l.vb), dim3(BS), 0, c->stream, (const double*)c->pvec.p + l.lo, ucode, (const double*)c->uDict.p,
                                   (const double*)c->uInv.p + l.lo, l.n, f.fU);
            }
        } else for (ps_context* c : R) c->fusedStepHost = 0;
        CGScalars h{};
        const int batch = 25;
        int it = 0;
        bool finished = false, interrupted = false;
        for (ps_context* c : R) c->interrupted = false;
        while (it < maxit && !finished) {
            const int upto = std::min(maxit, it + batch);
            for (; it < upto; ++it) {
                if (fused && overlap) {
                    // (1) p on the cut layers -> the neighbours [comm stream]; meanwhile the S chunks that gather no halo value
                    const bool sample = it + 1 == upto;          // time the transports of the batch's last iteration (the host synchronises there anyway)
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        order(c, 0, true);
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        l.L.sList = c->distList[0].p; l.L.nSList = c->nDistList[0]; l.L.sPart = f.fS;
                        l.L.spmvS(0, c->pvec.p, c->ts.p);
                    }
                    if (sample && timed) HIP_CHECK(hipEventRecord(c0->distEv[2], cs(c0, true)));
                    valuesOut(&ps_context::pvec, true);          // pack, transport, unpack — axis after axis — on the comm stream
                    if (sample && timed) HIP_CHECK(hipEventRecord(c0->distEv[3], cs(c0, true)));
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        order(c, 1, false);
                        // (2) the S chunks next to a cut, the tiles, this rank's share of p.Ap
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        l.L.sList = c->distList[1].p; l.L.nSList = c->nDistList[1]; l.L.sPart = f.fS + f.sI;
                        l.L.spmvS(0, c->pvec.p, c->ts.p);
                        l.L.tiles(0, c->ts.p);
                        hipLaunchKernelGGL(k_fused_local_sum, dim3(1), dim3(1024), 0, c->stream, (const CGScalars*)l.sc, (const double*)f.fS, f.sBlocks, (const double*)f.fT,
                                           (int)c->regionCount, (const double*)f.fU, l.vb, (const double*)c->dotPartials3.p, l.vb, c->redbuf.p);
                    }
                    allreduce(2);
                    // (3) St on the chunks that hold halo rows: their share of the neighbours' A p -> [comm stream]; meanwhile St on the rest
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        FusedR fr{l.sc, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, it, c->r.p, jac ? c->dinvF.p : (const diag_t*)nullptr, f.fR, nullptr, 0., nullptr,
                                  (const double*)c->redbuf.p, (int)c->ownLo, (int)c->ownHi, c->Ap.p, f.stBF};
                        l.L.stList = c->distList[2].p; l.L.nStList = c->nDistList[2];
                        l.L.spmvSt(3, c->ts.p, c->pvec.p, nullptr, nullptr, nullptr, nullptr, &fr);
                        order(c, 4, true);
                        fr.rPart = f.fR + f.tB;
                        l.L.stList = c->distList[3].p; l.L.nStList = c->nDistList[3]; l.L.stOwnedOnly = true;
                        l.L.spmvSt(3, c->ts.p, c->pvec.p, nullptr, nullptr, nullptr, nullptr, &fr);
                        l.L.stOwnedOnly = false;
                    }
                    contributionsBack(&ps_context::Ap, true, false);   // [comm stream] the owners correct r from the receive buffers below
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        order(c, 5, false);
                        fixup(c, l.sc, jac, f.fX, f.gFix);
                        hipLaunchKernelGGL(k_sum_rr, dim3(1), dim3(1024), 0, c->stream, (const CGScalars*)l.sc, (const double*)f.fR, f.stBF, (const double*)f.fX, f.gFix, c->nFix > 0 ? 1 : ps_context::NLINK, c->redbuf.p);   // (the merged fix-up writes ONE set of partials)
                    }
                    allreduce(2);
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        const uint8_t* ucode = c->uCoded ? c->uCode.p + l.lo : nullptr;
                        hipLaunchKernelGGL(k_cg_update_xp_u, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, (const double*)nullptr, 0, jac ? 1 : 0, it,
                                           (const double*)c->r.p + l.lo, l.dv, c->x.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials3.p, ucode, (const double*)c->uDict.p,
                                           (const double*)c->uInv.p + l.lo, f.fU);
                    }
                    continue;
                }
                exchangeX(&ps_context::pvec);
                if (fused) {
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        l.L.spmvS(0, c->pvec.p, c->ts.p);
                        l.L.tiles(0, c->ts.p);
                        hipLaunchKernelGGL(k_fused_local_sum, dim3(1), dim3(1024), 0, c->stream, (const CGScalars*)l.sc, (const double*)f.fS, f.sBlocks, (const double*)f.fT,
                                           (int)c->regionCount, (const double*)f.fU, l.vb, (const double*)c->dotPartials3.p, l.vb, c->redbuf.p);
                    }
                    allreduce(2);
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        const FusedR fr{l.sc, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, it, c->r.p, jac ? c->dinvF.p : (const diag_t*)nullptr, f.fR, nullptr, 0., nullptr,
                                        (const double*)c->redbuf.p, (int)c->ownLo, (int)c->ownHi, c->Ap.p, f.stBF};
                        // (the chunks that hold work: halo rows without entries — most of a halo block — are in no list; r and A p of those rows stay as the solve's start left them: zero)
                        if (c->distListsOk) { l.L.stList = c->distList[4].p; l.L.nStList = c->nDistList[4]; }
                        l.L.spmvSt(3, c->ts.p, c->pvec.p, nullptr, nullptr, nullptr, nullptr, &fr);
                        l.L.stList = nullptr; l.L.nStList = 0;
                    }
                    // the halo rows' share of A p goes to its owners (the packing and transport of exchangeAddY; the owners correct r instead of adding into A p)
                    contributionsBack(&ps_context::Ap, false, false);
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        fixup(c, l.sc, jac, f.fX, f.gFix);
                        hipLaunchKernelGGL(k_sum_rr, dim3(1), dim3(1024), 0, c->stream, (const CGScalars*)l.sc, (const double*)f.fR, f.stBF, (const double*)f.fX, f.gFix, c->nFix > 0 ? 1 : ps_context::NLINK, c->redbuf.p);   // (the merged fix-up writes ONE set of partials)
                    }
                    allreduce(2);
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        FBuf& f = fb[q];
                        const uint8_t* ucode = c->uCoded ? c->uCode.p + l.lo : nullptr;
                        hipLaunchKernelGGL(k_cg_update_xp_u, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, (const double*)nullptr, 0, jac ? 1 : 0, it,
                                           (const double*)c->r.p + l.lo, l.dv, c->x.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials3.p, ucode, (const double*)c->uDict.p,
                                           (const double*)c->uInv.p + l.lo, f.fU);
                    }
                    continue;
                }
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    l.L.spmvS(0, c->pvec.p, c->ts.p);
                    l.L.tiles(0, c->ts.p);
                    l.L.spmvSt(0, c->ts.p, c->pvec.p, nullptr, c->Ap.p, c->dotPartials.p);
                    if (l.stBlocks <= 8192) {
                        hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartials.p, l.stBlocks, 0, 1, c->redbuf.p);
                    } else {
                        hipLaunchKernelGGL(k_reduce_partials, dim3(RED_BLOCKS), dim3(BS), 0, c->stream, l.sc, c->dotPartials.p, l.stBlocks, c->dotPartials2.p);
                        hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartials2.p, RED_BLOCKS, 0, 1, c->redbuf.p);
                    }
                    // ||x||^2 of the x updated last iteration rides along (stop test of the previous iteration, see k_cg_update_r)
                    hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartials3.p, l.vb, 0, 1, c->redbuf.p + 1);
                }
                exchangeAddY(&ps_context::Ap);
                allreduce(2);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    hipLaunchKernelGGL(k_cg_update_r, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, (const double*)nullptr, 0,
                                       (const double*)nullptr, 0, it, c->Ap.p + l.lo, l.dv, c->r.p + l.lo, l.n, c->dotPartialsR.p);
                    hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartialsR.p, l.vb, l.vb, 2, c->redbuf.p);
                }
                if (cheb) {   // z = M^-1 r (k-1 distributed applies), then {r.r, r.z} and x, p with the vector z
                    const Vec zf = chebyshevDist(&ps_context::r);
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, (const double*)c->chebPartials.p, l.vb, 0, 1, c->redbuf.p + 1);
                    }
                    allreduce(2);
                    for (size_t q = 0; q < R.size(); ++q) {
                        ps_context* c = R[q];
                        Loc& l = loc[q];
                        hipLaunchKernelGGL(k_cg_update_xp_z<double>, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, 1, (const double*)c->redbuf.p + 1, 1, it,
                                           (const double*)(c->*zf).p + l.lo, c->x.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials3.p);
                    }
                    continue;
                }
                allreduce(2);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    hipLaunchKernelGGL(k_cg_update_xp, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, (const double*)nullptr, 0,
                                       jac ? 1 : 0, it, c->r.p + l.lo, l.dv, c->x.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials3.p);
                }
            }
            // the stop test of the batch's last iteration; the ranks' interrupt requests ride along in the same all-reduce,
            // so that every rank leaves the loop at the same batch (a rank stopping alone would strand its neighbours in a receive)
            double wantStop = 0.;
            for (ps_context* c : R) if (c->interruptCb && c->interruptCb(c->interruptUser)) wantStop = 1.;
            for (size_t q = 0; q < R.size(); ++q) {
                hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, R[q]->stream, (const CGScalars*)loc[q].sc, R[q]->dotPartials3.p, loc[q].vb, 0, 1, R[q]->redbuf.p);
                HIP_CHECK(hipMemcpyAsync(R[q]->redbuf.p + 1, &wantStop, 8, hipMemcpyHostToDevice, R[q]->stream));
            }
            const auto ar0 = std::chrono::high_resolution_clock::now();
            if (useRccl || useTcp) syncAll();                    // (so that the clock below sees the collective alone)
            const auto ar1 = std::chrono::high_resolution_clock::now();
            allreduce(2);
            if (useRccl || useTcp) {
                syncAll();
                c0->distStats[5] += std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - ar1).count();
                c0->distStats[6] += 1.;
            }
            (void)ar0;
            for (size_t q = 0; q < R.size(); ++q)
                hipLaunchKernelGGL(k_cg_check, dim3(1), dim3(BS), 0, R[q]->stream, loc[q].sc, (const double*)R[q]->redbuf.p, (const double*)nullptr, 0, it - 1);
            double stopSum = 0.;
            HIP_CHECK(hipMemcpyAsync(&h, loc[0].sc, sizeof(h), hipMemcpyDeviceToHost, c0->stream));
            HIP_CHECK(hipMemcpyAsync(&stopSum, c0->redbuf.p + 1, 8, hipMemcpyDeviceToHost, c0->stream));
            syncAll();
            if (overlap && timed) {   // the transport of the batch's last x exchange, as the comm stream saw it
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, c0->distEv[2], c0->distEv[3]) == hipSuccess) { c0->distStats[3] += (double)ms; c0->distStats[4] += 1.; }
            }
            if (h.done) finished = true;
            else if (stopSum > 0. || (wantStop > 0. && !useRccl && !useTcp)) { interrupted = true; break; }
        }
        if (interrupted) {
            for (ps_context* c : R) { c->solveIterations = it; c->solveError = std::sqrt(h.rre); c->interrupted = true; }
            return PS_INCOMPLETE;
        }
        int iters = h.done ? h.iter : maxit;
        double err = std::sqrt(h.rre);
        if (iters == maxit) {
            // bicgstab_external_matrix_A (pcg.h:134-200), restarted from zero (Solver.cpp:784-799): the host-driven loop
            // of ps_context::solve() with distributed applies and dots.  Rare path.
            for (ps_context* c : R) {
                c->usedBiCGStab = 1;
                const size_t nl = (size_t)std::max<int64_t>(c->nSystem, 1);
                c->tmp1.alloc(nl); c->tmp2.alloc(nl); c->tmp3.alloc(nl); c->tmp4.alloc(nl); c->tmp5.alloc(nl);
            }
            using Vec = DevBuf<double> ps_context::*;
            const Vec X = &ps_context::x, Rv = &ps_context::r, Pv = &ps_context::pvec, B = &ps_context::b, H = &ps_context::Ap,
                      Rhat = &ps_context::tmp1, V = &ps_context::tmp2, S = &ps_context::tmp3, T = &ps_context::tmp4, E = &ps_context::tmp5;
            auto apply = [&](Vec in, Vec out) {
                exchangeX(in);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Launch L = mk(c, nullptr);
                    L.spmvS(0, (c->*in).p, c->ts.p);
                    L.tiles(0, c->ts.p);
                    L.spmvSt(0, c->ts.p, (c->*in).p, nullptr, (c->*out).p, c->dotPartials.p);
                }
                exchangeAddY(out);
            };
            auto dot = [&](Vec a, Vec bvec) {
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    hipLaunchKernelGGL(k_dot, dim3(l.vb), dim3(BS), 0, c->stream, (c->*a).p + l.lo, (c->*bvec).p + l.lo, l.n, c->dotPartials.p);
                    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(BS), 0, c->stream, c->dotPartials.p, l.vb, c->redbuf.p);
                }
                allreduce(1);
                double out = 0.;
                HIP_CHECK(hipMemcpyAsync(&out, c0->redbuf.p, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
                syncAll();
                return out;
            };
            auto lin = [&](Vec out, double ca, Vec a, double cb, Vec bvec, double cc, Vec c3) {   // out = ca a + cb b + cc c on the owned range
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    if (l.n <= 0) continue;
                    hipLaunchKernelGGL(k_lin, dim3(l.vb), dim3(BS), 0, c->stream, (c->*out).p + l.lo, ca, (const double*)(c->*a).p + l.lo, cb,
                                       bvec ? (const double*)(c->*bvec).p + l.lo : (const double*)nullptr, cc,
                                       c3 ? (const double*)(c->*c3).p + l.lo : (const double*)nullptr, l.n);
                }
            };
            auto zero = [&](Vec v) { for (ps_context* c : R) HIP_CHECK(hipMemsetAsync((c->*v).p, 0, (size_t)std::max<int64_t>(c->nSystem, 1) * 8, c->stream)); };
            zero(X);
            lin(Rv, 1., B, 0., nullptr, 0., nullptr);            // r = b - A*0
            lin(Rhat, 1., Rv, 0., nullptr, 0., nullptr);
            zero(Pv); zero(V);
            double rhoCurr = 1., rhoOld = 1., alpha = 1., beta = 0., omega = 1., rre = 0.;
            iters = maxit;
            for (int i = 0; i < maxit; ++i) {
                rhoOld = rhoCurr;
                rhoCurr = dot(Rhat, Rv);
                beta = (rhoCurr / rhoOld) * (alpha / omega);
                lin(Pv, 1., Rv, beta, Pv, -beta * omega, V);      // p = r + beta (p - omega v)
                apply(Pv, V);
                alpha = rhoCurr / dot(Rhat, V);
                lin(H, 1., X, alpha, Pv, 0., nullptr);             // h = x + alpha p
                lin(S, 1., Rv, -alpha, V, 0., nullptr);            // s = r - alpha v
                apply(S, T);
                omega = dot(T, S) / dot(T, T);
                lin(X, 1., H, omega, S, 0., nullptr);              // x = h + omega s
                const double xmag = std::sqrt(dot(X, X));
                apply(X, E);
                lin(E, 1., B, -1., E, 0., nullptr);                // err = b - A x
                const double rsnew = dot(E, E);
                rre = rsnew;
                if (std::sqrt(rsnew) / xmag < rre) rre = std::sqrt(rsnew) / xmag;
                if (rre < tol) { iters = i; break; }
                lin(Rv, 1., S, -omega, T, 0., nullptr);            // r = s - omega t
            }
            err = rre;
        }
        for (ps_context* c : R) { c->solveIterations = iters; c->solveError = err; }
        return iters == maxit ? PS_NOCONVERGE : PS_SUCCESS;
    }

    void recoverAndWriteBack(bool apply) {
        if (apply) exchangeX(&ps_context::x);
        for (ps_context* c : R) {
            c->buildValidFaces();
            if (apply) { c->recoverVelocityFromPressureStress(); c->applySolutionToVelocity(); }
            else for (int a = 0; a < 3; ++a)
                HIP_CHECK(hipMemcpyAsync(c->velOut[a].p, c->vel[a].p, (size_t)c->g.count(1 + a) * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            for (int a = 0; a < 3; ++a) {
                c->ownedFace[a].alloc((size_t)c->g.count(1 + a));
                hipLaunchKernelGGL(k_owned_faces, dim3(gridFor(c->g.count(1 + a), BS)), dim3(BS), 0, c->stream, c->g, a, c->own(), c->faceRow[a].p,
                                   c->reducedIdx[1 + a].p, (c->slabEnabled && c->regionCount > 0) ? c->regionOwned.p : (const int32_t*)nullptr,
                                   c->ownedFace[a].p);
            }
        }
        syncAll();
    }
};

int distStep(Dist& D, ps_stats* stats) {
    const auto w0 = std::chrono::high_resolution_clock::now();
    for (ps_context* c : D.R) c->redbuf.alloc(8);
    // a rank whose local setup throws must tell the others before they enter the first exchange (they would wait for ever)
    std::string failure;
    D.ensureStreams();
    for (int phase = 0; phase < 3; ++phase) {
        for (ps_context* c : D.R) {
            if (!failure.empty()) break;
            try { c->setupPhase(phase); }
            catch (const ps::Error& e) { failure = e.msg; }
            catch (const std::exception& e) { failure = e.what(); }
        }
        if (D.sumFlag(failure.empty() ? 0. : 1.) > 0.)
            throw Error(failure.empty() ? std::string("another rank failed during setup") : failure);
        if (phase < 2 && (phase == 0 || D.R[0]->P.doReducedRegions)) D.exchangeLabels(phase);
    }
    D.finishSetup();
    D.syncAll();
    const auto w1 = std::chrono::high_resolution_clock::now();
    int result = PS_INCOMPLETE;
    ps_context* c0 = D.R[0];
    if (c0->P.doSolve) result = D.solve();
    D.syncAll();
    const auto w2 = std::chrono::high_resolution_clock::now();
    // (as ps_context::solveStage: HDK_PolyStokes.C:566-583 — also with doSolve off, then from the zero solution vector)
    const bool apply = result != PS_UNSUPPORTED_SOLVER && !(c0->P.doSolve && c0->interrupted) && (result == PS_SUCCESS || c0->P.keepNonConvergedResults);
    D.recoverAndWriteBack(apply);
    for (ps_context* c : D.R) {
        c->lastStats.solveData[0] = c->solveError;
        c->lastStats.solveData[1] = c->solveIterations;
        c->lastStats.solveData[3] = std::chrono::duration<double, std::milli>(w2 - w1).count();
        c->lastStats.solveData[5] = std::chrono::duration<double, std::milli>(w1 - w0).count();
        c->lastStats.stage_ms[PS_STAGE_SOLVE] = c->lastStats.solveData[3];
        c->lastStats.result = result;
        c->lastStats.usedBiCGStab = c->usedBiCGStab;
        c->isSolved = true;
        c->registerArrays();
    }
    if (stats) *stats = c0->lastStats;
    return result;
}

}  // namespace

struct ps_group {
    std::vector<ps_context*> ranks;
    hipStream_t stream = nullptr;
};

int ps_dist_step_single(ps_context* c, ps_stats* stats) {   // one process per rank: RCCL (one GPU each) or the TCP transport
    Dist D;
    D.R.push_back(c);
    D.useRccl = c->rcclComm != nullptr;
    D.useTcp = !D.useRccl && c->hostComm != nullptr;
    return distStep(D, stats);
}
void ps_dist_release(ps_context* c) {
    if (c->rcclComm) {
        try { (void)rccl().CommDestroy(c->rcclComm); } catch (...) {}
        c->rcclComm = nullptr;
        ps::MemState& M = ps::memState(); std::lock_guard<std::mutex> lk(M.m); --M.asyncRanks;
    }
    if (c->hostComm) { delete (HostComm*)c->hostComm; c->hostComm = nullptr; }
    if (c->commStream && c->commStream != c->stream) (void)hipStreamDestroy(c->commStream);
    c->commStream = nullptr;
    for (int e = 0; e < 8; ++e) if (c->distEv[e]) { (void)hipEventDestroy(c->distEv[e]); c->distEv[e] = nullptr; }
}
#define PS_CATCH_ALL(ctx)                                                                    \
    catch (const ps::Error& e) { if (ctx) { (ctx)->err = e.msg; (ctx)->drainDeferred(false); } return PS_FAILED; }            \
    catch (const std::exception& e) { if (ctx) { (ctx)->err = e.what(); (ctx)->drainDeferred(false); } return PS_FAILED; }

extern "C" {

int32_t ps_set_brick(ps_context* c, const ps_brick* bk) {
    if (!c || !bk) return PS_FAILED;
    try {
        if (!c->uploaded) throw Error("ps_upload_fields first");
        const int n[3] = {c->g.nx, c->g.ny, c->g.nz};
        const int L = 16;
        if (bk->world < 1 || bk->dims[0] < 1 || bk->dims[1] < 1 || bk->dims[2] < 1 || bk->dims[0] * bk->dims[1] * bk->dims[2] != bk->world || bk->rank < 0 || bk->rank >= bk->world)
            throw Error("bad decomposition: world = dims[0] * dims[1] * dims[2] ranks, 0 <= rank < world");
        const int coord[3] = {bk->rank % bk->dims[0], (bk->rank / bk->dims[0]) % bk->dims[1], bk->rank / (bk->dims[0] * bk->dims[1])};
        if (c->P.doReducedRegions && !c->P.doTile && bk->world > 1) throw Error("the decomposition needs doTile (tile-local regions)");
        // every label inside the owned box must equal the global one: the classification reaches L + S cells, the tile
        // relabelling tilePadding more, the trilinear samplers one cell each side — all of it has to lie inside the halo block
        if (bk->world > 1 && c->P.doReducedRegions &&
            c->P.activeLiquidBoundaryLayerSize + c->P.activeSolidBoundaryLayerSize + c->P.tilePadding + 2 > 16)
            throw Error("activeLiquidBoundaryLayerSize + activeSolidBoundaryLayerSize + tilePadding + 2 exceeds the 16-cell halo of the decomposition");
        for (int a = 0; a < 3; ++a) {
            const int lo = bk->lo[a], hi = bk->hi[a];
            if ((bk->hasLower[a] != 0) != (coord[a] > 0) || (bk->hasUpper[a] != 0) != (coord[a] + 1 < bk->dims[a])) throw Error("the brick's neighbours do not match its position in dims");
            if (lo % L || hi % L) { if (!(hi == n[a] && !bk->hasUpper[a] && lo % L == 0)) throw Error("cuts must be multiples of 16"); }
            if (c->P.doReducedRegions && c->P.doTile && (lo % c->P.tileSize || (bk->hasUpper[a] && hi % c->P.tileSize))) throw Error("cuts must be multiples of the tile size");
            if (lo < 0 || hi > n[a] || lo >= hi) throw Error("bad owned range");
            if ((bk->hasLower[a] && lo < 16) || (bk->hasUpper[a] && n[a] - hi < 16)) throw Error("a halo of at least 16 cells is required next to a cut");
            // one halo block per cut, the same on both sides of it (the ranks exchange the labels of whole halo blocks: Dist::exchangeLabels)
            if (bk->world > 1) {
                int al = L;
                if (c->P.doReducedRegions && c->P.doTile) { int x = L, y = c->P.tileSize; while (y) { const int t = x % y; x = y; y = t; } al = L / x * c->P.tileSize; }
                if ((bk->hasLower[a] && lo != al) || (bk->hasUpper[a] && n[a] - hi != al)) throw Error("the halo next to a cut must be one block of lcm(16, tileSize) cells");
                if ((bk->hasLower[a] || bk->hasUpper[a]) && hi - lo < al) throw Error("a brick must be at least one halo block thick");
            }
            if (!bk->hasLower[a] && lo != 0) throw Error("without a lower neighbour the owned range must start at 0");
            if (!bk->hasUpper[a] && hi != n[a]) throw Error("without an upper neighbour the owned range must end at the last cell");
        }
        c->brick = *bk;
        c->slab.rank = bk->rank; c->slab.world = bk->world;
        c->slab.zLoOwned = bk->lo[2]; c->slab.zHiOwned = bk->hi[2]; c->slab.hasLower = bk->hasLower[2]; c->slab.hasUpper = bk->hasUpper[2]; c->slab.zGlobalOwned = bk->globalLo[2];
        c->slabEnabled = bk->world > 1;
        for (int a = 0; a < 3; ++a) c->gOff[a] = c->slabEnabled ? bk->globalLo[a] - bk->lo[a] : 0;
        c->blockMapOwned = -1;
        c->isSetup = false;
        return PS_SUCCESS;
    } PS_CATCH_ALL(c)
}
int32_t ps_set_slab(ps_context* c, const ps_slab* slab) {   // z-slabs: the decomposition 1 x 1 x world
    if (!c || !slab) return PS_FAILED;
    ps_brick b{};
    b.rank = slab->rank; b.world = slab->world;
    b.dims[0] = b.dims[1] = 1; b.dims[2] = slab->world;
    b.lo[0] = b.lo[1] = 0; b.hi[0] = c->g.nx; b.hi[1] = c->g.ny;
    b.lo[2] = slab->zLoOwned; b.hi[2] = slab->zHiOwned;
    b.hasLower[2] = slab->hasLower; b.hasUpper[2] = slab->hasUpper;
    b.globalLo[2] = slab->zGlobalOwned;
    return ps_set_brick(c, &b);
}

// What the last distributed solve of this rank did: [0] bytes it sends per iteration over its cuts, [1] owned DOFs, [2] 1 if the
// exchanges ran under the rows that do not need them, [3] / [4] summed ms / samples of one x-exchange transport (comm stream events,
// batch ends), [5] / [6] summed ms / samples of one scalar all-reduce incl. its synchronisation (host clock, batch ends), [7] halo cells whose label the owners' exchange changed (Dist::exchangeLabels)
int32_t ps_dist_stats(ps_context* c, double* out8) {
    if (!c || !out8) return PS_FAILED;
    for (int q = 0; q < 8; ++q) out8[q] = c->distStats[q];
    return PS_SUCCESS;
}
int32_t ps_comm_unique_id(void* id128) {
    if (!id128) return PS_FAILED;
    try { ncclCheck(rccl().GetUniqueId(id128), "ncclGetUniqueId"); return PS_SUCCESS; }
    catch (const ps::Error& e) { std::fprintf(stderr, "%s\n", e.msg.c_str()); return PS_FAILED; }
    catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return PS_FAILED; }
}
int32_t ps_comm_init_rccl(ps_context* c, const void* id128, int32_t rank, int32_t world) {
    if (!c || !id128) return PS_FAILED;
    try {
        HIP_CHECK(hipSetDevice(c->device));
        PsNcclUid id;
        std::memcpy(&id, id128, sizeof(id));
        void* comm = nullptr;
        ncclCheck(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
        ps_dist_release(c);
        c->rcclComm = comm;
        { ps::MemState& M = ps::memState(); std::lock_guard<std::mutex> lk(M.m); ++M.asyncRanks; }   // (ps_common.hpp: several of these in one process = threaded ranks)
        return PS_SUCCESS;
    } PS_CATCH_ALL(c)
}
// Host-staged transport between one process per rank (several ranks may share a GPU): rank r listens on base_port + r of
// `host` (dotted IPv4, e.g. "127.0.0.1").  Collective: every rank of the world calls it.
int32_t ps_comm_init_tcp(ps_context* c, int32_t rank, int32_t world, const char* host, int32_t base_port) {
    if (!c || !host) return PS_FAILED;
    try {
        if (world < 1 || world > 64 || rank < 0 || rank >= world || base_port < 1024 || base_port + world > 65535) throw Error("ps_comm_init_tcp: bad rank / world / port");
        ps_dist_release(c);
        HostComm* H = new HostComm();
        try { H->init(rank, world, host, base_port); } catch (...) { delete H; throw; }
        c->hostComm = H;
        return PS_SUCCESS;
    } PS_CATCH_ALL(c)
}

// exercises the RCCL entry points used by the distributed solve (all-reduce, grouped send/recv to self) on this
// rank's communicator; returns PS_SUCCESS when the values come back right.
int32_t ps_comm_selftest(ps_context* c) {
    if (!c) return PS_FAILED;
    ps::SinkScope sinkScope_(&c->deferred);
    try {
        if (!c->rcclComm) throw Error("no communicator");
        HIP_CHECK(hipSetDevice(c->device));
        c->redbuf.alloc(8); c->sendLo[2].alloc(8); c->recvLo[2].alloc(8);
        const double v[4] = {1.5, -2.0, 3.25, 4.0};
        HIP_CHECK(hipMemcpyAsync(c->redbuf.p, v, 32, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->sendLo[2].p, v, 32, hipMemcpyHostToDevice, c->stream));
        Rccl& L = rccl();
        ncclCheck(L.AllReduce(c->redbuf.p, c->redbuf.p, 3, NCCL_DOUBLE, NCCL_SUM, c->rcclComm, c->stream), "ncclAllReduce");
        ncclCheck(L.GroupStart(), "ncclGroupStart");
        ncclCheck(L.Send(c->sendLo[2].p, 4, NCCL_DOUBLE, c->slab.rank, c->rcclComm, c->stream), "ncclSend");
        ncclCheck(L.Recv(c->recvLo[2].p, 4, NCCL_DOUBLE, c->slab.rank, c->rcclComm, c->stream), "ncclRecv");
        ncclCheck(L.GroupEnd(), "ncclGroupEnd");
        double a[4], b[4];
        HIP_CHECK(hipMemcpyAsync(a, c->redbuf.p, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpyAsync(b, c->recvLo[2].p, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 4; ++i) if (b[i] != v[i]) throw Error("send/recv self-test mismatch");
        if (a[3] != v[3]) throw Error("all-reduce touched elements beyond count");
        // with a slab set: one ring step between REAL neighbours (send up, receive from below), the pattern of Dist::transport
        const int world = c->slab.world, rank = c->slab.rank;
        if (c->slabEnabled && world > 1) {
            if (a[0] != world * v[0] || a[1] != world * v[1] || a[2] != world * v[2]) throw Error("all-reduce self-test: wrong sum over the ranks");
            const double mine[4] = {(double)rank, 100. + rank, -1. - rank, 0.5 * rank};
            HIP_CHECK(hipMemcpyAsync(c->sendLo[2].p, mine, 32, hipMemcpyHostToDevice, c->stream));
            const int up = (rank + 1) % world, down = (rank + world - 1) % world;
            ncclCheck(L.GroupStart(), "ncclGroupStart");
            ncclCheck(L.Send(c->sendLo[2].p, 4, NCCL_DOUBLE, up, c->rcclComm, c->stream), "ncclSend");
            ncclCheck(L.Recv(c->recvLo[2].p, 4, NCCL_DOUBLE, down, c->rcclComm, c->stream), "ncclRecv");
            ncclCheck(L.GroupEnd(), "ncclGroupEnd");
            HIP_CHECK(hipMemcpyAsync(b, c->recvLo[2].p, 32, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (b[0] != (double)down || b[1] != 100. + down || b[2] != -1. - down || b[3] != 0.5 * down) throw Error("neighbour send/recv self-test mismatch");
        }
        return PS_SUCCESS;   // a[0..2] = world * v (checked by the caller, who knows the world size)
    } PS_CATCH_ALL(c)
}

ps_group* ps_group_create(int32_t device, int32_t world) {
    if (world < 1 || world > 16) return nullptr;
    ps_group* g = new ps_group();
    for (int q = 0; q < world; ++q) {
        ps_context* c = ps_context_create(device);
        if (!c) { for (ps_context* d : g->ranks) ps_context_destroy(d); delete g; return nullptr; }
        if (q == 0) g->stream = c->stream;
        else { (void)hipStreamDestroy(c->stream); c->stream = g->stream; c->ownsStream = false; }   // one shared stream: launches are ordered
        g->ranks.push_back(c);
    }
    return g;
}
void ps_group_destroy(ps_group* g) {
    if (!g) return;
    for (size_t q = g->ranks.size(); q-- > 0;) ps_context_destroy(g->ranks[q]);
    delete g;
}
ps_context* ps_group_rank(ps_group* g, int32_t rank) { return (g && rank >= 0 && rank < (int)g->ranks.size()) ? g->ranks[(size_t)rank] : nullptr; }
int32_t ps_group_step(ps_group* g, ps_stats* stats) {
    if (!g || g->ranks.empty()) return PS_FAILED;
    try {
        Dist D;
        D.R = g->ranks;
        D.useRccl = false;
        ps::SinkScope sinkScope_(&g->ranks[0]->deferred);   // (the ranks of a group share one stream and one host thread: one list)
        const int result = distStep(D, stats);
        for (ps_context* c : g->ranks) c->drainDeferred(true);
        return result;
    } PS_CATCH_ALL(g->ranks[0])
}

}  // extern "C"

