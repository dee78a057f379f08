// polystokes_amd — MI355X-native PolyStokes hot path.  Shared host/device definitions.
//
// Data layout in HBM: every grid quantity is a dense x-fastest array (SoA, one array per sample
// grid); the 7 sample grids are indexed 0 = cell centre, 1..3 = face X/Y/Z, 4..6 = edge YZ/XZ/XY
// (edge axis 0,1,2 as in exec/HDK_PolyStokesSolver.h:397-411).  Labels/indices are int32 on device
// (the reference's exint fields, Solver.h:327-335; values identical, width halved to save HBM traffic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/polystokes.h"

#define PS_RD PS_REDUCED_DOF   // 26 (quadratic regions) or 11 (-DPS_AFFINE_REGIONS)

// Environment switches.  The lab build (libpolystokes_hip.so: tests, A/B measurements) reads the PS_* variables named at their points of
// use; the RELEASE build (-DPS_RELEASE: libpolystokes_hip_release.so, what the Houdini shim links) reads NONE of them — PS_ENV(...) is a
// null pointer at preprocessing time, so not even the names reach the binary (tests/test_abi_cpu.py checks its strings) — except
// PS_VERBOSE, which only prints.  A drop-in DSO inside another program must not change its results with a stray variable.
#include <stdio.h>
#include <stdlib.h>
#ifdef PS_RELEASE
#define PS_ENV(name) (static_cast<const char*>(nullptr))
#else
#define PS_ENV(name) getenv(name)
#endif
#define PS_ENV_VERBOSE() getenv("PS_VERBOSE")
// the lab build names, once, every switch that changes RESULTS or loads code when it finds it set
#ifdef PS_RELEASE
#define PS_ENV_LOUD(name) (static_cast<const char*>(nullptr))
#else
#define PS_ENV_LOUD(name) ps_env_loud(name)
inline const char* ps_env_loud(const char* name) {
    const char* v = getenv(name);
    if (v) fprintf(stderr, "[polystokes] WARNING: test-only switch %s=%s is active in this process (lab build; the release build ignores it)\n", name, v);
    return v;
}
#endif

#define HIP_CHECK(x)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) throw ps::Error(std::string(#x) + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace ps {

struct Error {
    std::string msg;
    explicit Error(std::string m) : msg(std::move(m)) {}
};

struct Grid {
    int nx, ny, nz;
    int order;  // ps_index_order
    __host__ __device__ int3 dims(int s) const {
        int3 d = make_int3(nx, ny, nz);
        if (s >= 1 && s <= 3) { if (s == 1) d.x++; else if (s == 2) d.y++; else d.z++; }
        else if (s >= 4) { const int e = s - 4; if (e != 0) d.x++; if (e != 1) d.y++; if (e != 2) d.z++; }
        return d;
    }
    __host__ __device__ int64_t count(int s) const { const int3 d = dims(s); return (int64_t)d.x * d.y * d.z; }
};

__host__ __device__ inline int64_t lin3(const int3 d, int i, int j, int k) { return i + (int64_t)d.x * (j + (int64_t)d.y * k); }
__host__ __device__ inline bool oob3(const int3 d, int i, int j, int k) {
    return i < 0 || i >= d.x || j < 0 || j >= d.y || k < 0 || k >= d.z;
}
__host__ __device__ inline int comp(const int3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
__host__ __device__ inline void addc(int3& v, int a, int q) { if (a == 0) v.x += q; else if (a == 1) v.y += q; else v.z += q; }
__host__ __device__ inline int3 unlin3(const int3 d, int64_t c) {
    int3 r;
    r.x = (int)(c % d.x);
    const int64_t q = c / d.x;
    r.y = (int)(q % d.y);
    r.z = (int)(q / d.y);
    return r;
}

// Position t in the traversal order of serialAssignFieldIndices (Classifier.cpp:1738-1770) -> (i,j,k).
// PS_ORDER_VOXEL_TILES: UT_VoxelArray order — 16^3 voxel tiles, tile-linear (x fastest), x-fastest inside.
__host__ __device__ inline int3 orderToIjk(const int3 d, int order, int64_t t) {
    if (order == PS_ORDER_LINEAR) return unlin3(d, t);
    const int T = 16;
    const int ntz = (d.z + T - 1) / T, nty = (d.y + T - 1) / T, ntx = (d.x + T - 1) / T;
    const int64_t slab = (int64_t)d.x * d.y * T;
    int tz = (int)(t / slab);
    if (tz > ntz - 1) tz = ntz - 1;
    t -= slab * tz;
    const int hz = (tz * T + T <= d.z) ? T : d.z - tz * T;
    const int64_t rowv = (int64_t)d.x * T * hz;
    int ty = (int)(t / rowv);
    if (ty > nty - 1) ty = nty - 1;
    t -= rowv * ty;
    const int hy = (ty * T + T <= d.y) ? T : d.y - ty * T;
    const int64_t tilev = (int64_t)T * hy * hz;
    int tx = (int)(t / tilev);
    if (tx > ntx - 1) tx = ntx - 1;
    t -= tilev * tx;
    const int wx = (tx * T + T <= d.x) ? T : d.x - tx * T;
    int3 r;
    r.x = tx * T + (int)(t % wx);
    const int64_t q = t / wx;
    r.y = ty * T + (int)(q % hy);
    r.z = tz * T + (int)(q / hy);
    return r;
}
__host__ __device__ inline int64_t ijkToOrder(const int3 d, int order, int i, int j, int k) {
    if (order == PS_ORDER_LINEAR) return lin3(d, i, j, k);
    const int T = 16;
    const int tz = k / T, ty = j / T, tx = i / T;
    const int hz = (tz * T + T <= d.z) ? T : d.z - tz * T;
    const int hy = (ty * T + T <= d.y) ? T : d.y - ty * T;
    const int wx = (tx * T + T <= d.x) ? T : d.x - tx * T;
    return (int64_t)d.x * d.y * T * tz + (int64_t)d.x * T * hz * ty + (int64_t)T * hy * hz * tx +
           (i - tx * T) + (int64_t)wx * ((j - ty * T) + (int64_t)hy * (k - tz * T));
}

__host__ __device__ inline bool isActiveL(int l) { return l == PS_ACTIVEFLUID || l == PS_BOUNDARY; }   // Solver.h:708-710
__host__ __device__ inline bool isReducedL(int l) { return l == PS_REDUCED || l == PS_BOUNDARY; }      // Solver.h:711-713

template <class T>
struct Set7 {
    T* p[7];
};
template <class T>
struct Set8 {
    T* p[8];
};

// Ownership (multi-GPU): which entities of the local (brick + halo) grid this rank owns.  Along an axis a sample grid lives either
// IN the cell layers (offset 0.5 along that axis: cells; X faces along y and z; ...) or ON the planes between them (offset 0: X faces
// along x; XY edges along x and y; ...).  A layer q is owned iff lo <= q < hi; the plane q on a cut belongs to the rank above it, the
// last plane of the whole domain to the last rank.  (A z-slab: lo = 0, hi = n, no upper neighbour along x and y.)
struct Own {
    int enabled, lo[3], hi[3], hasUpper[3];
    __host__ __device__ bool layerA(int a, int q) const { return q >= lo[a] && q < hi[a]; }
    __host__ __device__ bool planeA(int a, int q) const { return (q >= lo[a] && q < hi[a]) || (q == hi[a] && !hasUpper[a]); }
    // on planes: faceX (1): x; faceY (2): y; faceZ (3): z; edgeYZ (4): y, z; edgeXZ (5): x, z; edgeXY (6): x, y
    __host__ __device__ static bool onPlane(int s, int a) {
        return a == 0 ? (s == 1 || s == 5 || s == 6) : (a == 1 ? (s == 2 || s == 4 || s == 6) : (s == 3 || s == 4 || s == 5));
    }
    __host__ __device__ bool along(int s, int a, int q) const { return onPlane(s, a) ? planeA(a, q) : layerA(a, q); }
    __host__ __device__ bool sample(int s, int i, int j, int k) const { return !enabled || (along(s, 0, i) && along(s, 1, j) && along(s, 2, k)); }
    __host__ __device__ bool cell(int i, int j, int k) const { return sample(0, i, j, k); }
};

#ifdef PS_AFFINE_REGIONS
// Affine basis row C_a(x), exec/HDK_PolyStokesSolver.cpp:2153-2184 (AFFINE_REGIONS, 11 DOF): branch-free selects as below.
__host__ __device__ inline void basisRow(const double ox, const double oy, const double oz, int axis, double* v) {
    const bool a0 = axis == 0, a1 = axis == 1, a2 = axis == 2;
    v[0] = a0 ? 1. : 0.;
    v[1] = a1 ? 1. : 0.;
    v[2] = a2 ? 1. : 0.;
    v[3] = a0 ? ox : (a2 ? -oz : 0.);
    v[4] = a0 ? oy : 0.;
    v[5] = a0 ? oz : 0.;
    v[6] = a1 ? ox : 0.;
    v[7] = a1 ? oy : (a2 ? -oz : 0.);
    v[8] = a1 ? oz : 0.;
    v[9] = a2 ? ox : 0.;
    v[10] = a2 ? oy : 0.;
}
__host__ __device__ inline double basisDot(const double ox, const double oy, const double oz, int axis, const double* c) {
    if (axis == 0) return c[0] + ox * c[3] + oy * c[4] + oz * c[5];
    if (axis == 1) return c[1] + ox * c[6] + oy * c[7] + oz * c[8];
    return c[2] - oz * c[3] - oz * c[7] + ox * c[9] + oy * c[10];
}
#else
// Polynomial basis row C_a(x), exec/HDK_PolyStokesSolver.cpp:2105-2149 (QUADRATIC_REGIONS, 26 DOF).
__host__ __device__ inline void basisRow(const double ox, const double oy, const double oz, int axis, double* v) {
    const double qx[9] = {ox, oy, oz, ox * ox, ox * oy, ox * oz, oy * oy, oy * oz, oz * oz};
    const bool a0 = axis == 0, a1 = axis == 1, a2 = axis == 2;
    v[0] = a0 ? 1. : 0.;
    v[1] = a1 ? 1. : 0.;
    v[2] = a2 ? 1. : 0.;
    const double z3 = -oz, z6 = -2. * ox * oz, z7 = -1. * oy * oz, z8 = -0.5 * oz * oz;
    const double z16 = -1. * ox * oz, z18 = -2. * oy * oz;
    v[3] = a0 ? qx[0] : (a2 ? z3 : 0.);
    v[4] = a0 ? qx[1] : 0.;
    v[5] = a0 ? qx[2] : 0.;
    v[6] = a0 ? qx[3] : (a2 ? z6 : 0.);
    v[7] = a0 ? qx[4] : (a2 ? z7 : 0.);
    v[8] = a0 ? qx[5] : (a2 ? z8 : 0.);
    v[9] = a0 ? qx[6] : 0.;
    v[10] = a0 ? qx[7] : 0.;
    v[11] = a0 ? qx[8] : 0.;
    v[12] = a1 ? qx[0] : 0.;
    v[13] = a1 ? qx[1] : (a2 ? z3 : 0.);
    v[14] = a1 ? qx[2] : 0.;
    v[15] = a1 ? qx[3] : 0.;
    v[16] = a1 ? qx[4] : (a2 ? z16 : 0.);
    v[17] = a1 ? qx[5] : 0.;
    v[18] = a1 ? qx[6] : (a2 ? z18 : 0.);
    v[19] = a1 ? qx[7] : (a2 ? z8 : 0.);
    v[20] = a1 ? qx[8] : 0.;
    v[21] = a2 ? ox : 0.;
    v[22] = a2 ? oy : 0.;
    v[23] = a2 ? qx[3] : 0.;
    v[24] = a2 ? qx[4] : 0.;
    v[25] = a2 ? qx[6] : 0.;
}
// dot(C_a(x), c) without materialising the row
__host__ __device__ inline double basisDot(const double ox, const double oy, const double oz, int axis, const double* c) {
    if (axis == 0)
        return c[0] + ox * c[3] + oy * c[4] + oz * c[5] + ox * ox * c[6] + ox * oy * c[7] + ox * oz * c[8] +
               oy * oy * c[9] + oy * oz * c[10] + oz * oz * c[11];
    if (axis == 1)
        return c[1] + ox * c[12] + oy * c[13] + oz * c[14] + ox * ox * c[15] + ox * oy * c[16] + ox * oz * c[17] +
               oy * oy * c[18] + oy * oz * c[19] + oz * oz * c[20];
    return c[2] - oz * c[3] - 2. * ox * oz * c[6] - oy * oz * c[7] - 0.5 * oz * oz * c[8] - oz * c[13] -
           ox * oz * c[16] - 2. * oy * oz * c[18] - 0.5 * oz * oz * c[19] + ox * c[21] + oy * c[22] +
           ox * ox * c[23] + ox * oy * c[24] + oy * oy * c[25];
}

#endif
// gv += w * C_AX(o) with the axis a compile-time constant: only the entries the row of that axis has (10 / 10 / 14 of the 26), each the
// same product and the same sum as  basisRow(...); gv[n] += w * row[n]  forms (the other entries would add w * 0): bit-identical
// sums, a third of the instructions.  The viscosity blocks are sums of thousands of these with heavy cancellation, and on stiff
// scenes the velocities at a 1e-8 tolerance follow their last bits — the summation order of the oracle stays.
template <int AX>
__host__ __device__ inline void basisAccum(const double ox, const double oy, const double oz, const double w, double* gv) {
#ifdef PS_AFFINE_REGIONS
    if (AX == 0) { gv[0] += w * 1.; gv[3] += w * ox; gv[4] += w * oy; gv[5] += w * oz; }
    else if (AX == 1) { gv[1] += w * 1.; gv[6] += w * ox; gv[7] += w * oy; gv[8] += w * oz; }
    else { gv[2] += w * 1.; gv[3] += w * (-oz); gv[7] += w * (-oz); gv[9] += w * ox; gv[10] += w * oy; }
#else
    const double qx[9] = {ox, oy, oz, ox * ox, ox * oy, ox * oz, oy * oy, oy * oz, oz * oz};
    if (AX == 0) {
        gv[0] += w * 1.;
#pragma unroll
        for (int m = 0; m < 9; ++m) gv[3 + m] += w * qx[m];
    } else if (AX == 1) {
        gv[1] += w * 1.;
#pragma unroll
        for (int m = 0; m < 9; ++m) gv[12 + m] += w * qx[m];
    } else {
        const double z3 = -oz, z6 = -2. * ox * oz, z7 = -1. * oy * oz, z8 = -0.5 * oz * oz;
        const double z16 = -1. * ox * oz, z18 = -2. * oy * oz;
        gv[2] += w * 1.;
        gv[3] += w * z3; gv[6] += w * z6; gv[7] += w * z7; gv[8] += w * z8;
        gv[13] += w * z3; gv[16] += w * z16; gv[18] += w * z18; gv[19] += w * z8;
        gv[21] += w * ox; gv[22] += w * oy; gv[23] += w * qx[3]; gv[24] += w * qx[4]; gv[25] += w * qx[6];
    }
#endif
}


// face position packed in 32 bits: 10 bits per coordinate + 2 bits axis (grids up to 1023^3)
__host__ __device__ inline uint32_t packFace(int i, int j, int k, int axis) {
    return (uint32_t)i | ((uint32_t)j << 10) | ((uint32_t)k << 20) | ((uint32_t)axis << 30);
}
__host__ __device__ inline void unpackFace(uint32_t q, int& i, int& j, int& k, int& axis) {
    i = q & 1023; j = (q >> 10) & 1023; k = (q >> 20) & 1023; axis = q >> 30;
}

// The Jacobi diagonal as the PCG kernels read it (ps_context::dinvF): 16 bits per DOF — the upper half of the fp32 value of 1 / A_jj, rounded
// to nearest even (sign — the system is assembled negative definite, with the reference's signs — 8 exponent bits, 7 mantissa bits: 0.4 %
// relative).  The preconditioner is ANY fixed diagonal of one sign: the
// recurrences, the stop test and every sum stay fp64, x converges to the same tolerance; what changes is the operator D^-1 A whose
// spectrum sets the iteration count — by the rounding of D (r05: 987 iterations at 256^3 either way).  It is read by both step kernels
// of every iteration: 8 B/DOF as fp32 — the 6 % a Jacobi iteration cost over an identity one — 4 B/DOF now.  -DPS_DIAG_FP32: the fp32 copy of r02-r05.
#ifdef PS_DIAG_FP32
typedef float diag_t;
__host__ __device__ inline double diagValue(diag_t v) { return (double)v; }
__host__ __device__ inline diag_t diagStore(double v) { return (float)v; }
#else
typedef uint16_t diag_t;
__host__ __device__ inline double diagValue(diag_t v) { return (double)__builtin_bit_cast(float, (uint32_t)v << 16); }
__host__ __device__ inline diag_t diagStore(double v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, (float)v);
    return (diag_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}
#endif

inline bool debugPoisonOn() {
    static const bool on = [] { const char* e = PS_ENV_LOUD("PS_DEBUG_POISON"); return e && atoi(e) != 0; }();
    return on;
}
// hipFree synchronises the WHOLE device — every stream of the process, other contexts' included.  Inside a step that is a stall (the setup
// grows scratch buffers while kernels are in flight), and with two ranks of one communicator in one process (tests/mp_rank.py: ranks as
// threads over the asynchronous transport) it is a deadlock: rank A blocks in hipFree behind rank B's stream, which waits for a message
// A has not enqueued yet.  A step therefore never frees: a buffer that is grown or dropped goes on the list of the context whose entry
// point the calling thread is in (ps_context::deferred, made current by SinkScope in every ABI entry; buffers freed outside any entry go
// to a process-wide orphan list), and that list is released where the context's stream has just been synchronised — at the END of every
// setup / solve / step, error paths included (ps_context::drainDeferred; r06: unconditionally — r05 kept up to 1 GiB of dead buffers on a
// process-wide list across steps of the host application) — and when the context is destroyed.
// The one case that must not release inside a step is the test arrangement above (asyncRanks > 1: several ranks with their own
// asynchronous communicator in ONE process): rank B may already wait, in its NEXT step, for a message rank A sends only after its
// hipFree returns.  There the lists wait for ps_context_destroy, and running out of memory is an error (ps_last_error), not a retry.
struct DeferredFrees { std::mutex m; std::vector<std::pair<void*, size_t>> v; size_t bytes = 0; };
struct MemState {
    std::mutex m;
    long long liveBytes = 0, peakBytes = 0;     // bytes hipMalloc'ed through DevBuf and not yet hipFree'd (deferred buffers included)
    int asyncRanks = 0;                         // live contexts that own an asynchronous communicator (ps_comm_init_rccl)
    int contexts = 0;                           // live contexts
    std::vector<DeferredFrees*> lists;          // every live context's list (+ the orphan list): the out-of-memory path releases them all
};
inline MemState& memState() { static MemState* s = new MemState; return *s; }   // never destroyed: a context may outlive the statics at process exit
inline DeferredFrees& orphanFrees() {
    static DeferredFrees* d = [] { DeferredFrees* q = new DeferredFrees; std::lock_guard<std::mutex> lk(memState().m); memState().lists.push_back(q); return q; }();
    return *d;
}
inline DeferredFrees*& currentSink() { static thread_local DeferredFrees* s = nullptr; return s; }
struct SinkScope {
    DeferredFrees* prev;
    explicit SinkScope(DeferredFrees* d) : prev(currentSink()) { if (d) currentSink() = d; }
    ~SinkScope() { currentSink() = prev; }
};
inline void memAccount(long long delta) {
    MemState& M = memState();
    std::lock_guard<std::mutex> lk(M.m);
    M.liveBytes += delta;
    if (M.liveBytes > M.peakBytes) M.peakBytes = M.liveBytes;
}
// hipFree everything on the list.  The caller has synchronised every stream that used the buffers (the context's own).
inline size_t releaseDeferred(DeferredFrees& d) {
    std::vector<std::pair<void*, size_t>> v;
    {
        std::lock_guard<std::mutex> lk(d.m);
        v.swap(d.v);
        d.bytes = 0;
    }
    size_t bytes = 0;
    for (auto& q : v) { (void)hipFree(q.first); bytes += q.second; }
    if (bytes) memAccount(-(long long)bytes);
    return bytes;
}
inline bool threadedRanks() { MemState& M = memState(); std::lock_guard<std::mutex> lk(M.m); return M.asyncRanks > 1; }
// PS_DEBUG_ALLOC_LIMIT=<bytes> (lab build, tests): an allocation that would take the bytes held through DevBuf above the limit fails as if
// the device were full — the out-of-memory paths below run without filling 288 GB
inline long long debugAllocLimit() {
    static const long long lim = [] { const char* e = PS_ENV_LOUD("PS_DEBUG_ALLOC_LIMIT"); return e ? atoll(e) : 0LL; }();
    return lim;
}
inline hipError_t mallocCounted(void** out, size_t bytes) {
    const long long lim = debugAllocLimit();
    if (lim > 0) {
        MemState& M = memState();
        std::lock_guard<std::mutex> lk(M.m);
        if (M.liveBytes + (long long)bytes > lim) { *out = nullptr; return hipErrorOutOfMemory; }
    }
    const hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) memAccount((long long)bytes); else { (void)hipGetLastError(); *out = nullptr; }
    return e;
}
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        if (count <= n && p) { poison(); return; }
        free();
        if (count == 0) count = 1;
        const size_t bytes = count * sizeof(T);
        if (mallocCounted((void**)&p, bytes) != hipSuccess) {
            // Out of memory.  Buffers may be waiting on the deferred lists: releasing them means hipFree, i.e. a device-wide synchronisation
            // in the middle of a step.  With one asynchronous rank per process (production: one process per GPU; single domain; in-process
            // groups on one stream) that is only a stall: synchronise, release every list, try once more.  With several asynchronous ranks
            // in this process it can deadlock them (see above): fail THIS rank with a message instead of hanging two.
            size_t waiting = 0;
            { MemState& M = memState(); std::lock_guard<std::mutex> lk(M.m); for (DeferredFrees* d : M.lists) { std::lock_guard<std::mutex> l2(d->m); waiting += d->bytes; } }
            if (threadedRanks())
                throw Error("out of device memory: " + std::to_string(bytes) + " bytes requested, " + std::to_string(waiting) +
                            " bytes of dropped buffers wait for release, which cannot happen inside a step while several ranks of one communicator share this process");
            if (waiting > 0) {
                (void)hipDeviceSynchronize();
                std::vector<DeferredFrees*> lists;
                { MemState& M = memState(); std::lock_guard<std::mutex> lk(M.m); lists = M.lists; }
                for (DeferredFrees* d : lists) releaseDeferred(*d);
            }
            if (mallocCounted((void**)&p, bytes) != hipSuccess) {
                p = nullptr;
                long long live; { MemState& M = memState(); std::lock_guard<std::mutex> lk(M.m); live = M.liveBytes; }
                throw Error("out of device memory: " + std::to_string(bytes) + " bytes requested with " + std::to_string(live) + " bytes held by this library" +
                            (waiting ? " (after releasing " + std::to_string(waiting) + " bytes of dropped buffers)" : std::string()));
            }
        }
        n = count;
        poison();
    }
    // PS_DEBUG_POISON=1 (tests only): whatever alloc() hands out — a fresh block or the buffer a previous step left — is filled with
    // 0xff bytes (NaN / -1) first, so that a kernel consuming words nobody wrote this step shows up in the results
    void poison() {
        if (!debugPoisonOn() || !p) return;
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemset((void*)p, 0xff, n * sizeof(T)));
        HIP_CHECK(hipDeviceSynchronize());
    }
    void free() {                       // (deferred: see DeferredFrees above)
        if (p) {
            DeferredFrees& d = currentSink() ? *currentSink() : orphanFrees();
            std::lock_guard<std::mutex> lk(d.m);
            d.v.emplace_back((void*)p, n * sizeof(T));
            d.bytes += n * sizeof(T);
        }
        p = nullptr;
        n = 0;
    }
    ~DevBuf() {                         // the context goes away: its stream has been synchronised (ps_context_destroy)
        if (p) { (void)hipFree(p); memAccount(-(long long)(n * sizeof(T))); }
        p = nullptr;
        n = 0;
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

inline int gridFor(int64_t n, int block) { return (int)((n + block - 1) / block); }

}  // namespace ps
