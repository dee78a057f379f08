// SpMV kernels: block helpers, the CSR-stream one-shot kernels and the persistent kernels on the compressed stream.
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once


constexpr int BS = 256;
constexpr int VGRID = 1024;   // capped grid for streaming vector kernels (grid-stride); 4 blocks per CU measured best

__device__ inline double waveReduceSum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// deterministic block sum (wave shuffles, then the 4 wave sums in order); result valid in thread 0
__device__ inline double blockReduceSum(double v) {
    __shared__ double ws[BS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = waveReduceSum(v);
    if (lane == 0) ws[w] = v;
    __syncthreads();
    double s = 0.;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < BS / 64; ++i) s += ws[i];
    }
    __syncthreads();
    return s;
}

// same sum (same order), valid in every thread
__device__ inline double blockSumAll(double v) {
    __shared__ double wsA[BS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = waveReduceSum(v);
    if (lane == 0) wsA[w] = v;
    __syncthreads();
    double s = 0.;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) s += wsA[i];
    __syncthreads();
    return s;
}

// Streaming phase of the CSR-stream SpMV: the block's contiguous nnz range [p0,p1) (<= BS*MAXNNZ entries)
// is read with a fixed-trip, fully unrolled loop so that all MAXNNZ (col,val) loads of a thread — and then
// all MAXNNZ gathers — are in flight together (memory-level parallelism instead of a dependent chain).
template <int SLOTS, bool PACKED>
__device__ inline void streamProducts(const int32_t* __restrict__ col, const double* __restrict__ val, const int8_t* __restrict__ code,
                                      double scale, const double* __restrict__ x, int p0, int p1, double* __restrict__ prod) {
    int c[SLOTS];
    double v[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int p = p0 + threadIdx.x + u * BS;
        const bool ok = p < p1;
        c[u] = ok ? __builtin_nontemporal_load(col + p) : -1;
        if (PACKED) v[u] = ok ? (double)__builtin_nontemporal_load(code + p) * scale : 0.;   // exact: see DevCSR::code
        else v[u] = ok ? __builtin_nontemporal_load(val + p) : 0.;
    }
    double xv[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : 0.;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u)
        if (c[u] >= 0) prod[threadIdx.x + u * BS] = v[u] * xv[u];
}

// ---- CSR-stream SpMV ------------------------------------------------------------------------------
// One-shot variant: a block owns BS consecutive rows.  (More rows per thread was tried: 2 and 4 rows per thread are
// 5-100 % slower — registers and LDS cost more occupancy than the extra loads in flight buy.)
// MODE 0: out[row] = (row < nA ? dt*McInv[row] : 1) * (S x)[row]     (operator, forward half)
// MODE 1: out[row] = (S x)[row]                                       (velocity recovery)
template <int MODE, int MAXNNZ, bool PACKED>
__global__ void __launch_bounds__(BS) k_spmv_S(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                                               const int8_t* __restrict__ code, double scale, const double* __restrict__ x, int rows, int nA,
                                               double dt, const double* __restrict__ McInv, double* __restrict__ out,
                                               const int* __restrict__ done) {
    if (done && *done) return;
    constexpr int RPT = 1;
    __shared__ double prod[BS * MAXNNZ * RPT];
    const int r0 = blockIdx.x * (BS * RPT);
    const int r1 = min(r0 + BS * RPT, rows);
    const int p0 = ptr[r0], p1 = ptr[r1];
    // per-row loads that do not depend on the stream: issue them first
    int pa[RPT], pb[RPT];
    double sc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = r0 + threadIdx.x + q * BS;
        const bool ok = row < rows;
        pa[q] = ok ? ptr[row] : 0;
        pb[q] = ok ? ptr[row + 1] : 0;
        sc[q] = (MODE == 0 && ok && row < nA) ? dt * McInv[row] : 1.;
    }
    streamProducts<MAXNNZ * RPT, PACKED>(col, val, code, scale, x, p0, p1, prod);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = r0 + threadIdx.x + q * BS;
        if (row < rows) {
            double s = 0.;
            for (int e = pa[q] - p0; e < pb[q] - p0; ++e) s += prod[e];
            out[row] = s * sc[q];
        }
    }
}
// MODE 0: out[j] = -(St t)[j] - 0.5*uInv[j]*xin[j];  partial[block] = sum xin[j]*out[j]
// MODE 1: out[j] = -(St t)[j] + add[j]                                   (right-hand side b)
template <int MODE, int MAXNNZ, bool PACKED>
__global__ void __launch_bounds__(BS) k_spmv_St(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                const int8_t* __restrict__ code, double scale, const double* __restrict__ t, int rows, int nP,
                                                const double* __restrict__ uInv, const double* __restrict__ xin, const double* __restrict__ add,
                                                double* __restrict__ out, double* __restrict__ partial, const int* __restrict__ done) {
    if (done && *done) return;
    constexpr int RPT = 1;
    __shared__ double prod[BS * MAXNNZ * RPT];
    const int r0 = blockIdx.x * (BS * RPT);
    const int r1 = min(r0 + BS * RPT, rows);
    const int p0 = ptr[r0], p1 = ptr[r1];
    int pa[RPT], pb[RPT];
    double e0[RPT], e1[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = r0 + threadIdx.x + q * BS;
        const bool ok = row < rows;
        pa[q] = ok ? ptr[row] : 0;
        pb[q] = ok ? ptr[row + 1] : 0;
        if (MODE == 0) { e0[q] = ok ? xin[row] : 0.; e1[q] = ok ? uInv[row] : 0.; }   // uInv is full length (0 on pressure rows)
        else { e0[q] = ok ? add[row] : 0.; e1[q] = 0.; }
    }
    streamProducts<MAXNNZ * RPT, PACKED>(col, val, code, scale, t, p0, p1, prod);
    __syncthreads();
    double d = 0.;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = r0 + threadIdx.x + q * BS;
        if (row < rows) {
            double s = 0.;
            for (int e = pa[q] - p0; e < pb[q] - p0; ++e) s += prod[e];
            double y;
            if (MODE == 0) {
                y = -s;
                y -= 0.5 * e1[q] * e0[q];
                d += e0[q] * y;
            } else {
                y = -s + e0[q];
            }
            out[row] = y;
        }
    }
    if (MODE == 0) {
        const double bs = blockReduceSum(d);
        if (threadIdx.x == 0) partial[blockIdx.x] = bs;
    }
}


// ---- persistent, software-pipelined kernels on the compressed stream -----------------------------------
// PMC (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 85 %) shows the one-shot kernels above are latency bound: every block walks
// three dependent memory round trips (row-pointer bounds -> (col,val) stream -> gather) at the occupancy cap of
// 8 waves/SIMD.  Here a block loops over row chunks (grid = #CUs x 16) and, while the gathers / LDS reduction of
// chunk i are in flight, the stream of chunk i+1 is already loading into a second register set and the bounds of
// chunk i+2 are being fetched.  They read the compressed form of the matrix built by ps_context::buildCol16:
//   * per chunk (<= 256 consecutive rows, starting at lattice-block / tile boundaries) a 4-entry-aligned run of (16-bit windowed
//     column, int8 value code): 3 B per entry, fetched as one 8-byte + one 4-byte load per lane for 4 consecutive entries — chunks
//     whose runs are byte-identical share ONE run (ps_blocks.hip: k_chunk_share), which then comes from cache,
//   * 16 window bases and a Chunk record (below) per chunk, one row-length byte per row (prefix-summed in the block)
// and reproduce the fp64 CSR product bit for bit (same values, same summation order within a row).
//
// chunk walk of a persistent block.  Plain: chunk = block + it * grid.  Grouped (G = xcdAware > 0): workgroups b, b+8, ...
// run on XCD b & 7 (verified with s_getreg HW_REG_XCC_ID), so runs of G consecutive chunks are dealt to the XCDs round
// robin — rows that gather the same lines of x (k-plane neighbours, a few chunks apart) then share ONE L2, while
// the chip as a whole still sweeps one compact window of memory.
// MODE 2 of the St kernel: one term of the Chebyshev preconditioner fused into the epilogue (ps_context::chebyshevApply), in its
// three-term form (no difference vector to keep):  Az = (A z_j)[row];  z_{j+1} = z_j + c1 (z_j - z_{j-1}) + c2 dinv (r - Az);
// partial += r z_{j+1}.  z_j is the kernel's x (xin), z_{j-1} is read from zprev (null: zero) and z_{j+1} goes to `out` — the
// caller passes the z_{j-1} buffer: every row is read and then written by the same thread.
struct ChebArgs { const double* r; const diag_t* dinv; const double* zprev; double c1, c2; };   // dinv: the stored diagonal (ps_common.hpp)
// MODE 3 of the St kernel: the residual update of the PCG step inside the epilogue.  x . A x is known BEFORE the kernel starts,
// from the factored form:  x.Ax = -( sum_active s_f t_f  +  sum_tiles w.v  +  1/2 sum_j uInv_j x_j^2 )  (partials of the S kernel,
// of the tile kernel and of k_cg_update_xp) — so every workgroup forms alpha itself and does r -= alpha (A p) on its rows with
// (A p)[row] still in a register: A p is never written or read back, and k_cg_update_r is not launched.
struct FusedR {
    CGScalars* sc;
    const double* sPart; int sCount;      // k_spmv_S_pipe: active-face share
    const double* tPart; int tCount;      // k_tile_apply: one value per region
    const double* uPart; int uCount;      // k_cg_update_xp / k_uinv_pp: 1/2 sum uInv p^2 is formed here (partials hold sum uInv p^2)
    const double* xxPart; int xxCount;    // ||x||^2 partials of the previous k_cg_update_xp (stop test of the previous iteration)
    int it;
    double* r; const diag_t* dinvF;       // residual (updated in place), stored Jacobi diagonal (ps_common.hpp: diag_t; null: identity)
    double* rPart;                        // out: partials of r.r at [block], of r.z at [gridDim + block]
    // Chebyshev preconditioner: the polynomial's first term on the new r, z_1 = dinv r / theta -> cz (null: not asked for; then r.z
    // above is that of the Jacobi diagonal)
    const diag_t* dinvC; double invTheta; double* cz;
    // Slab decomposition (ps_dist.hpp): red = {S + T + 1/2 U summed over the ranks, ||x||^2 over the ranks} replaces the partial
    // sums above; only rows in [ownLo, ownHi) are this rank's DOFs — the others (halo DOFs) carry contributions to a neighbour's
    // rows: their y goes to yOut and the owner subtracts alpha times it afterwards (k_dist_fixup).  Single domain: red = null,
    // [0, rows), yOut = null.
    const double* red; int ownLo, ownHi; double* yOut;
    // the kernel may run as two launches (rows next to a cut first, the rest under the exchange: ps_dist.hpp): partials of r.r at
    // rPart[block], of r.z at rPart[rStride + block]; 0 = one launch (stride = its grid)
    int rStride;
};
// Walk of a persistent workgroup over the chunk ids: runs of G = 1 << sh consecutive chunks are dealt to the 8 XCDs round robin
// (workgroup b runs on XCD b & 7), inside an XCD to its workgroups in order; sh < 0: plain grid-stride walk
struct ChunkWalk {
    int sh, x, l, per, rs;   // rs: a workgroup takes runs of 1 << rs CONSECUTIVE chunks (bits 16.. of the launch parameter): its waves then find the
                             // lines the previous chunk gathered in the CU's L1
    __device__ explicit ChunkWalk(int g)
        : sh((g & 0xffff) > 0 ? 31 - __builtin_clz((unsigned)(g & 0xffff)) : -1), x(blockIdx.x & 7), l(blockIdx.x >> 3), per(gridDim.x >> 3), rs((g >> 16) & 7) {}
    __device__ int at(int it) const {
        const int j = it >> rs, o = it & ((1 << rs) - 1);
        int run;
        if (sh < 0) run = blockIdx.x + j * gridDim.x;
        else {
            const int q = l + j * per;
            run = ((((q >> sh) << 3) + x) << sh) + (q & ((1 << sh) - 1));
        }
        return (run << rs) + o;
    }
};
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// Every access of the loop body goes through a buffer descriptor (buffer_load/store ... offen): 32-bit byte offsets
// instead of 64-bit address arithmetic, and hardware bounds checking (a load past `bytes` returns 0, a store is dropped),
// so the body has NO branches: lanes past the end of a chunk / of the rows load and multiply harmless values into LDS
// slots no row reads.  (Arrays must be < 4 GiB: checked by ps_context::buildCol16.)
__device__ inline __amdgpu_buffer_rsrc_t bufRsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ inline double bufLoadF64(__amdgpu_buffer_rsrc_t r, unsigned byteOff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byteOff, 0, 0));
}
template <bool NT> __device__ inline double bufLoadF64epi(__amdgpu_buffer_rsrc_t r, unsigned byteOff);
__device__ inline void bufStoreF64(__amdgpu_buffer_rsrc_t r, unsigned byteOff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)byteOff, 0, 0);
}
// Cache policies of the three access classes (compile-time, scripts/build_variant.sh builds A/B copies).  Measured at 256^3
// (profiles/r02_cache_policy.txt): result stores nt (aux 2: written once, read by the NEXT kernel; St -7 %, S -3 % against
// default-policy stores; sc1 / sc0 sc1 stores +4 %); stream loads nt (sc1 / sc0 sc1 / default within 1 %); gathers default
// policy — nt gathers bypass the CU's L1 and run 2x slower: what cross-row reuse there is comes from L1.
#ifndef PS_STORE_AUX
#define PS_STORE_AUX 2
#endif
#ifndef PS_STREAM_AUX
#define PS_STREAM_AUX 2
#endif
#ifndef PS_GATHER_AUX
#define PS_GATHER_AUX 0
#endif
#ifndef PS_EPI_AUX
#define PS_EPI_AUX 2       // the per-row streams of the epilogues (row length, x, uInv / codes): read once per launch -> nt (St -3 %)
#endif
#ifndef PS_DIAG_AUX
#define PS_DIAG_AUX PS_EPI_AUX   // the stored Jacobi diagonal in the residual-update epilogue (read again by the x/p update half an iteration later)
#endif
#ifndef PS_UC_AUX
#define PS_UC_AUX PS_EPI_AUX     // the uInv codes in the two-unit St kernels (read again by the x/p update)
#endif
// POL (template parameter of the pipelined kernels), bit 0: the non-temporal policies above for the result stores and the per-row
// epilogue streams; bit 1: for the matrix stream (col16 / code4 / val4).  Chosen per system (Launch::policy): streams that are
// touched once per launch should not sweep the caches of a 45 M-row system, but a system that fits in the 256 MB memory-side
// cache (or nearly) is served from it between kernels if they are allowed to stay; and a matrix stream whose runs are SHARED
// between chunks (DevCSR::uniqueLen) is read again and again — it must stay cached (non-temporal it is 11 % slower than unshared).
// entry `row` of the stored Jacobi diagonal (diag_t: 16 bits, the upper half of an fp32 value; fp32 under -DPS_DIAG_FP32) as a float — 0 past the buffer
template <bool NT> __device__ inline float bufLoadDiag(__amdgpu_buffer_rsrc_t r, unsigned row) {
#ifdef PS_DIAG_FP32
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(row * 4u), 0, NT ? PS_DIAG_AUX : 0));
#else
    const uint32_t h = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)(row * 2u), 0, NT ? PS_DIAG_AUX : 0);
    return __builtin_bit_cast(float, h << 16);
#endif
}
template <bool NT> __device__ inline double bufLoadF64epi(__amdgpu_buffer_rsrc_t r, unsigned byteOff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byteOff, 0, NT ? PS_EPI_AUX : 0));
}
template <bool NT> __device__ inline void bufStoreF64nt(__amdgpu_buffer_rsrc_t r, unsigned byteOff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)byteOff, 0, NT ? PS_STORE_AUX : 0);
}
__device__ inline double bufGatherF64(__amdgpu_buffer_rsrc_t r, unsigned byteOff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byteOff, 0, PS_GATHER_AUX));
}
// Element type of a vector a kernel gathers from / streams: double everywhere except the INNER vectors of the Chebyshev polynomial in its
// single-precision form (PS_PRE_CHEBYSHEV_F32: z_j and the face-row vector t of the polynomial's operator applies are STORED as fp32 — half the
// bytes of every gather and epilogue stream, half the lines a unit's gathers touch; every product, sum and recurrence stays fp64 in registers).
// Index = element index (the byte offset is formed here); loads past the buffer return 0, stores are dropped (buffer descriptors).
template <class T> struct VecIO;
template <> struct VecIO<double> {
    static constexpr unsigned BYTES = 8;
    __device__ static inline double gather(__amdgpu_buffer_rsrc_t r, unsigned i) { return bufGatherF64(r, i * 8u); }
    template <bool NT> __device__ static inline double loadEpi(__amdgpu_buffer_rsrc_t r, unsigned i) { return bufLoadF64epi<NT>(r, i * 8u); }
    __device__ static inline double load(__amdgpu_buffer_rsrc_t r, unsigned i) { return bufLoadF64(r, i * 8u); }
    template <bool NT> __device__ static inline void store(__amdgpu_buffer_rsrc_t r, unsigned i, double v) { bufStoreF64nt<NT>(r, i * 8u, v); }
    __device__ static inline double stored(double v) { return v; }              // the value a later kernel reads back
};
template <> struct VecIO<float> {
    static constexpr unsigned BYTES = 4;
    __device__ static inline double gather(__amdgpu_buffer_rsrc_t r, unsigned i) { return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4u), 0, PS_GATHER_AUX)); }
    template <bool NT> __device__ static inline double loadEpi(__amdgpu_buffer_rsrc_t r, unsigned i) { return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4u), 0, NT ? PS_EPI_AUX : 0)); }
    __device__ static inline double load(__amdgpu_buffer_rsrc_t r, unsigned i) { return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4u), 0, 0)); }
    template <bool NT> __device__ static inline void store(__amdgpu_buffer_rsrc_t r, unsigned i, double v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r, (int)(i * 4u), 0, NT ? PS_STORE_AUX : 0);
    }
    __device__ static inline double stored(double v) { return (double)(float)v; }
};
// one lane's share of a chunk's stream: NV groups of 4 consecutive entries (non-temporal: read once)
// F64 = false: int8 value codes (3 B per entry with the column);  F64 = true: the fp64 values themselves in the same 4-aligned
// chunk layout (10 B per entry) — the form that runs when the stencil values are not code * scale (user-supplied weights)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NV, bool F64> struct Stream4;
template <int NV> struct Stream4<NV, false> { u32x2 c[NV]; unsigned v[NV]; };
template <int NV> struct Stream4<NV, true> { u32x2 c[NV]; };   // the fp64 values are fetched for the CURRENT chunk only (below):
// double-buffering 16 more VGPR pairs per lane costs more occupancy than the prefetch buys
template <int NV> struct Vals4 { u32x4 v[NV][2]; };
template <int NV, bool NT>
__device__ inline void loadVals4(const double* val4, int p0, int p1, Vals4<NV>& s) {
    const __amdgpu_buffer_rsrc_t rVal = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(val4 + p0), 0, (int)(((unsigned)(p1 - p0 + 3) & ~3u) * 8u), 0x00020000);
#pragma unroll
    for (int w = 0; w < NV; ++w) {
        const unsigned rel = 4u * (threadIdx.x + w * BS) * 8u;                   // byte offset inside this chunk's values
        const bool in = (int)(p0 + 4 * (threadIdx.x + w * BS)) < p1;
        s.v[w][0] = __builtin_amdgcn_raw_buffer_load_b128(rVal, in ? (int)rel : -1, 0, NT ? PS_STREAM_AUX : 0);
        s.v[w][1] = __builtin_amdgcn_raw_buffer_load_b128(rVal, in ? (int)(rel + 16u) : -1, 0, NT ? PS_STREAM_AUX : 0);
    }
}
// (the fp64 value array can exceed the 4 GiB a buffer descriptor spans: its descriptor is rebuilt per chunk on the chunk's base)
template <int NV, bool F64, bool NT>
__device__ inline void loadStream4(__amdgpu_buffer_rsrc_t rCol, __amdgpu_buffer_rsrc_t rCode, const double* val4, int p0, int p1, Stream4<NV, F64>& s,
                                   unsigned tid = threadIdx.x) {               // tid: the thread's index inside its 256-thread group
#pragma unroll
    for (int w = 0; w < NV; ++w) {
        const unsigned first = (unsigned)p0 + 4u * (tid + w * BS);     // p0 is a multiple of 4
        // groups past the end of the chunk: offset 0xffffffff is out of range -> zeros without a memory access
        // (zeros decode to window 0 / offset 0 / value 0, like the padding inside the last group)
        const bool in = (int)first < p1;
        s.c[w] = __builtin_amdgcn_raw_buffer_load_b64(rCol, in ? (int)(first * 2u) : -1, 0, NT ? PS_STREAM_AUX : 0);
        if constexpr (!F64) s.v[w] = __builtin_amdgcn_raw_buffer_load_b32(rCode, in ? (int)first : -1, 0, NT ? PS_STREAM_AUX : 0);
    }
}
__device__ inline unsigned streamCol(u32x2 c, int j, int myBase) {   // window base (lane `window` of every 16-lane group) + 12-bit offset
    const unsigned w = j < 2 ? c.x : c.y;
    const unsigned raw = (w >> (16 * (j & 1))) & 0xffffu;
    return (unsigned)__shfl(myBase, (int)(raw >> 12), 16) + (raw & 4095u);
}
__device__ inline double streamVal(unsigned v, int j, double scale) {   // exact: see DevCSR::code
    return (double)((int)(v << (24 - 8 * j)) >> 24) * scale;
}
__device__ inline double streamVal(const u32x4 (&v)[2], int j, double) {
    const u32x4 q = v[j >> 1];
    const u32x2 h = (j & 1) ? u32x2{q.z, q.w} : u32x2{q.x, q.y};
    return __builtin_bit_cast(double, h);
}
template <bool F64, class S, class V>
__device__ inline double prodVal(const S& cur, const V& cv, int w, int j, double scale) {
    if constexpr (F64) return streamVal(cv.v[w], j, scale); else return streamVal(cur.v[w], j, scale);
}
// inclusive prefix sum over the 64 lanes with DPP moves (VALU only, no LDS round trips): Hillis-Steele inside each row of
// 16 lanes (row_shr 1,2,4,8; lanes without a source keep 0), then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3
__device__ inline int waveInclusiveScan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
// wave sum of a double with DPP moves only (no LDS-pipe bpermutes): same row_shr / row_bcast ladder as the scan above on the two
// 32-bit halves; the total lands in lane 63 (the other lanes hold partial prefix sums)
__device__ inline double waveSumToLane63(double v) {
#define PS_DPP_STEP(CTRL, RMASK)                                                                                   \
    {                                                                                                               \
        const long long b = __double_as_longlong(v);                                                                \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, RMASK, 0xf, false);            \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, RMASK, 0xf, false);                     \
        v += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                                            \
    }
    PS_DPP_STEP(0x111, 0xf) PS_DPP_STEP(0x112, 0xf) PS_DPP_STEP(0x114, 0xf) PS_DPP_STEP(0x118, 0xf)
    PS_DPP_STEP(0x142, 0xa) PS_DPP_STEP(0x143, 0xc)
#undef PS_DPP_STEP
    return v;
}
// sum of the row's products prod[ea .. ea+len) in entry order; all (<= ML) LDS reads are issued up front
template <int ML, int PL>
__device__ inline double rowSum(const double* prod, int ea, int len) {
    double v[ML];
#pragma unroll
    for (int k = 0; k < ML; ++k) { const int e = min(ea + k, 4 * PL - 1); v[k] = prod[(e & 3) * PL + (e >> 2)]; }
    double s = 0.;
#pragma unroll
    for (int k = 0; k < ML; ++k) s = k < len ? s + v[k] : s;
    return s;
}
// chunkInfo entry (DevCSR::chunkInfo): x = first entry of the run, y = entries | rows << 16, z = first row, w = first row of the
// chunk whose run this is (== z unless the run is shared: the per-row BYTE streams — lengths, value-set codes — are read there too)
struct Chunk { int q0, q1, row0, rows, src; };
__host__ __device__ inline Chunk decodeChunk(int4 v) { return Chunk{v.x, v.x + (v.y & 0xffff), v.z, (int)((unsigned)v.y >> 16), v.w}; }
constexpr unsigned ROW_NONE = 0x1fffffffu;   // row index of an idle lane: beyond any array (rows * 8 < 4 GiB), positive as an int
// Both kernels: gathers of the current chunk, prefetch of the next, products to LDS (entry e of the chunk at
// prod[(e & 3) * PL + (e >> 2)]: conflict-free writes), row offsets from the length bytes (wave scans + 4 wave totals).
template <int MODE, int NV, bool F64, int POL>
__global__ void __launch_bounds__(BS) k_spmv_S_pipe(const uint16_t* __restrict__ col16, const int8_t* __restrict__ code4, const double* __restrict__ val4, int streamLen,
                                                    const int32_t* __restrict__ winBase, const int4* __restrict__ chunkInfo,
                                                    const uint8_t* __restrict__ len8, double scale, const double* __restrict__ x, int cols, int rows,
                                                    int nA, double dt, const double* __restrict__ McInv, double* __restrict__ out,
                                                    const int* __restrict__ done, int nChunks, int xcdAware,
                                                    const uint8_t* __restrict__ mcCode, const double* __restrict__ mcDict, double* __restrict__ stPart) {
    // stPart (MODE 0, may be null): per workgroup, the sum over its ACTIVE rows of s_f t_f = dt McInv_f s_f^2 — the active-face
    // share of x . A x, so that the residual update can run inside the St kernel (ps_solve.hip: fused step)
    if (done && *done) return;
    constexpr int PL = BS * NV;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;   // stores + per-row epilogue streams | the matrix stream
    constexpr bool BNT = NT && SNT;                               // the per-row byte streams go with the runs: cached when runs are shared
    __shared__ double prod[4 * PL];
    __shared__ __align__(16) int wtot[BS / 64];
    __shared__ double dict[MODE == 0 ? 256 : 1];      // value-set coded McInv (ps_context.hpp: mcCode): first read after the loop's first barrier
    if (MODE == 0 && mcCode) dict[threadIdx.x] = mcDict[threadIdx.x];
    static_assert(BS == 256, "four waves per block");
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(col16, (size_t)streamLen * 2), rCode = bufRsrc(code4, F64 ? 0 : (size_t)streamLen),
                                 rLen = bufRsrc(len8, (size_t)rows), rX = bufRsrc(x, (size_t)cols * 8), rMc = bufRsrc(McInv, (size_t)nA * 8),
                                 rMcc = bufRsrc(mcCode, mcCode ? (size_t)nA : 0), rOut = bufRsrc(out, (size_t)rows * 8);
    const ChunkWalk W(xcdAware);
    int it = 0;
    int chunk = W.at(0);
    if (chunk >= nChunks) { if (MODE == 0 && stPart && threadIdx.x == 0) stPart[blockIdx.x] = 0.; return; }
    double stAcc = 0.;
    Chunk pr = decodeChunk(chunkInfo[chunk]);
    Stream4<NV, F64> cur, nxt;
    loadStream4<NV, F64, SNT>(rCol, rCode, val4, pr.q0, pr.q1, cur);
    int myBase = winBase[chunk * 16 + (threadIdx.x & 15)], nBase = 0;
    int nchunk = W.at(1);
    Chunk npr{0, 0, 0, 0, 0};
    if (nchunk < nChunks) npr = decodeChunk(chunkInfo[nchunk]);
    while (true) {
        const bool live = (int)threadIdx.x < pr.rows;                                   // lanes past the chunk's rows: every access out of range
        const unsigned row = live ? (unsigned)pr.row0 + threadIdx.x : ROW_NONE, srow = live ? (unsigned)pr.src + threadIdx.x : ROW_NONE;
        const int len = (int)__builtin_amdgcn_raw_buffer_load_b8(rLen, (int)srow, 0, BNT ? PS_EPI_AUX : 0);   // 0 past the last row
        double sc = 1.;
        int mcc = 0;
        if (MODE == 0) {
            if (mcCode) mcc = (int)__builtin_amdgcn_raw_buffer_load_b8(rMcc, (int)srow, 0, BNT ? PS_EPI_AUX : 0);
            else { const double m = bufLoadF64(rMc, row * 8u); sc = (int)row < nA ? dt * m : 1.; }
        }
        double xv[4 * NV];
#pragma unroll
        for (int w = 0; w < NV; ++w) {
            if (w > 0 && pr.q0 + 4 * w * BS >= pr.q1) break;                  // block-uniform: this group of the chunk is empty
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[4 * w + j] = bufGatherF64(rX, streamCol(cur.c[w], j, myBase) * 8u);
        }
        Vals4<F64 ? NV : 0> cv;
        if constexpr (F64) loadVals4<NV, SNT>(val4, pr.q0, pr.q1, cv);
        const bool hasNext = nchunk < nChunks;
        if (hasNext) {
            loadStream4<NV, F64, SNT>(rCol, rCode, val4, npr.q0, npr.q1, nxt);
            nBase = winBase[nchunk * 16 + (threadIdx.x & 15)];
        }
        const int nn = W.at(it + 2);
        Chunk nnpr{0, 0, 0, 0, 0};
        if (nn < nChunks) nnpr = decodeChunk(chunkInfo[nn]);
#pragma unroll
        for (int w = 0; w < NV; ++w) {
            if (w > 0 && pr.q0 + 4 * w * BS >= pr.q1) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) prod[j * PL + threadIdx.x + w * BS] = prodVal<F64>(cur, cv, w, j, scale) * xv[4 * w + j];
        }
        const int incl = waveInclusiveScan(len);
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        {
            const int4 wt = *reinterpret_cast<const int4*>(wtot);
            const int wv = threadIdx.x >> 6;
            const int ea = incl - len + (wv > 0 ? wt.x : 0) + (wv > 1 ? wt.y : 0) + (wv > 2 ? wt.z : 0);
            const double s = rowSum<8, PL>(prod, ea, len);
            if (MODE == 0 && mcCode) sc = (int)row < nA ? dt * dict[mcc] : 1.;
            if (MODE == 0) stAcc += (int)row < nA ? s * (s * sc) : 0.;
            bufStoreF64nt<NT>(rOut, row * 8u, s * sc);                             // dropped past the last row
        }
        __syncthreads();
        if (!hasNext) break;
        chunk = nchunk; pr = npr; cur = nxt; myBase = nBase;
        nchunk = nn; npr = nnpr;
        ++it;
    }
    if (MODE == 0 && stPart) {
        const double bs = blockReduceSum(stAcc);
        if (threadIdx.x == 0) stPart[blockIdx.x] = bs;
    }
}
template <int MODE, int NV, bool F64, int POL>
__global__ void __launch_bounds__(BS) k_spmv_St_pipe(const uint16_t* __restrict__ col16, const int8_t* __restrict__ code4, const double* __restrict__ val4, int streamLen,
                                                     const int32_t* __restrict__ winBase, const int4* __restrict__ chunkInfo,
                                                     const uint8_t* __restrict__ len8, double scale, const double* __restrict__ t, int cols, int rows,
                                                     const double* __restrict__ uInv, const double* __restrict__ xin, const double* __restrict__ add,
                                                     double* __restrict__ out, double* __restrict__ partial, const int* __restrict__ done,
                                                     int nChunks, int xcdAware, ChebArgs cheb,
                                                     const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, FusedR fr) {
    if (done && *done) return;
    constexpr int PL = BS * NV;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;   // stores + per-row epilogue streams | the matrix stream
    constexpr bool BNT = NT && SNT;                               // the per-row byte streams go with the runs: cached when runs are shared
    __shared__ double prod[4 * PL];
    __shared__ __align__(16) int wtot[BS / 64];
    __shared__ double dict[MODE != 1 ? 256 : 1];      // value-set coded uInv (ps_context.hpp: uCode)
    if (MODE != 1 && uCode) dict[threadIdx.x] = uDict[threadIdx.x];
    double alpha = 0.;
    if (MODE == 3) {
        // same prologue as k_cg_update_r: [stop test of iteration it-1], alpha = rsold / p.Ap — identical in every workgroup
        CGScalars* sc = fr.sc;
        auto sumArr = [&](const double* a, int cnt) { double acc = 0.; for (int i = threadIdx.x; i < cnt; i += BS) acc += a[i]; return blockSumAll(acc); };
        const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
        if (fr.it > 0) {
            const double xx = fr.red ? fr.red[1] : sumArr(fr.xxPart, fr.xxCount);
            const double rr = sc->rr;
            double rre = rr;                               // pcg.h:319-325
            if (rr / xx < rre) rre = rr / xx;
            const bool fire = rre < sc->tol2;
            if (writer) { sc->xx = xx; sc->rre = rre; if (fire) { sc->done = 1; sc->iter = fr.it - 1; } }
            if (fire) return;                              // same verdict in every workgroup
        }
        const double pAp = fr.red ? -fr.red[0] : -(sumArr(fr.sPart, fr.sCount) + sumArr(fr.tPart, fr.tCount) + 0.5 * sumArr(fr.uPart, fr.uCount));
        alpha = sc->rsold2[fr.it & 1] / pAp;               // pcg.h:314
        if (writer) { sc->pAp = pAp; sc->alpha = alpha; }
    }
    static_assert(BS == 256, "four waves per block");
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(col16, (size_t)streamLen * 2), rCode = bufRsrc(code4, F64 ? 0 : (size_t)streamLen),
                                 rLen = bufRsrc(len8, (size_t)rows), rT = bufRsrc(t, (size_t)cols * 8),
                                 rE0 = bufRsrc(MODE == 1 ? add : xin, (size_t)rows * 8), rE1 = bufRsrc(uInv, (size_t)rows * 8),
                                 rOut = bufRsrc(out, (size_t)rows * 8),
                                 rCr = bufRsrc(cheb.r, MODE == 2 ? (size_t)rows * 8 : 0), rCi = bufRsrc(cheb.dinv, MODE == 2 ? (size_t)rows * sizeof(diag_t) : 0),
                                 rCd = bufRsrc(cheb.zprev, (MODE == 2 && cheb.zprev) ? (size_t)rows * 8 : 0), rUc = bufRsrc(uCode, uCode ? (size_t)rows : 0),
                                 rFr = bufRsrc(fr.r, MODE == 3 ? (size_t)rows * 8 : 0), rFd = bufRsrc(fr.dinvF, (MODE == 3 && fr.dinvF) ? (size_t)rows * sizeof(diag_t) : 0),
                                 rF64 = bufRsrc(fr.dinvC, (MODE == 3 && fr.cz) ? (size_t)rows * sizeof(diag_t) : 0), rFcz = bufRsrc(fr.cz, (MODE == 3 && fr.cz) ? (size_t)rows * 8 : 0),
                                 rFy = bufRsrc(fr.yOut, (MODE == 3 && fr.yOut) ? (size_t)rows * 8 : 0);
    const ChunkWalk W(xcdAware);
    int it = 0;
    int chunk = W.at(0);
    if (chunk >= nChunks) {
        if ((MODE == 0 || MODE == 2) && threadIdx.x == 0) partial[blockIdx.x] = 0.;
        if (MODE == 3 && threadIdx.x == 0) { fr.rPart[blockIdx.x] = 0.; fr.rPart[gridDim.x + blockIdx.x] = 0.; }
        return;
    }
    double dacc = 0., dacc2 = 0.;
    Chunk pr = decodeChunk(chunkInfo[chunk]);
    Stream4<NV, F64> cur, nxt;
    loadStream4<NV, F64, SNT>(rCol, rCode, val4, pr.q0, pr.q1, cur);
    int myBase = winBase[chunk * 16 + (threadIdx.x & 15)], nBase = 0;
    int nchunk = W.at(1);
    Chunk npr{0, 0, 0, 0, 0};
    if (nchunk < nChunks) npr = decodeChunk(chunkInfo[nchunk]);
    while (true) {
        const bool live = (int)threadIdx.x < pr.rows;                                   // lanes past the chunk's rows: every access out of range
        const unsigned row = live ? (unsigned)pr.row0 + threadIdx.x : ROW_NONE, srow = live ? (unsigned)pr.src + threadIdx.x : ROW_NONE;
        const int len = (int)__builtin_amdgcn_raw_buffer_load_b8(rLen, (int)srow, 0, BNT ? PS_EPI_AUX : 0);   // 0 past the last row
        const double e0 = bufLoadF64epi<NT>(rE0, row * 8u);                                       // x (MODE 0, 2) / the vector added (MODE 1)
        double e1 = 0., cr = 0., ci = 0., cd = 0.;
        int uc = 0;
        if (MODE != 1) { if (uCode) uc = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)srow, 0, BNT ? PS_EPI_AUX : 0); else e1 = bufLoadF64epi<NT>(rE1, row * 8u); }
        if (MODE == 2) { cr = bufLoadF64(rCr, row * 8u); ci = (double)bufLoadDiag<false>(rCi, row); cd = bufLoadF64(rCd, row * 8u); }   // cd = z_{j-1} (0: no buffer)
        float fdv = 1.f;
        if (MODE == 3) {
            cr = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rFr, (int)(row * 8u), 0, NT ? PS_EPI_AUX : 0));
            if (fr.dinvF) fdv = bufLoadDiag<NT>(rFd, row);
            if (fr.cz) ci = (double)bufLoadDiag<NT>(rF64, row);
        }
        double xv[4 * NV];
#pragma unroll
        for (int w = 0; w < NV; ++w) {
            if (w > 0 && pr.q0 + 4 * w * BS >= pr.q1) break;                  // block-uniform: this group of the chunk is empty
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[4 * w + j] = bufGatherF64(rT, streamCol(cur.c[w], j, myBase) * 8u);
        }
        Vals4<F64 ? NV : 0> cv;
        if constexpr (F64) loadVals4<NV, SNT>(val4, pr.q0, pr.q1, cv);
        const bool hasNext = nchunk < nChunks;
        if (hasNext) {
            loadStream4<NV, F64, SNT>(rCol, rCode, val4, npr.q0, npr.q1, nxt);
            nBase = winBase[nchunk * 16 + (threadIdx.x & 15)];
        }
        const int nn = W.at(it + 2);
        Chunk nnpr{0, 0, 0, 0, 0};
        if (nn < nChunks) nnpr = decodeChunk(chunkInfo[nn]);
#pragma unroll
        for (int w = 0; w < NV; ++w) {
            if (w > 0 && pr.q0 + 4 * w * BS >= pr.q1) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) prod[j * PL + threadIdx.x + w * BS] = prodVal<F64>(cur, cv, w, j, scale) * xv[4 * w + j];
        }
        const int incl = waveInclusiveScan(len);
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        {
            const int4 wt = *reinterpret_cast<const int4*>(wtot);
            const int wv = threadIdx.x >> 6;
            const int ea = incl - len + (wv > 0 ? wt.x : 0) + (wv > 1 ? wt.y : 0) + (wv > 2 ? wt.z : 0);
            const double s = rowSum<6, PL>(prod, ea, len);
            if (MODE != 1 && uCode) e1 = dict[uc];
            double y;
            if (MODE == 0) { y = -s; y -= 0.5 * e1 * e0; dacc += e0 * y; }   // p.Ap: running sum over this block's chunks (0 past the last row)
            else if (MODE == 1) y = -s + e0;
            else if (MODE == 3) {
                y = -s; y -= 0.5 * e1 * e0;                                      // (A p)[row], not stored
                const bool mine = (int)row >= fr.ownLo && (int)row < fr.ownHi;   // (idle lanes: false)
                if (fr.yOut) bufStoreF64nt<NT>(rFy, (!mine && live) ? row * 8u : 0xfffffff8u, y);   // a neighbour's row: its share of A p
                const double rv = mine ? cr - alpha * y : 0.;                    // pcg.h:316
                dacc += rv * rv;
                dacc2 += (fr.dinvF && mine) ? rv * ((double)fdv * rv) : 0.;      // (the diagonal of a halo row is not this rank's: may be anything)
                if (fr.cz) {                                                     // k_cheb_first on this row
                    const double v = ci * rv * fr.invTheta;
                    bufStoreF64nt<NT>(rFcz, row * 8u, v);
                    dacc2 += rv * v;
                }
                y = rv;
            }
            else {
                double az = -s; az -= 0.5 * e1 * e0;
                const double dn = cheb.c1 * (e0 - cd) + cheb.c2 * (ci * (cr - az));
                y = e0 + dn;
                dacc += cr * y;                                                  // r.z of the updated z
            }
            if (MODE == 3) bufStoreF64nt<NT>(rFr, ((int)row >= fr.ownLo && (int)row < fr.ownHi) ? row * 8u : 0xfffffff8u, y);
            else bufStoreF64nt<NT>(rOut, row * 8u, y);
        }
        __syncthreads();    // protects the LDS reuse
        if (!hasNext) break;
        chunk = nchunk; pr = npr; cur = nxt; myBase = nBase;
        nchunk = nn; npr = nnpr;
        ++it;
    }
    if (MODE == 0 || MODE == 2) {
        const double bs = blockReduceSum(dacc);
        if (threadIdx.x == 0) partial[blockIdx.x] = bs;   // gridDim.x partials (Launch::stBlocks)
    }
    if (MODE == 3) {
        const double b0 = blockReduceSum(dacc), b1 = (fr.dinvF || fr.cz) ? blockReduceSum(dacc2) : 0.;
        if (threadIdx.x == 0) { fr.rPart[blockIdx.x] = b0; fr.rPart[gridDim.x + blockIdx.x] = b1; }
    }
}


// ---- row-per-lane kernels on DevCSR::ecol / ecode (ps_blocks.hip:buildEll) ------------------------------------------------
// Why (profiles/r03_spmv_issue.md): the CU's vector-memory path walks a gather instruction one aligned quad of lanes per cycle and
// pays one L1 tag lookup per DISTINCT 128-byte line in the quad.  With 4 consecutive stream entries per lane (kernels above) the
// lanes of a quad hold entries 16 apart — four lines, four cycles: both SpMVs sat at 0.8 lookups per cycle and CU, the L1 pipeline
// saturated while 86 % of the lookups hit.  Here a lane owns a ROW: instruction k gathers entry k of 64 consecutive rows (the three
// faces of a voxel, then the next voxel), so the lanes of a quad address the same or neighbouring lines.  The row's sum stays in a
// register: no products through LDS, no row-length scan, no barrier; the four waves of a workgroup walk the chunk list together but
// never wait for each other.  Same products, same summation order as the kernels above: bit-identical results.
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
struct EllUnit { int W, colByte, codeByte, row0, rows; };   // wave-uniform
__device__ inline EllUnit ellUnit(int4 ci, int wv) {
    const unsigned ws = (unsigned)ci.w >> 12;
    int pre = 0, preC = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int w = (int)((ws >> (4 * u)) & 15u);
        if (u < wv) { pre += w; preC += w > 4 ? 8 : (w > 0 ? 4 : 0); }
    }
    EllUnit e;
    e.W = (int)((ws >> (4 * wv)) & 15u);
    e.colByte = ci.x * 2 + 128 * pre;
    e.codeByte = ci.y + 64 * preC;
    e.row0 = ci.z + 64 * wv;
    e.rows = min(64, max(0, (ci.w & 0xfff) - 64 * wv));
    if (e.rows == 0) e.W = 0;
    return e;
}
struct EllRegs { unsigned c0, c1, c2, c3, v0, v1; };   // scalars, every one written by every path: arrays written in part went to scratch memory
template <bool NT>
__device__ inline EllRegs ellLoad(__amdgpu_buffer_rsrc_t rCol, __amdgpu_buffer_rsrc_t rCode, const EllUnit& e, unsigned lane) {
    constexpr int AUX = NT ? PS_STREAM_AUX : 0;
    const int cb = e.colByte + (int)lane * 2 * e.W;
    EllRegs r{0u, 0u, 0u, 0u, 0u, 0u};
    if (e.W == 8) {
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rCol, cb, 0, AUX);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rCode, e.codeByte + (int)lane * 8, 0, AUX);
        r = EllRegs{q.x, q.y, q.z, q.w, v.x, v.y};
    } else if (e.W == 6) {
        const u32x3 q = __builtin_amdgcn_raw_buffer_load_b96(rCol, cb, 0, AUX);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rCode, e.codeByte + (int)lane * 8, 0, AUX);
        r = EllRegs{q.x, q.y, q.z, 0u, v.x, v.y};
    } else if (e.W == 4) {
        const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rCol, cb, 0, AUX);
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rCode, e.codeByte + (int)lane * 4, 0, AUX);
        r = EllRegs{q.x, q.y, 0u, 0u, v, 0u};
    } else if (e.W == 2) {
        const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(rCol, cb, 0, AUX);
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rCode, e.codeByte + (int)lane * 4, 0, AUX);
        r = EllRegs{q, 0u, 0u, 0u, v, 0u};
    }
    return r;
}
template <int K> __device__ inline unsigned ellColWord(const EllRegs& r) { return K < 2 ? r.c0 : (K < 4 ? r.c1 : (K < 6 ? r.c2 : r.c3)); }
// the row's W gathers (all in flight together), and — after the caller has issued its prefetches behind them — the products in
// entry order (CSR order)
struct EllX { double x0, x1, x2, x3, x4, x5, x6, x7; };
template <int W, class T = double>
__device__ inline EllX ellGather(const EllRegs& r, int myBase, __amdgpu_buffer_rsrc_t rX) {
    const unsigned cw[4] = {r.c0, r.c1, r.c2, r.c3};
    double xv[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const unsigned raw = (cw[k >> 1] >> (16 * (k & 1))) & 0xffffu;
        const unsigned col = (unsigned)__shfl(myBase, (int)(raw >> 12), 16) + (raw & 4095u);
        xv[k] = VecIO<T>::gather(rX, col);
    }
    return EllX{xv[0], xv[1], xv[2], xv[3], xv[4], xv[5], xv[6], xv[7]};
}
template <class T = double>
__device__ inline EllX ellGatherW(int W, const EllRegs& r, int myBase, __amdgpu_buffer_rsrc_t rX) {
    if (W == 8) return ellGather<8, T>(r, myBase, rX);
    if (W == 6) return ellGather<6, T>(r, myBase, rX);
    if (W == 4) return ellGather<4, T>(r, myBase, rX);
    if (W == 2) return ellGather<2, T>(r, myBase, rX);
    return EllX{0., 0., 0., 0., 0., 0., 0., 0.};
}
template <int W>
__device__ inline double ellSum(const EllRegs& r, const EllX& X, double scale) {
    const unsigned vw[2] = {r.v0, r.v1};
    const double xv[8] = {X.x0, X.x1, X.x2, X.x3, X.x4, X.x5, X.x6, X.x7};
    double s = 0.;
#pragma unroll
    for (int k = 0; k < W; ++k) s += streamVal(vw[k >> 2], k & 3, scale) * xv[k];
    return s;
}
__device__ inline double ellSumW(int W, const EllRegs& r, const EllX& X, double scale) {
    if (W == 8) return ellSum<8>(r, X, scale);
    if (W == 6) return ellSum<6>(r, X, scale);
    if (W == 4) return ellSum<4>(r, X, scale);
    if (W == 2) return ellSum<2>(r, X, scale);
    return 0.;
}
// MODE 0 / 1 as k_spmv_S_pipe
template <int MODE, int POL, bool LIST>
__global__ void __launch_bounds__(BS) k_spmv_S_ell(const uint16_t* __restrict__ ecol, const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes,
                                                   const int32_t* __restrict__ winBase, const int4* __restrict__ echunk, double scale,
                                                   const double* __restrict__ x, int cols, int rows, int nA, double dt, const double* __restrict__ McInv,
                                                   double* __restrict__ out, const int* __restrict__ done, int nChunks, int xcdAware,
                                                   const uint8_t* __restrict__ mcCode, const double* __restrict__ mcDict, double* __restrict__ stPart,
                                                   const int32_t* __restrict__ list) {
    // LIST: `list` holds the chunks this launch works on, nChunks of them — the slab decomposition runs the chunks that touch no
    // halo column while the halo values are still in flight, and the others afterwards (ps_dist.hpp).  A template parameter: the
    // single-domain kernels must not pay registers for it (the plain MODE 3 St kernel sits at the edge of 7 waves per SIMD)
    if (done && *done) return;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;
    __shared__ double dict[MODE == 0 ? 256 : 1];
    if (MODE == 0 && mcCode) dict[threadIdx.x] = mcDict[threadIdx.x];
    __syncthreads();                                                    // the only barrier before the final reduction
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rX = bufRsrc(x, (size_t)cols * 8),
                                 rMc = bufRsrc(McInv, (size_t)nA * 8), rMcc = bufRsrc(mcCode, mcCode ? (size_t)nA : 0), rOut = bufRsrc(out, (size_t)rows * 8);
    const ChunkWalk Wk(xcdAware);
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double stAcc = 0.;
    int it = 0;
    int chunk = Wk.at(0);                                               // POSITIONS in the walk (< nChunks); the chunk itself is list[position] when there is a list
    if (chunk < nChunks) {
        const int id0 = LIST ? list[chunk] : chunk;
        EllUnit cu = ellUnit(echunk[id0], wv);
        EllRegs cur = ellLoad<SNT>(rCol, rCode, cu, lane), nxt{0u, 0u, 0u, 0u, 0u, 0u};
        int myBase = winBase[id0 * 16 + (lane & 15)], nBase = 0;
        int nchunk = Wk.at(1), nid = 0;
        int4 nci = make_int4(0, 0, 0, 0);
        if (nchunk < nChunks) { nid = LIST ? list[nchunk] : nchunk; nci = echunk[nid]; }
        while (true) {
            // (1) this unit's gathers first: their address arithmetic waits on nothing but the window lookups
            const bool live = (int)lane < cu.rows;
            const unsigned row = live ? (unsigned)cu.row0 + lane : ROW_NONE;
            double sc = 1.;
            int mcc = 0;
            if (MODE == 0) {
                if (mcCode) mcc = (int)__builtin_amdgcn_raw_buffer_load_b8(rMcc, (int)row, 0, NT ? PS_EPI_AUX : 0);
                else { const double m = bufLoadF64(rMc, row * 8u); sc = (int)row < nA ? dt * m : 1.; }
            }
            const EllX X = ellGatherW(cu.W, cur, myBase, rX);
            // (2) behind them: the next unit's stream and the record of the chunk after it
            const bool hasNext = nchunk < nChunks;
            EllUnit nu{0, 0, 0, 0, 0};
            if (hasNext) {
                nu = ellUnit(nci, wv);
                nxt = ellLoad<SNT>(rCol, rCode, nu, lane);
                nBase = winBase[nid * 16 + (lane & 15)];
            }
            const int nn = Wk.at(it + 2);
            int nnid = 0;
            int4 nnci = make_int4(0, 0, 0, 0);
            if (nn < nChunks) { nnid = LIST ? list[nn] : nn; nnci = echunk[nnid]; }
            // (3) products, epilogue
            const double s = ellSumW(cu.W, cur, X, scale);
            if (MODE == 0 && mcCode) sc = (int)row < nA ? dt * dict[mcc] : 1.;
            if (MODE == 0) stAcc += (int)row < nA ? s * (s * sc) : 0.;
            bufStoreF64nt<NT>(rOut, row * 8u, s * sc);                       // dropped past the chunk's last row (row = ROW_NONE)
            if (!hasNext) break;
            chunk = nchunk; cu = nu; cur = nxt; myBase = nBase;
            nchunk = nn; nci = nnci; nid = nnid;
            ++it;
        }
    }
    if (MODE == 0 && stPart) {
        const double bs = blockReduceSum(stAcc);
        if (threadIdx.x == 0) stPart[blockIdx.x] = bs;
    }
}
// MODE 0 .. 3 as k_spmv_St_pipe (same prologue, same per-row epilogue, same thread <-> row assignment: bit-identical partial sums)
// FX (MODE 3 only): facts about the launch as compile-time constants, each worth buffer descriptors and branches (the generic MODE 3
// needs 100 SGPRs, spills them into VGPR lanes and fits 6 waves per SIMD).  Bit 0: value-set coded uInv and no Chebyshev first term
// (the Jacobi / identity PCG step); bit 1: a single domain (no halo rows, no all-reduced sums, one launch); bit 2: a rank's launch over
// chunks of owned rows only (ps_dist.hpp: the St launch that runs under the exchange — most of a rank's rows).
template <int MODE, int POL, int FX, bool LIST>
__global__ void __launch_bounds__(BS) k_spmv_St_ell(const uint16_t* __restrict__ ecol, const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes,
                                                    const int32_t* __restrict__ winBase, const int4* __restrict__ echunk, double scale,
                                                    const double* __restrict__ t, int cols, int rows, const double* __restrict__ uInv,
                                                    const double* __restrict__ xin, const double* __restrict__ add, double* __restrict__ out,
                                                    double* __restrict__ partial, const int* __restrict__ done, int nChunks, int xcdAware, ChebArgs cheb,
                                                    const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, FusedR fr, const int32_t* __restrict__ list) {
    // list: as in k_spmv_S_ell
    if (done && *done) return;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;
    static_assert(FX == 0 || MODE == 3, "FX: MODE 3 only");
    if (FX & 1) { fr.cz = nullptr; fr.dinvC = nullptr; uInv = nullptr; }                                              // (the launch site guarantees uCode != null)
    if (FX & 2) { fr.yOut = nullptr; fr.rStride = 0; fr.red = nullptr; fr.ownLo = 0; fr.ownHi = rows; }
    if (FX & 4) { fr.yOut = nullptr; fr.ownLo = 0; fr.ownHi = rows; }   // a rank of a decomposition, chunks of OWNED rows only (the launch under the exchange): no halo row to hand on
    const bool coded = (FX & 1) ? true : uCode != nullptr;
    __shared__ double dict[MODE != 1 ? 256 : 1];
    if (MODE != 1 && coded) dict[threadIdx.x] = uDict[threadIdx.x];
    double alpha = 0.;
    if (MODE == 3) {   // as k_spmv_St_pipe: [stop test of iteration it-1], alpha = rsold / p.Ap — identical in every workgroup
        CGScalars* sc = fr.sc;
        auto sumArr = [&](const double* a, int cnt) { double acc = 0.; for (int i = threadIdx.x; i < cnt; i += BS) acc += a[i]; return blockSumAll(acc); };
        const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
        if (fr.it > 0) {
            const double xx = fr.red ? fr.red[1] : sumArr(fr.xxPart, fr.xxCount);
            const double rr = sc->rr;
            double rre = rr;                               // pcg.h:319-325
            if (rr / xx < rre) rre = rr / xx;
            const bool fire = rre < sc->tol2;
            if (writer) { sc->xx = xx; sc->rre = rre; if (fire) { sc->done = 1; sc->iter = fr.it - 1; } }
            if (fire) return;                              // same verdict in every workgroup
        }
        const double pAp = fr.red ? -fr.red[0] : -(sumArr(fr.sPart, fr.sCount) + sumArr(fr.tPart, fr.tCount) + 0.5 * sumArr(fr.uPart, fr.uCount));
        alpha = sc->rsold2[fr.it & 1] / pAp;               // pcg.h:314
        if (writer) { sc->pAp = pAp; sc->alpha = alpha; }
    }
    __syncthreads();                                       // dict
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rT = bufRsrc(t, (size_t)cols * 8),
                                 rE0 = bufRsrc(MODE == 1 ? add : xin, (size_t)rows * 8), rE1 = bufRsrc(uInv, (size_t)rows * 8),
                                 rOut = bufRsrc(out, (size_t)rows * 8),
                                 rCr = bufRsrc(cheb.r, MODE == 2 ? (size_t)rows * 8 : 0), rCi = bufRsrc(cheb.dinv, MODE == 2 ? (size_t)rows * sizeof(diag_t) : 0),
                                 rCd = bufRsrc(cheb.zprev, (MODE == 2 && cheb.zprev) ? (size_t)rows * 8 : 0), rUc = bufRsrc(uCode, coded ? (size_t)rows : 0),
                                 rFr = bufRsrc(fr.r, MODE == 3 ? (size_t)rows * 8 : 0), rFd = bufRsrc(fr.dinvF, (MODE == 3 && fr.dinvF) ? (size_t)rows * sizeof(diag_t) : 0),
                                 rF64 = bufRsrc(fr.dinvC, (MODE == 3 && fr.cz) ? (size_t)rows * sizeof(diag_t) : 0), rFcz = bufRsrc(fr.cz, (MODE == 3 && fr.cz) ? (size_t)rows * 8 : 0),
                                 rFy = bufRsrc(fr.yOut, (MODE == 3 && fr.yOut) ? (size_t)rows * 8 : 0);
    const ChunkWalk Wk(xcdAware);
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double dacc = 0., dacc2 = 0.;
    int it = 0;
    int chunk = Wk.at(0);                                               // POSITIONS in the walk (< nChunks); the chunk itself is list[position] when there is a list
    if (chunk < nChunks) {
        const int id0 = LIST ? list[chunk] : chunk;
        EllUnit cu = ellUnit(echunk[id0], wv);
        EllRegs cur = ellLoad<SNT>(rCol, rCode, cu, lane), nxt{0u, 0u, 0u, 0u, 0u, 0u};
        int myBase = winBase[id0 * 16 + (lane & 15)], nBase = 0;
        int nchunk = Wk.at(1), nid = 0;
        int4 nci = make_int4(0, 0, 0, 0);
        if (nchunk < nChunks) { nid = LIST ? list[nchunk] : nchunk; nci = echunk[nid]; }
        while (true) {
            const bool live = (int)lane < cu.rows;
            const unsigned row = live ? (unsigned)cu.row0 + lane : ROW_NONE;
            // (1) the per-row streams of the epilogue and this unit's gathers
            const double e0 = bufLoadF64epi<NT>(rE0, row * 8u);                                       // x (MODE 0, 2, 3) / the vector added (MODE 1)
            double e1 = 0., cr = 0., ci = 0., cd = 0.;
            int uc = 0;
            if (MODE != 1) { if (coded) uc = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)row, 0, NT ? PS_EPI_AUX : 0); else e1 = bufLoadF64epi<NT>(rE1, row * 8u); }
            if (MODE == 2) { cr = bufLoadF64(rCr, row * 8u); ci = (double)bufLoadDiag<false>(rCi, row); cd = bufLoadF64(rCd, row * 8u); }   // cd = z_{j-1} (0: no buffer)
            float fdv = 1.f;
            if (MODE == 3) {
                cr = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rFr, (int)(row * 8u), 0, NT ? PS_EPI_AUX : 0));
                if (fr.dinvF) fdv = bufLoadDiag<NT>(rFd, row);
                if (fr.cz) ci = (double)bufLoadDiag<NT>(rF64, row);
            }
            const EllX X = ellGatherW(cu.W, cur, myBase, rT);
            // (2) behind them: the next unit's stream and the record of the chunk after it
            const bool hasNext = nchunk < nChunks;
            EllUnit nu{0, 0, 0, 0, 0};
            if (hasNext) {
                nu = ellUnit(nci, wv);
                nxt = ellLoad<SNT>(rCol, rCode, nu, lane);
                nBase = winBase[nid * 16 + (lane & 15)];
            }
            const int nn = Wk.at(it + 2);
            int nnid = 0;
            int4 nnci = make_int4(0, 0, 0, 0);
            if (nn < nChunks) { nnid = LIST ? list[nn] : nn; nnci = echunk[nnid]; }
            // (3) the row's sum and the fused epilogue
            const double s = ellSumW(cu.W, cur, X, scale);
            if (MODE != 1 && coded) e1 = dict[uc];
            double y;
            if (MODE == 0) { y = -s; y -= 0.5 * e1 * e0; dacc += e0 * y; }   // p.Ap: running sum over this block's chunks (0 past the last row)
            else if (MODE == 1) y = -s + e0;
            else if (MODE == 3) {
                y = -s; y -= 0.5 * e1 * e0;                                      // (A p)[row], not stored
                const bool mine = (int)row >= fr.ownLo && (int)row < fr.ownHi;   // (idle lanes: false)
                if (fr.yOut) bufStoreF64nt<NT>(rFy, (!mine && live) ? row * 8u : 0xfffffff8u, y);   // a neighbour's row: its share of A p
                const double rv = mine ? cr - alpha * y : 0.;                    // pcg.h:316
                dacc += rv * rv;
                dacc2 += (fr.dinvF && mine) ? rv * ((double)fdv * rv) : 0.;      // (the diagonal of a halo row is not this rank's: may be anything)
                if (fr.cz) {                                                     // k_cheb_first on this row
                    const double v = ci * rv * fr.invTheta;
                    bufStoreF64nt<NT>(rFcz, row * 8u, v);
                    dacc2 += rv * v;
                }
                y = rv;
            }
            else {
                double az = -s; az -= 0.5 * e1 * e0;
                const double dn = cheb.c1 * (e0 - cd) + cheb.c2 * (ci * (cr - az));
                y = e0 + dn;
                dacc += cr * y;                                                  // r.z of the updated z
            }
            if (MODE == 3) bufStoreF64nt<NT>(rFr, ((int)row >= fr.ownLo && (int)row < fr.ownHi) ? row * 8u : 0xfffffff8u, y);
            else bufStoreF64nt<NT>(rOut, row * 8u, y);
            if (!hasNext) break;
            chunk = nchunk; cu = nu; cur = nxt; myBase = nBase;
            nchunk = nn; nci = nnci; nid = nnid;
            ++it;
        }
    }
    if (MODE == 0 || MODE == 2) {
        const double bs = blockReduceSum(dacc);
        if (threadIdx.x == 0) partial[blockIdx.x] = bs;   // gridDim.x partials (Launch::stBlocks)
    }
    if (MODE == 3) {
        const double b0 = blockReduceSum(dacc), b1 = (fr.dinvF || fr.cz) ? blockReduceSum(dacc2) : 0.;
        if (threadIdx.x == 0) { fr.rPart[blockIdx.x] = b0; fr.rPart[(fr.rStride > 0 ? fr.rStride : (int)gridDim.x) + blockIdx.x] = b1; }
    }
}

// ---- two units in flight per wave (r04; default for the single-domain operator product, PS_S_DUAL=0 switches back) -------------------
// k_spmv_S_ell with twice the memory-level parallelism per wave: a workgroup takes TWO chunks per step, waves 0-1 the first, waves 2-3 the
// second; a wave owns units 2 (w & 1) and 2 (w & 1) + 1 of its chunk and has the streams, the gathers and the epilogue loads of both in
// flight before it sums either.  Same products, same order per row: bit-identical t; the per-workgroup partials of sum s.t group differently.
// TV: element type of x and of the output (double; float = the inner applies of the single-precision Chebyshev polynomial, VecIO)
template <int POL, bool LIST, class TV = double>
__global__ void __launch_bounds__(BS) k_spmv_S_ell2(const uint16_t* __restrict__ ecol, const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes,
                                                    const int32_t* __restrict__ winBase, const int4* __restrict__ echunk, double scale,
                                                    const TV* __restrict__ x, int cols, int rows, int nA, double dt, TV* __restrict__ out,
                                                    const int* __restrict__ done, int nChunks, const uint8_t* __restrict__ mcCode, const double* __restrict__ mcDict,
                                                    double* __restrict__ stPart, const int32_t* __restrict__ list) {   // LIST: as k_spmv_S_ell (nChunks = entries of the list)
    if (done && *done) return;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;
    __shared__ double dict[256];
    dict[threadIdx.x] = mcDict[threadIdx.x];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rX = bufRsrc(x, (size_t)cols * sizeof(TV)),
                                 rMcc = bufRsrc(mcCode, (size_t)nA), rOut = bufRsrc(out, (size_t)rows * sizeof(TV));
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wv >> 1, u0 = 2 * (wv & 1);
    double stAcc = 0.;
    // pairs of consecutive chunks, runs of 32 pairs dealt to the XCDs round robin (workgroup b runs on XCD b & 7)
    const int nPairs = (nChunks + 1) >> 1;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, per = gridDim.x >> 3;
    // step q of this workgroup -> its chunk (or -1): the pair, then this wave's half of it
    auto chunkAt = [&](int q) -> int {
        const int pair = ((((q >> 5) << 3) + xcd) << 5) + (q & 31);
        const int ch = 2 * pair + half;
        return (pair < nPairs && ch < nChunks) ? (LIST ? list[ch] : ch) : -1;
    };
    const int qEnd = ((nPairs + 255) >> 8) << 5;          // steps beyond the last run of 32 pairs per XCD
    int q = l;
    int chunk = q < qEnd ? chunkAt(q) : -1;
    int4 ci = make_int4(0, 0, 0, 0);
    int myBase = 0;
    if (chunk >= 0) { ci = echunk[chunk]; myBase = winBase[chunk * 16 + (lane & 15)]; }
    while (q < qEnd) {
        const int qn = q + per;
        const int nchunk = qn < qEnd ? chunkAt(qn) : -1;
        int4 nci = make_int4(0, 0, 0, 0);
        int nBase = 0;
        if (chunk >= 0) {
            const EllUnit ua = ellUnit(ci, u0), ub = ellUnit(ci, u0 + 1);
            const EllRegs sa = ellLoad<SNT>(rCol, rCode, ua, lane), sb = ellLoad<SNT>(rCol, rCode, ub, lane);
            if (nchunk >= 0) { nci = echunk[nchunk]; nBase = winBase[nchunk * 16 + (lane & 15)]; }     // the next step's record, behind this step's streams
            const unsigned rowA = (int)lane < ua.rows ? (unsigned)ua.row0 + lane : ROW_NONE, rowB = (int)lane < ub.rows ? (unsigned)ub.row0 + lane : ROW_NONE;
            const int mA = (int)__builtin_amdgcn_raw_buffer_load_b8(rMcc, (int)rowA, 0, NT ? PS_EPI_AUX : 0);
            const int mB = (int)__builtin_amdgcn_raw_buffer_load_b8(rMcc, (int)rowB, 0, NT ? PS_EPI_AUX : 0);
            const EllX XA = ellGatherW<TV>(ua.W, sa, myBase, rX);
            const EllX XB = ellGatherW<TV>(ub.W, sb, myBase, rX);
            const double a = ellSumW(ua.W, sa, XA, scale), b = ellSumW(ub.W, sb, XB, scale);
            const double scA = (int)rowA < nA ? dt * dict[mA] : 1., scB = (int)rowB < nA ? dt * dict[mB] : 1.;
            stAcc += (int)rowA < nA ? a * (a * scA) : 0.;
            stAcc += (int)rowB < nA ? b * (b * scB) : 0.;
            VecIO<TV>::template store<NT>(rOut, rowA, a * scA);
            VecIO<TV>::template store<NT>(rOut, rowB, b * scB);
        } else if (nchunk >= 0) { nci = echunk[nchunk]; nBase = winBase[nchunk * 16 + (lane & 15)]; }
        q = qn; chunk = nchunk; ci = nci; myBase = nBase;
    }
    if (stPart) {
        const double bs = blockReduceSum(stAcc);
        if (threadIdx.x == 0) stPart[blockIdx.x] = bs;
    }
}

// The same for the St kernel of the plain single-domain step (k_spmv_St_ell<3, POL, 3>: coded uInv, residual update in the epilogue): two
// units in flight per wave, two chunks per workgroup and step (r04; default, PS_ST_DUAL=0 switches back).  Compiled for six waves per SIMD
// (80 VGPRs): twelve units in flight per SIMD against the seven of the one-unit kernel.
// CZ: the Chebyshev polynomial's first term on the new r in the epilogue (z_1 = dinv r / theta -> fr.cz, r.z partials from it; fr.dinvF unused)
// DIST: a rank of a decomposition — alpha and ||x||^2 from the all-reduced sums (fr.red), partials at fr.rStride; with LIST the launch over
// the chunks of OWNED rows only that runs under the exchange (ps_dist.hpp; k_spmv_St_ell's FX bit 2)
// TZ: element type of fr.cz (CZ only; float: the single-precision Chebyshev polynomial — fr.cz then points at floats and r.z is formed
// with the value as stored)
// HALO (DIST only, r06): the launch may hold halo rows — rows outside [fr.ownLo, fr.ownHi) whose y is this rank's share of a neighbour's A p: it goes
// to fr.yOut and r is left alone there (k_spmv_St_ell's generic MODE 3 did this on one unit per wave; the rank's launch over the chunks next
// to a cut and the whole-rank launch of the sequential exchange now run two units per wave like the owned-rows launch)
// UC (r06): the stress diagonal uInv comes as one-byte codes into a 256-entry table (true) or, when it takes more than 256 values — a viscosity FIELD —, as the
// fp64 array itself, passed in uCode's place (false: 7 more bytes per row; until r06 such scenes ran the one-unit kernels and the fp64 polynomial)
template <int POL, bool CZ, bool DIST, bool LIST, class TZ = double, bool HALO = false, bool UC = true>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(6, 6))) k_spmv_St_ell2(const uint16_t* __restrict__ ecol, const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes,
                                                     const int32_t* __restrict__ winBase, const int4* __restrict__ echunk, double scale,
                                                     const double* __restrict__ t, int cols, int rows, const double* __restrict__ xin,
                                                     const int* __restrict__ done, int nChunks, const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, FusedR fr,
                                                     const int32_t* __restrict__ list) {
    if (done && *done) return;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;
    __shared__ double dict[UC ? 256 : 1];
    if (UC) dict[threadIdx.x] = uDict[threadIdx.x];
    double alpha;
    {   // as k_spmv_St_ell MODE 3
        CGScalars* sc = fr.sc;
        auto sumArr = [&](const double* a, int cnt) { double acc = 0.; for (int i = threadIdx.x; i < cnt; i += BS) acc += a[i]; return blockSumAll(acc); };
        const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
        if (fr.it > 0) {
            const double xx = DIST ? fr.red[1] : sumArr(fr.xxPart, fr.xxCount);
            const double rr = sc->rr;
            double rre = rr;                               // pcg.h:319-325
            if (rr / xx < rre) rre = rr / xx;
            const bool fire = rre < sc->tol2;
            if (writer) { sc->xx = xx; sc->rre = rre; if (fire) { sc->done = 1; sc->iter = fr.it - 1; } }
            if (fire) return;
        }
        const double pAp = DIST ? -fr.red[0] : -(sumArr(fr.sPart, fr.sCount) + sumArr(fr.tPart, fr.tCount) + 0.5 * sumArr(fr.uPart, fr.uCount));
        alpha = sc->rsold2[fr.it & 1] / pAp;               // pcg.h:314
        if (writer) { sc->pAp = pAp; sc->alpha = alpha; }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rT = bufRsrc(t, (size_t)cols * 8),
                                 rE0 = bufRsrc(xin, (size_t)rows * 8), rUc = bufRsrc(uCode, UC ? (size_t)rows : (size_t)rows * 8),
                                 rFr = bufRsrc(fr.r, (size_t)rows * 8), rFd = bufRsrc(fr.dinvF, (!CZ && fr.dinvF) ? (size_t)rows * sizeof(diag_t) : 0),
                                 rF64 = bufRsrc(fr.dinvC, CZ ? (size_t)rows * sizeof(diag_t) : 0), rFcz = bufRsrc(fr.cz, CZ ? (size_t)rows * sizeof(TZ) : 0),
                                 rFy = bufRsrc(fr.yOut, HALO ? (size_t)rows * 8 : 0);
    static_assert(!HALO || DIST, "halo rows exist on a rank of a decomposition only");
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wv >> 1, u0 = 2 * (wv & 1);
    double dacc = 0., dacc2 = 0.;
    const int nPairs = (nChunks + 1) >> 1;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, per = gridDim.x >> 3;
    auto chunkAt = [&](int q) -> int {
        const int pair = ((((q >> 5) << 3) + xcd) << 5) + (q & 31);
        const int ch = 2 * pair + half;
        return (pair < nPairs && ch < nChunks) ? (LIST ? list[ch] : ch) : -1;
    };
    const int qEnd = ((nPairs + 255) >> 8) << 5;
    int q = l;
    int chunk = q < qEnd ? chunkAt(q) : -1;
    int4 ci = make_int4(0, 0, 0, 0);
    int myBase = 0;
    if (chunk >= 0) { ci = echunk[chunk]; myBase = winBase[chunk * 16 + (lane & 15)]; }
    while (q < qEnd) {
        const int qn = q + per;
        const int nchunk = qn < qEnd ? chunkAt(qn) : -1;
        int4 nci = make_int4(0, 0, 0, 0);
        int nBase = 0;
        if (chunk >= 0) {
            const EllUnit ua = ellUnit(ci, u0), ub = ellUnit(ci, u0 + 1);
            const EllRegs sa = ellLoad<SNT>(rCol, rCode, ua, lane), sb = ellLoad<SNT>(rCol, rCode, ub, lane);
            if (nchunk >= 0) { nci = echunk[nchunk]; nBase = winBase[nchunk * 16 + (lane & 15)]; }
            const bool liveA = (int)lane < ua.rows, liveB = (int)lane < ub.rows;
            const unsigned rowA = liveA ? (unsigned)ua.row0 + lane : ROW_NONE, rowB = liveB ? (unsigned)ub.row0 + lane : ROW_NONE;
            const double eA = bufLoadF64epi<NT>(rE0, rowA * 8u), eB = bufLoadF64epi<NT>(rE0, rowB * 8u);
            int ucA = 0, ucB = 0;
            double uvA = 0., uvB = 0.;
            if (UC) {
                ucA = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)rowA, 0, NT ? PS_UC_AUX : 0);
                ucB = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)rowB, 0, NT ? PS_UC_AUX : 0);
            } else { uvA = bufLoadF64epi<NT>(rUc, rowA * 8u); uvB = bufLoadF64epi<NT>(rUc, rowB * 8u); }
            const double crA = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rFr, (int)(rowA * 8u), 0, NT ? PS_EPI_AUX : 0));
            const double crB = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rFr, (int)(rowB * 8u), 0, NT ? PS_EPI_AUX : 0));
            float fdA = 1.f, fdB = 1.f;
            double ciA = 0., ciB = 0.;
            if (CZ) { ciA = (double)bufLoadDiag<NT>(rF64, rowA); ciB = (double)bufLoadDiag<NT>(rF64, rowB); }
            else if (fr.dinvF) {
                fdA = bufLoadDiag<NT>(rFd, rowA);
                fdB = bufLoadDiag<NT>(rFd, rowB);
            }
            const EllX XA = ellGatherW(ua.W, sa, myBase, rT);
            const EllX XB = ellGatherW(ub.W, sb, myBase, rT);
            const double a = ellSumW(ua.W, sa, XA, scale), b = ellSumW(ub.W, sb, XB, scale);
            double yA = -a; yA -= 0.5 * (UC ? dict[ucA] : uvA) * eA;
            double yB = -b; yB -= 0.5 * (UC ? dict[ucB] : uvB) * eB;
            const bool mineA = HALO ? ((int)rowA >= fr.ownLo && (int)rowA < fr.ownHi) : liveA, mineB = HALO ? ((int)rowB >= fr.ownLo && (int)rowB < fr.ownHi) : liveB;   // (idle lanes: ROW_NONE is beyond ownHi)
            if (HALO) {                                                      // a neighbour's row: its share of A p
                bufStoreF64nt<NT>(rFy, (!mineA && liveA) ? rowA * 8u : 0xfffffff8u, yA);
                bufStoreF64nt<NT>(rFy, (!mineB && liveB) ? rowB * 8u : 0xfffffff8u, yB);
            }
            const double rvA = mineA ? crA - alpha * yA : 0., rvB = mineB ? crB - alpha * yB : 0.;   // pcg.h:316
            dacc += rvA * rvA; dacc += rvB * rvB;
            if (CZ) {                                                        // k_cheb_first on these rows
                const double vA = VecIO<TZ>::stored(ciA * rvA * fr.invTheta), vB = VecIO<TZ>::stored(ciB * rvB * fr.invTheta);
                VecIO<TZ>::template store<NT>(rFcz, rowA, vA); VecIO<TZ>::template store<NT>(rFcz, rowB, vB);
                dacc2 += rvA * vA; dacc2 += rvB * vB;
            } else if (fr.dinvF) { dacc2 += rvA * ((double)fdA * rvA); dacc2 += rvB * ((double)fdB * rvB); }
            bufStoreF64nt<NT>(rFr, (!HALO || mineA) ? rowA * 8u : 0xfffffff8u, rvA);
            bufStoreF64nt<NT>(rFr, (!HALO || mineB) ? rowB * 8u : 0xfffffff8u, rvB);
        } else if (nchunk >= 0) { nci = echunk[nchunk]; nBase = winBase[nchunk * 16 + (lane & 15)]; }
        q = qn; chunk = nchunk; ci = nci; myBase = nBase;
    }
    const double b0 = blockReduceSum(dacc), b1 = (CZ || fr.dinvF) ? blockReduceSum(dacc2) : 0.;
    if (threadIdx.x == 0) { fr.rPart[blockIdx.x] = b0; fr.rPart[((DIST && fr.rStride > 0) ? fr.rStride : (int)gridDim.x) + blockIdx.x] = b1; }
}

// ... and for MODE 2 (one term of the Chebyshev preconditioner in the epilogue; coded uInv): two units in flight per wave.
// TV: element type of t, xin (z_j), cheb.zprev (z_{j-1}) and out (z_{j+1}); r and every sum stay fp64
#ifdef PS_ST2C_WAVES      // A/B build knob (scripts/build_variant.sh): waves per SIMD the Chebyshev-term kernel is compiled for
#define PS_ST2C_ATTR __attribute__((amdgpu_waves_per_eu(PS_ST2C_WAVES, PS_ST2C_WAVES)))
#else
#define PS_ST2C_ATTR
#endif
template <int POL, class TV = double, bool UC = true>
__global__ void __launch_bounds__(BS) PS_ST2C_ATTR k_spmv_St_ell2c(const uint16_t* __restrict__ ecol, const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes,
                                                      const int32_t* __restrict__ winBase, const int4* __restrict__ echunk, double scale,
                                                      const TV* __restrict__ t, int cols, int rows, const TV* __restrict__ xin, TV* __restrict__ out,
                                                      double* __restrict__ partial, const int* __restrict__ done, int nChunks, ChebArgs cheb,
                                                      const uint8_t* __restrict__ uCode, const double* __restrict__ uDict) {
    if (done && *done) return;
    constexpr bool NT = (POL & 1) != 0, SNT = (POL & 2) != 0;
    __shared__ double dict[UC ? 256 : 1];
    if (UC) dict[threadIdx.x] = uDict[threadIdx.x];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rT = bufRsrc(t, (size_t)cols * sizeof(TV)),
                                 rE0 = bufRsrc(xin, (size_t)rows * sizeof(TV)), rUc = bufRsrc(uCode, UC ? (size_t)rows : (size_t)rows * 8), rOut = bufRsrc(out, (size_t)rows * sizeof(TV)),
                                 rCr = bufRsrc(cheb.r, (size_t)rows * 8), rCi = bufRsrc(cheb.dinv, (size_t)rows * sizeof(diag_t)),
                                 rCd = bufRsrc(cheb.zprev, cheb.zprev ? (size_t)rows * sizeof(TV) : 0);   // (cheb.zprev points at TV elements)
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wv >> 1, u0 = 2 * (wv & 1);
    double dacc = 0.;
    const int nPairs = (nChunks + 1) >> 1;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, per = gridDim.x >> 3;
    auto chunkAt = [&](int q) -> int {
        const int pair = ((((q >> 5) << 3) + xcd) << 5) + (q & 31);
        const int ch = 2 * pair + half;
        return (pair < nPairs && ch < nChunks) ? ch : -1;
    };
    const int qEnd = ((nPairs + 255) >> 8) << 5;
    int q = l;
    int chunk = q < qEnd ? chunkAt(q) : -1;
    int4 ci = make_int4(0, 0, 0, 0);
    int myBase = 0;
    if (chunk >= 0) { ci = echunk[chunk]; myBase = winBase[chunk * 16 + (lane & 15)]; }
    while (q < qEnd) {
        const int qn = q + per;
        const int nchunk = qn < qEnd ? chunkAt(qn) : -1;
        int4 nci = make_int4(0, 0, 0, 0);
        int nBase = 0;
        if (chunk >= 0) {
            const EllUnit ua = ellUnit(ci, u0), ub = ellUnit(ci, u0 + 1);
            const EllRegs sa = ellLoad<SNT>(rCol, rCode, ua, lane), sb = ellLoad<SNT>(rCol, rCode, ub, lane);
            if (nchunk >= 0) { nci = echunk[nchunk]; nBase = winBase[nchunk * 16 + (lane & 15)]; }
            const unsigned rowA = (int)lane < ua.rows ? (unsigned)ua.row0 + lane : ROW_NONE, rowB = (int)lane < ub.rows ? (unsigned)ub.row0 + lane : ROW_NONE;
            const double eA = VecIO<TV>::template loadEpi<NT>(rE0, rowA), eB = VecIO<TV>::template loadEpi<NT>(rE0, rowB);
            int ucA = 0, ucB = 0;
            double uvA = 0., uvB = 0.;
            if (UC) {
                ucA = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)rowA, 0, NT ? PS_UC_AUX : 0);
                ucB = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)rowB, 0, NT ? PS_UC_AUX : 0);
            } else { uvA = bufLoadF64epi<NT>(rUc, rowA * 8u); uvB = bufLoadF64epi<NT>(rUc, rowB * 8u); }
            const double crA = bufLoadF64(rCr, rowA * 8u), crB = bufLoadF64(rCr, rowB * 8u);
            const double ciA = (double)bufLoadDiag<false>(rCi, rowA), ciB = (double)bufLoadDiag<false>(rCi, rowB);
            const double cdA = VecIO<TV>::load(rCd, rowA), cdB = VecIO<TV>::load(rCd, rowB);     // z_{j-1} (0: no buffer)
            const EllX XA = ellGatherW<TV>(ua.W, sa, myBase, rT);
            const EllX XB = ellGatherW<TV>(ub.W, sb, myBase, rT);
            const double a = ellSumW(ua.W, sa, XA, scale), b = ellSumW(ub.W, sb, XB, scale);
            double azA = -a; azA -= 0.5 * (UC ? dict[ucA] : uvA) * eA;
            double azB = -b; azB -= 0.5 * (UC ? dict[ucB] : uvB) * eB;
            const double yA = VecIO<TV>::stored(eA + (cheb.c1 * (eA - cdA) + cheb.c2 * (ciA * (crA - azA))));
            const double yB = VecIO<TV>::stored(eB + (cheb.c1 * (eB - cdB) + cheb.c2 * (ciB * (crB - azB))));
            dacc += crA * yA; dacc += crB * yB;                          // r.z of the updated z AS STORED (0 past the last row: every load returned 0)
            VecIO<TV>::template store<NT>(rOut, rowA, yA);
            VecIO<TV>::template store<NT>(rOut, rowB, yB);
        } else if (nchunk >= 0) { nci = echunk[nchunk]; nBase = winBase[nchunk * 16 + (lane & 15)]; }
        q = qn; chunk = nchunk; ci = nci; myBase = nBase;
    }
    const double bs = blockReduceSum(dacc);
    if (threadIdx.x == 0) partial[blockIdx.x] = bs;
}
