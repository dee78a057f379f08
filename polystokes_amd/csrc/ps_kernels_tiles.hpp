// Per-tile reduced apply: J^T (gather), 26x26 BInv (solve), J (expand).
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once

// ---- per-tile reduced apply -------------------------------------------------------------------------
// off (everywhere a face position is formed): global index of the local cell (0, 0, 0) — a rank of a decomposition forms offsets with
// global indices, the same arithmetic as the single domain, so that a tile's matrices do not depend on the decomposition (0 without one)
__device__ inline void rowOffset(uint32_t packed, const double* __restrict__ COM, int region, double dx, int3 off, double* o, int* axis) {
    int i, j, k, a;
    unpackFace(packed, i, j, k, a);
    double p[3] = {(double)(i + off.x), (double)(j + off.y), (double)(k + off.z)};
    p[a] -= 0.5;
    o[0] = p[0] * dx - COM[(int64_t)region * 3 + 0];
    o[1] = p[1] * dx - COM[(int64_t)region * 3 + 1];
    o[2] = p[2] * dx - COM[(int64_t)region * 3 + 2];
    *axis = a;
}
// ---- the basis row through moments ----------------------------------------------------------------------------------
// Every entry of the 26-entry basis row C_a(o) (buildConversionCoefficients, Solver.cpp:2112-2145) is a constant times one
// of the ten monomials  mu = (1, ox, oy, oz, ox^2, ox oy, ox oz, oy^2, oy oz, oz^2)  of the face offset o.  So
//   w = sum_f C_f s_f   needs only the 3 x 10 moments  M_a[m] = sum over faces of axis a of mu_m(o_f) s_f   (30 sums per row
//   stream instead of 26 axis-dependent ones), mapped to the 26 entries once per region (momentsToW), and
//   t_f = C_f . v      is the 10-term dot of mu(o_f) with the per-axis coefficient vector V_a of v (vToAxisCoeffs).
// Rows of the three axes are interleaved (region, position, axis): the axis is a per-lane select, no divergence.
__device__ inline void faceMonomials(uint32_t f, double dx, int3 off, double cx, double cy, double cz, double* mu, int* axis) {
    int i, j, k, a;
    unpackFace(f, i, j, k, a);
    const double ox = ((double)(i + off.x) - (a == 0 ? 0.5 : 0.)) * dx - cx;
    const double oy = ((double)(j + off.y) - (a == 1 ? 0.5 : 0.)) * dx - cy;
    const double oz = ((double)(k + off.z) - (a == 2 ? 0.5 : 0.)) * dx - cz;
    mu[0] = 1.; mu[1] = ox; mu[2] = oy; mu[3] = oz; mu[4] = ox * ox; mu[5] = ox * oy; mu[6] = ox * oz; mu[7] = oy * oy; mu[8] = oy * oz; mu[9] = oz * oz;
    *axis = a;
}
#ifdef PS_AFFINE_REGIONS
// AFFINE_REGIONS (11 DOF): only the monomials 1, ox, oy, oz enter
__device__ inline double momentsToW(const double* M, int e) {
    const double* X = M; const double* Y = M + 10; const double* Z = M + 20;
    switch (e) {
        case 0: return X[0];
        case 1: return Y[0];
        case 2: return Z[0];
        case 3: return X[1] - Z[3];
        case 4: return X[2];
        case 5: return X[3];
        case 6: return Y[1];
        case 7: return Y[2] - Z[3];
        case 8: return Y[3];
        case 9: return Z[1];
        default: return Z[2];
    }
}
__device__ inline double vToAxisCoeff(const double* v, int q) {
    switch (q) {
        case 0: return v[0];  case 1: return v[3];  case 2: return v[4];  case 3: return v[5];
        case 10: return v[1]; case 11: return v[6]; case 12: return v[7]; case 13: return v[8];
        case 20: return v[2]; case 21: return v[9]; case 22: return v[10]; case 23: return -v[3] - v[7];
        default: return 0.;
    }
}
#else
// entry e of w from the 30 moments M[a * 10 + m]
__device__ inline double momentsToW(const double* M, int e) {
    const double* X = M; const double* Y = M + 10; const double* Z = M + 20;
    switch (e) {
        case 0: return X[0];
        case 1: return Y[0];
        case 2: return Z[0];
        case 3: return X[1] - Z[3];
        case 4: return X[2];
        case 5: return X[3];
        case 6: return X[4] - 2. * Z[6];
        case 7: return X[5] - Z[8];
        case 8: return X[6] - 0.5 * Z[9];
        case 9: return X[7];
        case 10: return X[8];
        case 11: return X[9];
        case 12: return Y[1];
        case 13: return Y[2] - Z[3];
        case 14: return Y[3];
        case 15: return Y[4];
        case 16: return Y[5] - Z[6];
        case 17: return Y[6];
        case 18: return Y[7] - 2. * Z[8];
        case 19: return Y[8] - 0.5 * Z[9];
        case 20: return Y[9];
        case 21: return Z[1];
        case 22: return Z[2];
        case 23: return Z[4];
        case 24: return Z[5];
        default: return Z[7];
    }
}
// V[a * 10 + m]: coefficient of monomial m in C_a(o) . v
__device__ inline double vToAxisCoeff(const double* v, int q) {
    switch (q) {
        case 0: return v[0];  case 1: return v[3];  case 2: return v[4];  case 3: return v[5];  case 4: return v[6];
        case 5: return v[7];  case 6: return v[8];  case 7: return v[9];  case 8: return v[10]; case 9: return v[11];
        case 10: return v[1]; case 11: return v[12]; case 12: return v[13]; case 13: return v[14]; case 14: return v[15];
        case 15: return v[16]; case 16: return v[17]; case 17: return v[18]; case 18: return v[19]; case 19: return v[20];
        case 20: return v[2]; case 21: return v[21]; case 22: return v[22]; case 23: return -v[3] - v[13]; case 24: return v[23];
        case 25: return v[24]; case 26: return -2. * v[6] - v[16]; case 27: return v[25]; case 28: return -v[7] - 2. * v[18];
        default: return -0.5 * v[8] - 0.5 * v[19];
    }
}
#endif
// The library is compiled with -ffp-contract=off (the fp32 SDF sampling must match the CPU restatement bit for bit).  The moment sums and the
// expansion of the tile apply are fp64 dot products with no such constraint (parity bound: 1e-10 on the operator): PS_TILE_FMA fuses them —
// half the VALU instructions of the kernel's two loops (r05; -DPS_TILE_NO_FMA: the separate multiply and add of r01 - r04).
#ifdef PS_TILE_NO_FMA
#define PS_TILE_FMA(a, b, c) ((a) * (b) + (c))
#else
#define PS_TILE_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif
// The 30 moments of a wave, summed over its 64 lanes, through LDS: 30 x (six DPP steps of two moves and an add) was 540 of the ~5000 VALU
// instructions a wave spends on a tile, in a kernel that is VALU-bound (profiles/r05_tile_apply_valu.txt).  Here the lanes store fifteen values at a
// time, lane l then adds the sixteen lanes 16 (l >> 4) .. of value l & 15 in lane order, and the four segment sums of a value are added in
// segment order: ~110 instructions, a fixed summation order (lanes 0..15, 16..31, 32..47, 48..63, then the four).  scratch: 15 x 65 + 64 doubles of
// the wave's own; out[0..29] written by the lanes that hold the totals; the caller synchronises before reading out.
constexpr int TILE_RED_SCRATCH = 15 * 65 + 64;
__device__ inline void waveSum30(const double* M, double* __restrict__ scratch, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int m = 0; m < 15; ++m) scratch[m * 65 + lane] = M[15 * h + m];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double p = 0.;
        if (n < 15) {
#pragma unroll
            for (int j = 0; j < 16; ++j) p += scratch[n * 65 + 16 * q + j];
        }
        scratch[15 * 65 + lane] = p;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < 15) out[15 * h + lane] = ((scratch[15 * 65 + lane] + scratch[15 * 65 + 16 + lane]) + scratch[15 * 65 + 32 + lane]) + scratch[15 * 65 + 48 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// one lane's share of the moments over the rows rr = first, first + stride, ... < end
// packed faces a lane keeps in registers between the gather and the expand of k_tile_apply: all of a lane's rows in the 256-thread form (few tiles:
// latency bound), 12 in the one-wave-per-tile form (126 VGPRs: the fourth wave per SIMD, i.e. all 4096 tiles of the 256^3 cavity resident at once)
template <int TB> struct TileFaceCache { static constexpr int N = TB >= 256 ? 16 : 12; };
template <bool CACHE, int U, int FC, class TS = double>
__device__ inline void tileAccumulate(int first, int stride, int end, const uint32_t* __restrict__ rrowFace, const TS* __restrict__ sred, double dx, int3 off,
                                      double cx, double cy, double cz, double* __restrict__ M, uint32_t* __restrict__ fcache) {
    // U (face, s) pairs are requested together: independent loads in flight, then the arithmetic
    int it = 0;
    for (int base = first; base < end; base += U * stride, ++it) {
        uint32_t f[U];
        double s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = base + u * stride;
            const bool ok = rr < end;
            f[u] = ok ? __builtin_nontemporal_load(rrowFace + rr) : 0u;
            s[u] = ok ? (double)__builtin_nontemporal_load(sred + rr) : 0.;   // 0 past the end: contributes nothing
        }
        if (CACHE && it < FC / U) {
#pragma unroll
            for (int q = 0; q < FC / U; ++q)
                if (q == it) {
#pragma unroll
                    for (int u = 0; u < U; ++u) fcache[q * U + u] = f[u];
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double mu[10];
            int axis;
            faceMonomials(f[u], dx, off, cx, cy, cz, mu, &axis);
            // A region's rows are ordered (item, long / short, AXIS, position) (ps_blocks.hip: k_skin): the 64 consecutive rows a wave takes
            // in one slot lie on ONE axis except where they straddle one of the few category boundaries.  The three axes' ten moments are
            // updated in turn, a block skipped (scalar branch) when no live lane of the wave is on its axis: the skipped updates would add
            // mu * 0 — every partial sum keeps its value, the VALU work of the loop falls from 30 to ~11 fused multiply-adds per row (r05)
            const bool real = base + u * stride < end;
#ifdef PS_TILE_NO_SKIP   // A/B build (scripts/build_variant.sh): all three blocks, always
            const unsigned long long on0 = 1ull | (unsigned long long)real, on1 = 1ull, on2 = 1ull;
#else
            const unsigned long long on0 = __ballot(real && axis == 0), on1 = __ballot(real && axis == 1), on2 = __ballot(real && axis == 2);
#endif
            if (on0 != 0ull) {
                const double sa = axis == 0 ? s[u] : 0.;
#pragma unroll
                for (int m = 0; m < 10; ++m) M[m] = PS_TILE_FMA(mu[m], sa, M[m]);
            }
            if (on1 != 0ull) {
                const double sa = axis == 1 ? s[u] : 0.;
#pragma unroll
                for (int m = 0; m < 10; ++m) M[10 + m] = PS_TILE_FMA(mu[m], sa, M[10 + m]);
            }
            if (on2 != 0ull) {
                const double sa = axis == 2 ? s[u] : 0.;
#pragma unroll
                for (int m = 0; m < 10; ++m) M[20 + m] = PS_TILE_FMA(mu[m], sa, M[20 + m]);
            }
        }
    }
}
// three-kernel tile apply, step 1 (regions too large for one workgroup): partial w (26) of one chunk of <= RC_ROWS rows of ONE
// region.  One wavefront per chunk, lane-strided rows, wave-shuffle reduction of the 30 moments; lane n < 26 stores entry n of w.
__global__ void __launch_bounds__(64) k_tile_gather(const int32_t* __restrict__ chunkRegion, const int32_t* __restrict__ chunkStart,
                                                    const int32_t* __restrict__ chunkEnd, const uint32_t* __restrict__ rrowFace,
                                                    const double* __restrict__ COM, double dx, int3 off, const double* __restrict__ sred,
                                                    double* __restrict__ wpart, const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double Ms[30];
    const int ch = blockIdx.x;
    const int r = chunkRegion[ch];
    double M[30];
#pragma unroll
    for (int n = 0; n < 30; ++n) M[n] = 0.;
    tileAccumulate<false, 4, 4>(chunkStart[ch] + (int)threadIdx.x, 64, chunkEnd[ch], rrowFace, sred, dx, off, COM[(int64_t)r * 3], COM[(int64_t)r * 3 + 1], COM[(int64_t)r * 3 + 2], M, nullptr);
#pragma unroll
    for (int n = 0; n < 30; ++n) {
        const double v = waveSumToLane63(M[n]);
        if (threadIdx.x == 63) Ms[n] = v;
    }
    __syncthreads();
    if (threadIdx.x < PS_RD) wpart[(int64_t)ch * PS_RD + threadIdx.x] = momentsToW(Ms, (int)threadIdx.x);
}
// Fused tile apply: ONE workgroup per region gathers w = J^T s over the region's skin rows, multiplies by the 26x26 block and
// expands t = J v in place — no partial-w round trip, one launch instead of three (regions up to TILE_FUSED_MAX_ROWS rows).
//   MODE 0: v = BInv w, t = J v                         (operator apply)
//   MODE 1: v = BInv (invDt rhsR - w) -> vreg, no expand (velocity recovery, Solver.cpp:509)
//   MODE 2: v = invDt BInv rhsR, t = J v, no gather      (right-hand side, AssembleSystem.cpp:448-452)
// TS: element type of the face-row vector (float: the inner applies of the single-precision Chebyshev polynomial, ps_kernels_spmv.hpp: VecIO)
template <int MODE, int TB, class TS = double>
__global__ void __launch_bounds__(TB) k_tile_apply(const int32_t* __restrict__ regionRowPtr, const uint32_t* __restrict__ rrowFace,
                                                   const double* __restrict__ COM, double dx, int3 off, const double* __restrict__ Binv,
                                                   const double* __restrict__ rhsR, double invDt, TS* __restrict__ sred,
                                                   double* __restrict__ vreg, const int* __restrict__ done, double* __restrict__ wvPart) {
    if (done && *done) return;
    __shared__ double msum[TB / 64][30];
    __shared__ double redScratch[MODE != 2 ? TB / 64 : 1][MODE != 2 ? TILE_RED_SCRATCH : 1];
    __shared__ double Ms[30], wv[PS_RD], vv[PS_RD], Vs[30];
    // 8.3 KB of reduction scratch per wave: the instantiations that ship (TB <= 256: 33 KB) stay inside the 64 KB a workgroup may take on any
    // CDNA target; TB = 512 / 1024 (PS_TILE_TB experiments, lab build only) need gfx950's 160 KB
    static_assert(TB > 256 || sizeof(msum) + sizeof(redScratch) + sizeof(Ms) + sizeof(wv) + sizeof(vv) + sizeof(Vs) <= 64 * 1024, "k_tile_apply: static LDS of a default instantiation above 64 KB");
    static_assert(sizeof(msum) + sizeof(redScratch) + sizeof(Ms) + sizeof(wv) + sizeof(vv) + sizeof(Vs) <= 160 * 1024, "k_tile_apply: static LDS above the 160 KB of gfx950");
    const int r = blockIdx.x;
    const int r0 = regionRowPtr[r], r1 = regionRowPtr[r + 1];
    const double cx = COM[(int64_t)r * 3], cy = COM[(int64_t)r * 3 + 1], cz = COM[(int64_t)r * 3 + 2];
    const int wave = threadIdx.x >> 6;
    // few regions (TB = 256: small grids, latency bound): a lane requests all its ~13 rows at once; many regions (TB = 64): 4 at a
    // time, occupancy hides the latency
    constexpr int U = TB >= 256 ? 16 : 4;
    constexpr int FC = TileFaceCache<TB>::N;
    uint32_t fcache[FC];
    if (MODE != 2) {
        double M[30];
#pragma unroll
        for (int n = 0; n < 30; ++n) M[n] = 0.;
        tileAccumulate<MODE == 0, U, FC, TS>(r0 + (int)threadIdx.x, TB, r1, rrowFace, sred, dx, off, cx, cy, cz, M, fcache);
#ifdef PS_TILE_DPP_REDUCE   // A/B build: the DPP ladder per moment of r01 - r04
#pragma unroll
        for (int n = 0; n < 30; ++n) {
            const double v = waveSumToLane63(M[n]);
            if ((threadIdx.x & 63) == 63) msum[wave][n] = v;
        }
#else
        waveSum30(M, redScratch[wave], msum[wave]);
#endif
        __syncthreads();
        if (threadIdx.x < 30) {
            double s = 0.;
#pragma unroll
            for (int q = 0; q < TB / 64; ++q) s += msum[q][threadIdx.x];
            Ms[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x < PS_RD) {
            double s = momentsToW(Ms, (int)threadIdx.x);
            if (MODE == 1) s = invDt * rhsR[(int64_t)r * PS_RD + threadIdx.x] - s;
            wv[threadIdx.x] = s;
        }
    } else if (threadIdx.x < PS_RD) wv[threadIdx.x] = rhsR[(int64_t)r * PS_RD + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < PS_RD) {
        const double* B = Binv + (int64_t)r * PS_RD * PS_RD + threadIdx.x * PS_RD;
        double s = 0.;
#pragma unroll
        for (int n = 0; n < PS_RD; ++n) s += B[n] * wv[n];
        if (MODE == 2) s *= invDt;
        vv[threadIdx.x] = s;
        if (MODE == 1) vreg[(int64_t)r * PS_RD + threadIdx.x] = s;
    }
    if (MODE == 1) return;
    __syncthreads();
    // w . v = w^T BInv w = sum over the tile's rows of s_f t_f: the tile's share of x . A x (fused residual update, ps_solve.hip)
    if (MODE == 0 && wvPart && threadIdx.x == 0) {
        double s = 0.;
#pragma unroll
        for (int n = 0; n < PS_RD; ++n) s += wv[n] * vv[n];
        wvPart[r] = s;
    }
    if (threadIdx.x < 30) Vs[threadIdx.x] = vToAxisCoeff(vv, (int)threadIdx.x);
    __syncthreads();
    int it = 0;
    for (int base = r0 + (int)threadIdx.x; base < r1; base += U * TB, ++it) {
        uint32_t f[U];
        if (MODE == 0 && it < FC / U) {   // the faces this lane already decoded in the gather
#pragma unroll
            for (int q = 0; q < FC / U; ++q)
                if (q == it) {
#pragma unroll
                    for (int u = 0; u < U; ++u) f[u] = fcache[q * U + u];
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) f[u] = base + u * TB < r1 ? rrowFace[base + u * TB] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * TB >= r1) break;
            double mu[10];
            int axis;
            faceMonomials(f[u], dx, off, cx, cy, cz, mu, &axis);
            const double* V = Vs + 10 * axis;                        // LDS broadcast-ish reads (3 distinct rows per wave)
            double t = 0.;
#pragma unroll
            for (int m = 0; m < 10; ++m) t = PS_TILE_FMA(mu[m], V[m], t);
            sred[base + u * TB] = (TS)t;
        }
    }
}
// MODE 0: v = BInv w ;  MODE 1: v = BInv (invDt*rhsR - w)  (velocity recovery, Solver.cpp:509)
// MODE 2: v = invDt * BInv rhsR  (right-hand side, AssembleSystem.cpp:448-452; no gather)
template <int MODE>
__global__ void __launch_bounds__(64) k_tile_solve(const int32_t* __restrict__ regionChunkPtr, const double* __restrict__ wpart,
                                                   const double* __restrict__ Binv, const double* __restrict__ rhsR, double invDt,
                                                   double* __restrict__ vreg, const int* __restrict__ done, double* __restrict__ wvPart) {
    if (done && *done) return;
    __shared__ double w[PS_RD];
    __shared__ double vloc[PS_RD];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (lane < PS_RD) {
        double s = 0.;
        if (MODE != 2)
            for (int ch = regionChunkPtr[r]; ch < regionChunkPtr[r + 1]; ++ch) s += wpart[(int64_t)ch * PS_RD + lane];
        if (MODE == 1) s = invDt * rhsR[(int64_t)r * PS_RD + lane] - s;
        if (MODE == 2) s = rhsR[(int64_t)r * PS_RD + lane];
        w[lane] = s;
    }
    __syncthreads();
    if (lane < PS_RD) {
        const double* B = Binv + (int64_t)r * PS_RD * PS_RD + lane * PS_RD;
        double s = 0.;
#pragma unroll
        for (int n = 0; n < PS_RD; ++n) s += B[n] * w[n];
        if (MODE == 2) s *= invDt;
        vreg[(int64_t)r * PS_RD + lane] = s;
        vloc[lane] = s;
    }
    if (MODE == 0 && wvPart) {
        __syncthreads();
        if (lane == 0) {
            double s = 0.;
#pragma unroll
            for (int n = 0; n < PS_RD; ++n) s += w[n] * vloc[n];
            wvPart[r] = s;
        }
    }
}
// t_f = C_f . v_region(f).  One block per chunk of <= RC_ROWS rows of ONE region: the 26 coefficients are block-uniform
// (scalar loads), each thread expands RC_ROWS/256 rows.
__global__ void __launch_bounds__(BS) k_tile_expand(const int32_t* __restrict__ chunkRegion, const int32_t* __restrict__ chunkStart,
                                                    const int32_t* __restrict__ chunkEnd, const uint32_t* __restrict__ rrowFace,
                                                    const double* __restrict__ COM, double dx, int3 off, const double* __restrict__ vreg,
                                                    double* __restrict__ tred, const int* __restrict__ done) {
    if (done && *done) return;
    const int ch = blockIdx.x;
    const int r = chunkRegion[ch];
    const int b0 = chunkStart[ch], e = chunkEnd[ch];
    const double cx = COM[(int64_t)r * 3 + 0], cy = COM[(int64_t)r * 3 + 1], cz = COM[(int64_t)r * 3 + 2];
    double v[PS_RD];
#pragma unroll
    for (int n = 0; n < PS_RD; ++n) v[n] = vreg[(int64_t)r * PS_RD + n];
    constexpr int PER = RC_ROWS / BS;
    uint32_t fq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int rr = b0 + threadIdx.x + q * BS;
        fq[q] = rr < e ? rrowFace[rr] : 0u;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int rr = b0 + threadIdx.x + q * BS;
        if (rr < e) {
            int i, j, k, axis;
            unpackFace(fq[q], i, j, k, axis);
            const double ox = ((double)(i + off.x) - (axis == 0 ? 0.5 : 0.)) * dx - cx;
            const double oy = ((double)(j + off.y) - (axis == 1 ? 0.5 : 0.)) * dx - cy;
            const double oz = ((double)(k + off.z) - (axis == 2 ? 0.5 : 0.)) * dx - cz;
            tred[rr] = basisDot(ox, oy, oz, axis, v);
        }
    }
}

