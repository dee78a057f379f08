// Per-tile reduced apply: J^T (gather), 26x26 BInv (solve), J (expand).
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once

// ---- per-tile reduced apply -------------------------------------------------------------------------
__device__ inline void rowOffset(uint32_t packed, const double* __restrict__ COM, int region, double dx, double* o, int* axis) {
    int i, j, k, a;
    unpackFace(packed, i, j, k, a);
    double p[3] = {(double)i, (double)j, (double)k};
    p[a] -= 0.5;
    o[0] = p[0] * dx - COM[(int64_t)region * 3 + 0];
    o[1] = p[1] * dx - COM[(int64_t)region * 3 + 1];
    o[2] = p[2] * dx - COM[(int64_t)region * 3 + 2];
    *axis = a;
}
// partial w (26) of one chunk of <= RC_ROWS reduced rows of ONE face axis:  w += C_f * s_f.  One wavefront per chunk:
// all RC_ROWS/64 (face, s) pairs of a lane are requested up front (independent loads in flight together), then only
// the 10 / 10 / 14 non-zero entries of that axis' basis row (buildConversionCoefficients, Solver.cpp:2112-2145) are
// accumulated in registers and wave-shuffle reduced; no LDS, no barrier.
template <int AXIS>
__device__ inline void tileGatherAxis(int b0, int e, const uint32_t* __restrict__ rrowFace, const double* __restrict__ sred, double dx,
                                      double cx, double cy, double cz, double* __restrict__ wout) {
    constexpr int PER = RC_ROWS / 64;
    constexpr int NW = AXIS == 2 ? 14 : 10;
    uint32_t fq[PER];
    double sq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int rr = b0 + threadIdx.x + q * 64;
        const bool ok = rr < e;
        fq[q] = ok ? __builtin_nontemporal_load(rrowFace + rr) : 0u;
        sq[q] = ok ? __builtin_nontemporal_load(sred + rr) : 0.;     // 0 for the lanes past the end: contributes nothing
    }
    double w[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) w[n] = 0.;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        int i, j, k, axis;
        unpackFace(fq[q], i, j, k, axis);
        const double s = sq[q];
        const double ox = ((double)i - (AXIS == 0 ? 0.5 : 0.)) * dx - cx;
        const double oy = ((double)j - (AXIS == 1 ? 0.5 : 0.)) * dx - cy;
        const double oz = ((double)k - (AXIS == 2 ? 0.5 : 0.)) * dx - cz;
        if constexpr (AXIS != 2) {    // x-row: entries 0,3..11 ; y-row: entries 1,12..20
            w[0] += s; w[1] += ox * s; w[2] += oy * s; w[3] += oz * s;
            w[4] += ox * ox * s; w[5] += ox * oy * s; w[6] += ox * oz * s; w[7] += oy * oy * s; w[8] += oy * oz * s; w[9] += oz * oz * s;
        } else {            // z-row: entries 2,3,6,7,8,13,16,18,19,21..25
            w[0] += s; w[1] += (-oz) * s; w[2] += (-2. * ox * oz) * s; w[3] += (-1. * oy * oz) * s; w[4] += (-0.5 * oz * oz) * s;
            w[5] += (-oz) * s; w[6] += (-1. * ox * oz) * s; w[7] += (-2. * oy * oz) * s; w[8] += (-0.5 * oz * oz) * s;
            w[9] += ox * s; w[10] += oy * s; w[11] += ox * ox * s; w[12] += ox * oy * s; w[13] += oy * oy * s;
        }
    }
    constexpr int slotX[10] = {0, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    constexpr int slotY[10] = {1, 12, 13, 14, 15, 16, 17, 18, 19, 20};
    constexpr int slotZ[14] = {2, 3, 6, 7, 8, 13, 16, 18, 19, 21, 22, 23, 24, 25};
    // lane n < 26 ends up holding entry n of the chunk's partial w (0 for the entries this axis never touches): one store
    double mine = 0.;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const double v = __shfl(waveReduceSum(w[n]), 0);
        if ((int)threadIdx.x == (AXIS == 0 ? slotX[n] : (AXIS == 1 ? slotY[n] : slotZ[n]))) mine = v;
    }
    if (threadIdx.x < PS_RD) wout[threadIdx.x] = mine;
}
__global__ void __launch_bounds__(64) k_tile_gather(const int32_t* __restrict__ chunkRegion, const int32_t* __restrict__ chunkStart,
                                                    const int32_t* __restrict__ chunkEnd, const int32_t* __restrict__ chunkAxis,
                                                    const uint32_t* __restrict__ rrowFace, const double* __restrict__ COM, double dx,
                                                    const double* __restrict__ sred, double* __restrict__ wpart, const int* __restrict__ done) {
    if (done && *done) return;
    const int ch = blockIdx.x;
    const int r = chunkRegion[ch];
    const double cx = COM[(int64_t)r * 3 + 0], cy = COM[(int64_t)r * 3 + 1], cz = COM[(int64_t)r * 3 + 2];
    const int e = chunkEnd[ch], b0 = chunkStart[ch], axis = chunkAxis[ch];
    double* wout = wpart + (int64_t)ch * PS_RD;
    if (axis == 0) tileGatherAxis<0>(b0, e, rrowFace, sred, dx, cx, cy, cz, wout);
    else if (axis == 1) tileGatherAxis<1>(b0, e, rrowFace, sred, dx, cx, cy, cz, wout);
    else tileGatherAxis<2>(b0, e, rrowFace, sred, dx, cx, cy, cz, wout);
}
// MODE 0: v = BInv w ;  MODE 1: v = BInv (invDt*rhsR - w)  (velocity recovery, Solver.cpp:509)
// MODE 2: v = invDt * BInv rhsR  (right-hand side, AssembleSystem.cpp:448-452; no gather)
template <int MODE>
__global__ void __launch_bounds__(64) k_tile_solve(const int32_t* __restrict__ regionChunkPtr, const double* __restrict__ wpart,
                                                   const double* __restrict__ Binv, const double* __restrict__ rhsR, double invDt,
                                                   double* __restrict__ vreg, const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double w[PS_RD];
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
    }
}
// t_f = C_f . v_region(f).  One block per chunk of <= RC_ROWS rows of ONE region: the 26 coefficients are block-uniform
// (scalar loads), each thread expands RC_ROWS/256 rows.
__global__ void __launch_bounds__(BS) k_tile_expand(const int32_t* __restrict__ chunkRegion, const int32_t* __restrict__ chunkStart,
                                                    const int32_t* __restrict__ chunkEnd, const uint32_t* __restrict__ rrowFace,
                                                    const double* __restrict__ COM, double dx, const double* __restrict__ vreg,
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
            const double ox = ((double)i - (axis == 0 ? 0.5 : 0.)) * dx - cx;
            const double oy = ((double)j - (axis == 1 ? 0.5 : 0.)) * dx - cy;
            const double oz = ((double)k - (axis == 2 ? 0.5 : 0.)) * dx - cz;
            tred[rr] = basisDot(ox, oy, oz, axis, v);
        }
    }
}

