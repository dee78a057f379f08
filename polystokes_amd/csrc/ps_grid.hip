// Grid stages on the GPU: volume-fraction weights, cell/face/edge classification, boundary-layer
// floods, tiling, connected reduced regions, DOF numbering.  One thread per voxel, coalesced x-fastest
// reads; neighbour reads hit L1/L2 (each voxel is re-read by at most its 6 neighbours).
// Reference: exec/HDK_PolyStokesSolver.cpp:238-326, exec/HDK_PolyStokesSolver_Classifier.cpp (all).
#include <limits.h>

#include "ps_context.hpp"

using namespace ps;

namespace {

constexpr int BS = 256;

// sample offsets inside a voxel (Solver.h:193-222), order centre, faceX, faceY, faceZ, edgeYZ, edgeXZ, edgeXY: 0.5 (true) or 0 per axis
constexpr bool kSampleOffsetHalf[7][3] = {{true, true, true}, {false, true, true}, {true, false, true}, {true, true, false},
                                          {true, false, false}, {false, true, false}, {false, false, true}};

// SIM_RawField::getValue(pos) restated: trilinear between voxel centres, streak border, fp32,
// lerp(a,b,t) = a + (b-a)*t, x then y then z.  The library is built with -ffp-contract=off so the
// result is bit-identical to the CPU restatement.
__device__ inline float sampleCenterField(const float* __restrict__ f, int nx, int ny, int nz, float px, float py, float pz) {
    const int n[3] = {nx, ny, nz};
    const float p[3] = {px, py, pz};
    int i0[3], i1[3];
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = p[a] - 0.5f;
        if (u < 0.f) u = 0.f;
        if (u > (float)(n[a] - 1)) u = (float)(n[a] - 1);
        int b = (int)u;
        if (b >= n[a] - 1) { b = n[a] - 1; i0[a] = b; i1[a] = b; t[a] = 0.f; }
        else { i0[a] = b; i1[a] = b + 1; t[a] = u - (float)b; }
    }
    const int64_t sx = 1, sy = nx, sz = (int64_t)nx * ny;
    auto at = [&](int i, int j, int k) { return f[i * sx + j * sy + k * sz]; };
    auto L = [](float a, float b, float tt) { return a + (b - a) * tt; };
    const float c00 = L(at(i0[0], i0[1], i0[2]), at(i1[0], i0[1], i0[2]), t[0]);
    const float c10 = L(at(i0[0], i1[1], i0[2]), at(i1[0], i1[1], i0[2]), t[0]);
    const float c01 = L(at(i0[0], i0[1], i1[2]), at(i1[0], i0[1], i1[2]), t[0]);
    const float c11 = L(at(i0[0], i1[1], i1[2]), at(i1[0], i1[1], i1[2]), t[0]);
    const float c0 = L(c00, c10, t[1]);
    const float c1 = L(c01, c11, t[1]);
    return L(c0, c1, t[2]);
}

// computeSDFWeightsSampled(sdf, 2, invert=false, minweight=0): Solver.cpp:292-326 (HDK body out of tree,
// restated: fraction of the 2x2x2 sub-samples of the voxel box whose SDF value is < 0).
// All seven sample grids in one launch.  Along an axis the sub-samples of the voxel index q sit at u = q - 0.75, q - 0.25 or q + 0.25
// (cell-centre coordinates): the grids centred on that axis use the last two, the grids on a face of it the first two.  So the 56
// sub-samples of the seven grids at (i, j, k) are 27 distinct points, and their trilinear values share the lerps: 27 along x (two
// loads each), 27 along y, 27 along z — the SAME operations on the same operands as sampleCenterField above does for each of them
// (x, then y, then z; the index / clamp arithmetic per axis is that function's), so every value is bit-identical; 54 loads per voxel
// instead of 448 (the r02 kernels, one launch per grid, were bound by exactly those loads: 14 launches, 2.6 ms at 256^3).
struct AxisSamples { int i0[3], i1[3]; float t[3]; };
__device__ inline AxisSamples axisSamples(int q, int n) {
    AxisSamples A;
    const float pos[3] = {((float)q + 0.f) + -0.25f, ((float)q + 0.f) + 0.25f, ((float)q + 0.5f) + 0.25f};   // (index + offset) +- 0.25, formed as the oracle forms it
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float u = pos[c] - 0.5f;
        if (u < 0.f) u = 0.f;
        if (u > (float)(n - 1)) u = (float)(n - 1);
        int b = (int)u;
        if (b >= n - 1) { b = n - 1; A.i0[c] = b; A.i1[c] = b; A.t[c] = 0.f; }
        else { A.i0[c] = b; A.i1[c] = b + 1; A.t[c] = u - (float)b; }
    }
    return A;
}
__global__ void __launch_bounds__(BS) k_sdf_weights7(Grid g, const float* __restrict__ sdf, int negate, Set7<float> dst) {
    const int ex = g.nx + 1, ey = g.ny + 1;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)ex * ey * (g.nz + 1)) return;
    const int i = (int)(c % ex), j = (int)((c / ex) % ey), k = (int)(c / ((int64_t)ex * ey));
    const AxisSamples X = axisSamples(i, g.nx), Y = axisSamples(j, g.ny), Z = axisSamples(k, g.nz);
    auto L = [](float a, float b, float tt) { return a + (b - a) * tt; };
    // the cell rows / planes the y and z samples touch: at most three distinct ones each (j - 1 .. j + 1 clamped), named by value
    int jr[3] = {Y.i0[0], Y.i0[2], Y.i1[2]}, kr[3] = {Z.i0[0], Z.i0[2], Z.i1[2]};
    // (i0[0] = i0[1] <= i0[2] <= i1[2]; i1[0] = i1[1] is i0[2] or — at the upper border — i0[0]; so every index used is among the three)
    float xl[3][3][3];                                     // [x sample][row][plane]
#pragma unroll
    for (int pz = 0; pz < 3; ++pz)
#pragma unroll
        for (int py = 0; py < 3; ++py) {
            const float* row = sdf + (int64_t)g.nx * (jr[py] + (int64_t)g.ny * kr[pz]);
#pragma unroll
            for (int cx = 0; cx < 3; ++cx) xl[cx][py][pz] = L(row[X.i0[cx]], row[X.i1[cx]], X.t[cx]);
        }
    auto pick = [](int idx, const int* set) { return idx == set[0] ? 0 : (idx == set[1] ? 1 : 2); };
    int cnt[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int cz = 0; cz < 3; ++cz) {
        const int z0 = pick(Z.i0[cz], kr), z1 = pick(Z.i1[cz], kr);
#pragma unroll
        for (int cy = 0; cy < 3; ++cy) {
            const int y0 = pick(Y.i0[cy], jr), y1 = pick(Y.i1[cy], jr);
#pragma unroll
            for (int cx = 0; cx < 3; ++cx) {
                auto sel = [&](int py, int pz) {
                    const float a0 = pz == 0 ? xl[cx][0][0] : (pz == 1 ? xl[cx][0][1] : xl[cx][0][2]);
                    const float a1 = pz == 0 ? xl[cx][1][0] : (pz == 1 ? xl[cx][1][1] : xl[cx][1][2]);
                    const float a2 = pz == 0 ? xl[cx][2][0] : (pz == 1 ? xl[cx][2][1] : xl[cx][2][2]);
                    return py == 0 ? a0 : (py == 1 ? a1 : a2);
                };
                const float c0 = L(sel(y0, z0), sel(y1, z0), Y.t[cy]);
                const float c1 = L(sel(y0, z1), sel(y1, z1), Y.t[cy]);
                float v = L(c0, c1, Z.t[cz]);
                if (negate) v = -v;
                const int in = v < 0.f ? 1 : 0;
                // which grids own this sub-sample: per axis, sample index 1 belongs to both kinds, 0 to the grids on a face of the
                // axis (offset 0), 2 to the grids centred on it (offset 0.5)
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const bool hx = kSampleOffsetHalf[s][0], hy = kSampleOffsetHalf[s][1], hz = kSampleOffsetHalf[s][2];
                    if ((hx ? cx >= 1 : cx <= 1) && (hy ? cy >= 1 : cy <= 1) && (hz ? cz >= 1 : cz <= 1)) cnt[s] += in;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int3 d = g.dims(s);
        if (i < d.x && j < d.y && k < d.z) dst.p[s][lin3(d, i, j, k)] = (float)cnt[s] / 8.0f;
    }
}

// Classifier.cpp:56-128
__global__ void k_classify_cells(Grid g, Set7<const float> lw, Set7<const float> fw, int32_t* __restrict__ lab) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    bool inSolve = lw.p[0][c] > 0.f;
    if (!inSolve) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int3 fd = g.dims(1 + a);
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                int3 f = q;
                addc(f, a, dir);
                if (lw.p[1 + a][lin3(fd, f.x, f.y, f.z)] > 0.f) inSolve = true;
            }
        }
    }
    const bool inFluid = !(fw.p[0][c] == 0.f);
    lab[c] = inSolve ? (inFluid ? PS_GENERICFLUID : PS_SOLID) : PS_UNSOLVED;
}

// buildInitialAirBoundaryLayer, Classifier.cpp:364-430.  mark = 1 for layer-0 cells.
__global__ void k_air_layer0(Grid g, Set7<const float> lw, int32_t* __restrict__ lab, int32_t* __restrict__ mark, int setActive) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    int m = 0;
    if (lab[c] == PS_GENERICFLUID) {
        const int3 q = unlin3(d, c);
        bool bnd = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int3 fd = g.dims(1 + a);
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                int3 nb = q;
                addc(nb, a, dir ? 1 : -1);
                if (comp(nb, a) < 0 || comp(nb, a) >= comp(d, a)) continue;
                int3 f = q;
                addc(f, a, dir);
                const int nl = lab[lin3(d, nb.x, nb.y, nb.z)];
                // neighbours may concurrently turn GENERIC->ACTIVE; only UNSOLVED matters here
                if (nl == PS_UNSOLVED) bnd = true;
                if (lw.p[1 + a][lin3(fd, f.x, f.y, f.z)] < 1.f) bnd = true;
            }
        }
        if (bnd) m = 1;
    }
    mark[c] = m;
    if (m && setActive) lab[c] = PS_ACTIVEFLUID;
}

// buildNextLiquidBoundaryLayer (Classifier.cpp:432-508) + setActiveLayerCells (Solver.cpp:2022-2060):
// a GENERICFLUID cell joins layer `cur+1` if a neighbour is in layer `cur` across a face with liquid > 0.
__global__ void k_air_next(Grid g, Set7<const float> lw, int32_t* __restrict__ lab, int32_t* __restrict__ mark, int cur) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    if (lab[c] != PS_GENERICFLUID) return;
    const int3 q = unlin3(d, c);
    bool join = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int3 fd = g.dims(1 + a);
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            int3 nb = q;
            addc(nb, a, dir ? 1 : -1);
            if (comp(nb, a) < 0 || comp(nb, a) >= comp(d, a)) continue;
            int3 f = q;
            addc(f, a, dir);   // the face between q and nb: cellToFaceMap(nb, a, 1-dir) == cellToFaceMap(q, a, dir)
            if (mark[lin3(d, nb.x, nb.y, nb.z)] == cur && lw.p[1 + a][lin3(fd, f.x, f.y, f.z)] > 0.f) join = true;
        }
    }
    if (join) { mark[c] = cur + 1; lab[c] = PS_ACTIVEFLUID; }
}

// buildInitialSolidBoundaryLayer, Classifier.cpp:573-641.  mark = 1 -> VISITED in layer 0.
__global__ void k_solid_layer0(Grid g, int32_t* __restrict__ lab, int32_t* __restrict__ mark, const int32_t* __restrict__ labIn) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    int m = 0;
    const int l = labIn[c];
    if (l == PS_GENERICFLUID || l == PS_ACTIVEFLUID) {
        const int3 q = unlin3(d, c);
        bool bnd = false;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                int3 nb = q;
                addc(nb, a, dir ? 1 : -1);
                if (comp(nb, a) < 0 || comp(nb, a) >= comp(d, a)) { bnd = true; continue; }   // :617-621
                if (labIn[lin3(d, nb.x, nb.y, nb.z)] == PS_SOLID) bnd = true;
            }
        if (bnd) m = 1;
    }
    mark[c] = m;
    if (m) lab[c] = PS_ACTIVEFLUID;
}

// buildNextSolidBoundaryLayer, Classifier.cpp:643-703
__global__ void k_solid_next(Grid g, Set7<const float> lw, int32_t* __restrict__ lab, int32_t* __restrict__ mark, int cur) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    if (mark[c] != 0) return;   // VISITED already
    const int l = lab[c];
    if (l != PS_ACTIVEFLUID && l != PS_GENERICFLUID) return;
    const int3 q = unlin3(d, c);
    bool join = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int3 fd = g.dims(1 + a);
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            int3 nb = q;
            addc(nb, a, dir ? 1 : -1);
            if (comp(nb, a) < 0 || comp(nb, a) >= comp(d, a)) continue;
            int3 f = q;
            addc(f, a, dir);
            if (mark[lin3(d, nb.x, nb.y, nb.z)] == cur && lw.p[1 + a][lin3(fd, f.x, f.y, f.z)] > 0.f) join = true;
        }
    }
    if (join) { mark[c] = cur + 1; lab[c] = PS_ACTIVEFLUID; }
}

// constructTiles (Classifier.cpp:705-746) + overwriteIndices GENERIC->REDUCED (:189)
__global__ void k_tiles_and_relabel(Grid g, int32_t* __restrict__ lab, int doTile, int tileSize, int pad) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    if (lab[c] != PS_GENERICFLUID) return;
    int out = PS_REDUCED;
    if (doTile) {
        const int3 q = unlin3(d, c);
        const int mx = q.x % tileSize, my = q.y % tileSize, mz = q.z % tileSize;
        if (mx < pad || my < pad || mz < pad) out = PS_ACTIVEFLUID;
    }
    lab[c] = out;
}

__global__ void k_relabel(int32_t* __restrict__ lab, int64_t n, int from, int to) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n && lab[c] == from) lab[c] = to;
}
struct FillMany { int32_t* p[16]; int64_t n[16]; };   // blockIdx.y = the array, grid-stride over its entries
__global__ void k_fill_i32_many(FillMany F, int v) {
    int32_t* __restrict__ a = F.p[blockIdx.y];
    const int64_t n = F.n[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void k_fill_i32(int32_t* __restrict__ a, int64_t n, int v) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) a[c] = v;
}

// findFaceLabelFromCenter, Classifier.cpp:784-832
__global__ void k_classify_faces(Grid g, int axis, Set7<const float> lw, Set7<const float> fw, int32_t* __restrict__ lab) {
    const int3 d = g.dims(1 + axis);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    const int3 cd = g.dims(0);
    bool act = false;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        int3 cc = q;
        addc(cc, axis, dir - 1);
        if (oob3(cd, cc.x, cc.y, cc.z)) continue;
        if (lw.p[0][lin3(cd, cc.x, cc.y, cc.z)] > 0.f) act = true;
    }
    if (!act) {
        for (int ea = 0; ea < 3 && !act; ++ea) {
            if (ea == axis) continue;
            const int3 ed = g.dims(4 + ea);
            for (int dir = 0; dir < 2; ++dir) {
                int3 e = q;
                addc(e, 3 - axis - ea, dir);
                if (lw.p[4 + ea][lin3(ed, e.x, e.y, e.z)] > 0.f) { act = true; break; }
            }
        }
    }
    int r = PS_UNSOLVED;
    if (act) r = fw.p[1 + axis][c] < 0.5f ? PS_SOLID : PS_GENERICFLUID;
    lab[c] = r;
}

__device__ inline float streakRead(const float* __restrict__ f, const int3 d, int i, int j, int k) {
    i = i < 0 ? 0 : (i >= d.x ? d.x - 1 : i);
    j = j < 0 ? 0 : (j >= d.y ? d.y - 1 : j);
    k = k < 0 ? 0 : (k >= d.z ? d.z - 1 : k);
    return f[lin3(d, i, j, k)];
}

// findEdgeLabelFromFaceAlt, Classifier.cpp:1021-1067
__global__ void k_classify_edges(Grid g, int e, Set7<const float> lw, Set7<const float> fw, int32_t* __restrict__ lab) {
    const int3 d = g.dims(4 + e);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    bool in = lw.p[4 + e][c] != 0.f && fw.p[4 + e][c] != 0.f;
    if (in) {
        const int fa = e == 0 ? 1 : 0, fb = e == 2 ? 1 : 2;
        const int3 da = g.dims(1 + fa), db = g.dims(1 + fb);
        int3 a2 = q; addc(a2, fb, -1);
        int3 b2 = q; addc(b2, fa, -1);
        in = streakRead(lw.p[1 + fa], da, q.x, q.y, q.z) != 0.f && !oob3(da, a2.x, a2.y, a2.z) &&
             streakRead(lw.p[1 + fa], da, a2.x, a2.y, a2.z) != 0.f &&
             streakRead(lw.p[1 + fb], db, q.x, q.y, q.z) != 0.f && !oob3(db, b2.x, b2.y, b2.z) &&
             streakRead(lw.p[1 + fb], db, b2.x, b2.y, b2.z) != 0.f;
    }
    lab[c] = in ? PS_GENERICFLUID : PS_UNSOLVED;
}

// ---- connected components of REDUCED cells through faces with liquid weight > 0 -------------
// (SIM_VolumetricConnectedComponentBuilder call site Classifier.cpp:220-229.)  Min-label propagation
// over traversal-order indices with pointer jumping; converges to the smallest order index of the
// component whatever the interleaving (labels only decrease).
// k_cc_init also folds what a step needs to know about a cell's six links (neighbour in bounds and REDUCED, the face between them
// with liquid weight > 0) into one byte per cell: a step then reads 5 bytes per cell and the labels it follows, not two label
// arrays and three weight fields (12 launches at 256^3: 3.1 -> 1.6 ms)
__global__ void k_cc_init(Grid g, const int32_t* __restrict__ lab, Set7<const float> lw, int32_t* __restrict__ cc, uint8_t* __restrict__ link) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    const bool red = lab[c] == PS_REDUCED;
    cc[c] = red ? (int32_t)ijkToOrder(d, g.order, q.x, q.y, q.z) : INT_MAX;
    int m = 0;
    if (red) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int3 fd = g.dims(1 + a);
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                int3 nb = q;
                addc(nb, a, dir ? 1 : -1);
                if (oob3(d, nb.x, nb.y, nb.z)) continue;
                int3 f = q;
                addc(f, a, dir);
                if (!(lw.p[1 + a][lin3(fd, f.x, f.y, f.z)] > 0.f)) continue;
                if (lab[lin3(d, nb.x, nb.y, nb.z)] != PS_REDUCED) continue;
                m |= 1 << (2 * a + dir);
            }
        }
    }
    link[c] = (uint8_t)m;
}
__global__ void k_cc_step(Grid g, const uint8_t* __restrict__ link, int32_t* cc, int32_t* __restrict__ changed) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int lk = link[c];
    const int mine = cc[c];
    if (mine == INT_MAX) return;                                      // not a REDUCED cell
    int m = mine;
    const int64_t stride[3] = {1, d.x, (int64_t)d.x * d.y};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            if (!((lk >> (2 * a + dir)) & 1)) continue;
            const int v = cc[c + (dir ? stride[a] : -stride[a])];
            if (v < m) m = v;
        }
    }
    // pointer jump through the current representative
    const int3 rq = orderToIjk(d, g.order, m);
    const int v2 = cc[lin3(d, rq.x, rq.y, rq.z)];
    if (v2 < m) m = v2;
    if (m < mine) { atomicMin(&cc[c], m); *changed = 1; }
}
// The same propagation inside one box of B^3 cells (B = tile size: with doTile a component never leaves its tile, the padding layer
// between two tiles is ACTIVE), labels in LDS, to the box's fix point — one launch instead of a dozen k_cc_step passes over the grid
// and two host round trips less.  Links that leave the box are left to k_cc_step, which runs afterwards until nothing changes
// (once, when every component is tile-local): the fix point — the smallest order index of each component — is the same.
__global__ void __launch_bounds__(256) k_cc_local(Grid g, int B, const uint8_t* __restrict__ link, int32_t* __restrict__ cc) {
    extern __shared__ int32_t shcc[];
    const int3 d = g.dims(0);
    const int x0 = blockIdx.x * B, y0 = blockIdx.y * B, z0 = blockIdx.z * B;
    const int ex = min(B, d.x - x0), ey = min(B, d.y - y0), ez = min(B, d.z - z0);
    const int nb = ex * ey * ez;
    uint8_t* shlk = (uint8_t*)(shcc + B * B * B);
    int any = 0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        const int lx = i % ex, ly = (i / ex) % ey, lz = i / (ex * ey);
        const int64_t c = lin3(d, x0 + lx, y0 + ly, z0 + lz);
        const int v = cc[c];
        int lk = link[c];
        // links that leave the box: not followed here
        if (lx == 0) lk &= ~1; if (lx == ex - 1) lk &= ~2;
        if (ly == 0) lk &= ~4; if (ly == ey - 1) lk &= ~8;
        if (lz == 0) lk &= ~16; if (lz == ez - 1) lk &= ~32;
        shcc[i] = v;
        shlk[i] = (uint8_t)lk;
        any |= (v != INT_MAX && lk != 0) ? 1 : 0;
    }
    if (!__syncthreads_or(any)) return;                                  // no REDUCED cell with a link inside the box
    const int st[3] = {1, ex, ex * ey};
    for (int guard = 0; guard < 4 * B * B; ++guard) {                    // (a component's longest shortest path inside the box)
        int changed = 0;
        for (int i = threadIdx.x; i < nb; i += 256) {
            const int lk = shlk[i];
            if (!lk) continue;
            // the six reads at once (a link that is not there reads the cell itself): one LDS latency per cell instead of up to seven
            int v[6];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                v[2 * a] = shcc[(lk & (1 << (2 * a))) ? i - st[a] : i];
                v[2 * a + 1] = shcc[(lk & (2 << (2 * a))) ? i + st[a] : i];
            }
            const int mine = shcc[i];
            int m = mine;
#pragma unroll
            for (int q = 0; q < 6; ++q) m = min(m, v[q]);
            if (m < mine) { atomicMin(&shcc[i], m); changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = threadIdx.x; i < nb; i += 256) {
        if (!shlk[i]) continue;
        const int lx = i % ex, ly = (i / ex) % ey, lz = i / (ex * ey);
        cc[lin3(d, x0 + lx, y0 + ly, z0 + lz)] = shcc[i];
    }
}
// cell region id = rank (in traversal order) of its component's first cell
__global__ void k_cc_assign(Grid g, const int32_t* __restrict__ lab, const int32_t* __restrict__ cc,
                            const int32_t* __restrict__ rootRank, int32_t* __restrict__ region) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    int r = PS_UNASSIGNED;
    if (lab[c] == PS_REDUCED) {
        const int3 rq = orderToIjk(d, g.order, cc[c]);
        r = rootRank[lin3(d, rq.x, rq.y, rq.z)];
    }
    region[c] = r;
}

// ---- fixReducedRegionBoundaries, Classifier.cpp:1073-1172 ------------------------------------
// The reference is a serial, in-place, traversal-ordered sweep repeated to a fix point.  Within one
// sweep, whether cell c applies its fix depends only on the fixes applied by earlier cells within
// distance 2.  F = {cells that apply the fix in this sweep} is the unique solution of a triangular
// system in traversal order; it is computed by iterating F <- eval(F) from F = {} until unchanged
// (exact after at most chain-length iterations; one iteration when nothing needs fixing).
__device__ inline bool anyEarlierFix(const Grid& g, const int3 d, const uint8_t* __restrict__ F, const int3 n, int64_t ordC) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            int3 m = n;
            addc(m, a, dir ? 1 : -1);
            if (oob3(d, m.x, m.y, m.z)) continue;
            if (F[lin3(d, m.x, m.y, m.z)] && ijkToOrder(d, g.order, m.x, m.y, m.z) < ordC) return true;
        }
    return false;
}
__global__ void k_fix_eval(Grid g, const int32_t* __restrict__ lab0, const int32_t* __restrict__ reg,
                           const uint8_t* __restrict__ Fin, uint8_t* __restrict__ Fout, int anyFin, int32_t* __restrict__ flags) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int l = lab0[c];
    const int3 q = unlin3(d, c);
    bool act = (l == PS_ACTIVEFLUID);
    int64_t ordC = 0;
    if (anyFin) ordC = ijkToOrder(d, g.order, q.x, q.y, q.z);
    if (!act && anyFin && l == PS_REDUCED) act = anyEarlierFix(g, d, Fin, q, ordC);   // demoted earlier in this sweep
    uint8_t out = 0;
    if (act) {
        bool seen = false, fix = false;
        int first = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                int3 nb = q;
                addc(nb, a, dir ? 1 : -1);
                if (oob3(d, nb.x, nb.y, nb.z)) continue;
                const int64_t nc = lin3(d, nb.x, nb.y, nb.z);
                if (lab0[nc] != PS_REDUCED) continue;
                if (anyFin && anyEarlierFix(g, d, Fin, nb, ordC)) continue;   // already demoted when c is visited
                const int r = reg[nc];
                if (!seen) { seen = true; first = r; }
                else if (r != first) fix = true;
            }
        out = fix ? 1 : 0;
    }
    Fout[c] = out;
    if (out != Fin[c]) flags[0] = 1;   // changed
    if (out) flags[1] = 1;             // any fix in this sweep
}
__global__ void k_fix_apply(Grid g, int32_t* __restrict__ lab, int32_t* __restrict__ reg, const uint8_t* __restrict__ F) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    if (lab[c] != PS_REDUCED) return;
    const int3 q = unlin3(d, c);
    bool hit = false;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            int3 m = q;
            addc(m, a, dir ? 1 : -1);
            if (oob3(d, m.x, m.y, m.z)) continue;
            if (F[lin3(d, m.x, m.y, m.z)]) hit = true;
        }
    if (hit) { lab[c] = PS_ACTIVEFLUID; reg[c] = PS_UNASSIGNED; }
}

// ---- fixSmallReducedRegions, Classifier.cpp:1174-1313 -----------------------------------------
__global__ void k_bbox_init(int32_t* __restrict__ bb, int64_t R) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    bb[r * 6 + 0] = bb[r * 6 + 1] = bb[r * 6 + 2] = INT_MAX;
    bb[r * 6 + 3] = bb[r * 6 + 4] = bb[r * 6 + 5] = INT_MIN;
}
// boxes of the surviving regions, renumbered (fixSmallReducedRegions): out[remap[r]] = in[r] where keep[r]
__global__ void k_bbox_compact(const int32_t* __restrict__ in, const int32_t* __restrict__ keep, const int32_t* __restrict__ remap, int64_t R,
                               int32_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R || !keep[r]) return;
    for (int q = 0; q < 6; ++q) out[(int64_t)remap[r] * 6 + q] = in[r * 6 + q];
}
__global__ void k_bbox(Grid g, const int32_t* __restrict__ lab, const int32_t* __restrict__ reg, int32_t* bb) {
    const int3 d = g.dims(0);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    if (lab[c] != PS_REDUCED) return;
    const int r = reg[c];
    const int3 q = unlin3(d, c);
    // interior cells cannot move the box: skip the atomics unless some neighbour is outside the region
    bool edge = false;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            int3 nb = q;
            addc(nb, a, dir ? 1 : -1);
            if (oob3(d, nb.x, nb.y, nb.z)) { edge = true; continue; }
            const int64_t nc = lin3(d, nb.x, nb.y, nb.z);
            if (lab[nc] != PS_REDUCED || reg[nc] != r) edge = true;
        }
    if (!edge) return;
    // the box only ever grows: a (possibly stale) read that already covers this cell makes the atomic redundant — without the
    // test every surface cell of a tile issues six atomics on the same 24 bytes (2.0 ms at 256^3; with it 0.3 ms)
    int32_t* b = bb + (int64_t)r * 6;
    if (q.x < b[0]) atomicMin(&b[0], q.x);
    if (q.y < b[1]) atomicMin(&b[1], q.y);
    if (q.z < b[2]) atomicMin(&b[2], q.z);
    if (q.x > b[3]) atomicMax(&b[3], q.x);
    if (q.y > b[4]) atomicMax(&b[4], q.y);
    if (q.z > b[5]) atomicMax(&b[5], q.z);
}
__global__ void k_small_flags(const int32_t* __restrict__ bb, int64_t R, int32_t* __restrict__ keep) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    bool rem = false;
    for (int a = 0; a < 3; ++a) {
        const int lo = bb[r * 6 + a], hi = bb[r * 6 + 3 + a];
        if (hi < lo) { rem = true; continue; }   // emptied region (reference overflows here; restated as removed)
        if (hi == lo) rem = true;                 // :1236
        if (lo > hi - 3) rem = true;              // :1239
    }
    keep[r] = rem ? 0 : 1;
}
__global__ void k_small_apply(int32_t* __restrict__ lab, int32_t* __restrict__ reg, int64_t n,
                              const int32_t* __restrict__ keep, const int32_t* __restrict__ remap) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    if (lab[c] != PS_REDUCED) return;
    const int r = reg[c];
    if (!keep[r]) { lab[c] = PS_ACTIVEFLUID; reg[c] = PS_UNASSIGNED; }
    else reg[c] = remap[r];
}

// constructFaceAxisReducedIndicesPartial, Classifier.cpp:1473-1528
__global__ void k_face_reduced(Grid g, int axis, const int32_t* __restrict__ clab, const int32_t* __restrict__ creg,
                               int32_t* __restrict__ flab, int32_t* __restrict__ freg) {
    const int3 d = g.dims(1 + axis);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    const int3 cd = g.dims(0);
    int idx = PS_UNASSIGNED;
    int3 m = q;
    addc(m, axis, -1);
    if (!oob3(cd, q.x, q.y, q.z) && clab[lin3(cd, q.x, q.y, q.z)] == PS_REDUCED) idx = creg[lin3(cd, q.x, q.y, q.z)];
    else if (!oob3(cd, m.x, m.y, m.z) && clab[lin3(cd, m.x, m.y, m.z)] == PS_REDUCED) idx = creg[lin3(cd, m.x, m.y, m.z)];
    if (idx != PS_UNASSIGNED) { flab[c] = PS_REDUCED; freg[c] = idx; }
}

// constructEdgeAxisReducedIndicesPartial, Classifier.cpp:1534-1659
__global__ void k_edge_reduced(Grid g, int e, Set7<const int32_t> lab, Set7<const int32_t> reg,
                               int32_t* __restrict__ elab, int32_t* __restrict__ ereg) {
    const int3 d = g.dims(4 + e);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int3 q = unlin3(d, c);
    const int fa = e == 0 ? 1 : 0, fb = e == 2 ? 1 : 2;
    const int3 da = g.dims(1 + fa), db = g.dims(1 + fb);
    int3 a2 = q; addc(a2, fb, -1);
    int3 b2 = q; addc(b2, fa, -1);
    auto red = [&](int fax, const int3 fd, const int3 p) {
        return !oob3(fd, p.x, p.y, p.z) && lab.p[1 + fax][lin3(fd, p.x, p.y, p.z)] == PS_REDUCED;
    };
    const bool ra1 = red(fa, da, q), ra2 = red(fa, da, a2), rb1 = red(fb, db, q), rb2 = red(fb, db, b2);
    int label = PS_UNASSIGNED, idx = PS_UNASSIGNED;
    if (ra1 && ra2 && rb1 && rb2) {
        if (e == 0) idx = reg.p[2][lin3(g.dims(2), q.x, q.y - 1, q.z)];   // faceY(i,j-1,k), :1629
        else idx = reg.p[1 + fa][lin3(da, q.x, q.y, q.z)];
        label = PS_REDUCED;
    } else if (ra1) { idx = reg.p[1 + fa][lin3(da, q.x, q.y, q.z)]; label = PS_BOUNDARY; }
    else if (ra2) { idx = reg.p[1 + fa][lin3(da, a2.x, a2.y, a2.z)]; label = PS_BOUNDARY; }
    else if (rb1) { idx = reg.p[1 + fb][lin3(db, q.x, q.y, q.z)]; label = PS_BOUNDARY; }
    else if (rb2) { idx = reg.p[1 + fb][lin3(db, b2.x, b2.y, b2.z)]; label = PS_BOUNDARY; }
    if (idx != PS_UNASSIGNED) { elab[c] = label; ereg[c] = idx; }
}

// buildValidFaces, Classifier.cpp:4-54
__global__ void k_valid(const int32_t* __restrict__ lab, int64_t n, float* __restrict__ valid) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int l = lab[c];
    valid[c] = (l == PS_UNSOLVED || l == PS_UNASSIGNED) ? 0.f : 1.f;
}

// ---- scans --------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;                      // per thread
constexpr int SCAN_TILE = BS * SCAN_ITEMS;         // per block
static_assert(SCAN_TILE == PS_SCAN_TILE, "ps_context.hpp: PS_SCAN_TILE");

__device__ inline int blockExclusiveScan(int v, int* total) {
    __shared__ int waveSums[BS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) waveSums[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) {
        if (i < w) base += waveSums[i];
        tot += waveSums[i];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// the voxel after q in the traversal order (its position is t): the x-neighbour, the next row / plane of the 16^3 tile, or — leaving the
// tile — a full decode.  (One full decode per position, six 64-bit divisions each, was most of the 1.9 ms of these kernels at 256^3.)
__device__ inline void orderAdvance(const int3 d, int order, int64_t t, int3& q) {
    if (order == PS_ORDER_LINEAR) { if (++q.x == d.x) { q.x = 0; if (++q.y == d.y) { q.y = 0; ++q.z; } } return; }
    const int T = 16;
    const int x0 = (q.x / T) * T, y0 = (q.y / T) * T;
    if (q.x + 1 < min(d.x, x0 + T)) { ++q.x; return; }
    if (q.y + 1 < min(d.y, y0 + T)) { q.x = x0; ++q.y; return; }
    if (q.z + 1 < min(d.z, (q.z / T) * T + T)) { q.x = x0; q.y = y0; ++q.z; return; }
    q = orderToIjk(d, order, t);
}
// flag of the traversal position t (voxel q) for the ordered index assignment
__device__ inline int orderedFlag(const int3 d, int mode, const int32_t* __restrict__ src, int64_t t, const int3 q, int64_t* linOut) {
    const int64_t c = lin3(d, q.x, q.y, q.z);
    *linOut = c;
    if (mode == 0) return isActiveL(src[c]) ? 1 : 0;   // serialAssignFieldIndices, Classifier.cpp:1764
    return src[c] == (int32_t)t ? 1 : 0;               // component root (cc label == own order index)
}
__global__ void k_ordered_count(Grid g, int s, int mode, const int32_t* __restrict__ src, int32_t* __restrict__ blockSums) {
    const int3 d = g.dims(s);
    const int64_t n = (int64_t)d.x * d.y * d.z;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int cnt = 0;
    int3 q = base < n ? orderToIjk(d, g.order, base) : make_int3(0, 0, 0);
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t t = base + i;
        int64_t c;
        if (t < n) {
            if (i) orderAdvance(d, g.order, t, q);
            cnt += orderedFlag(d, mode, src, t, q, &c);
        }
    }
    int tot;
    blockExclusiveScan(cnt, &tot);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}
__global__ void k_ordered_assign(Grid g, int s, int mode, const int32_t* __restrict__ src, const int32_t* __restrict__ blockOffs,
                                 int32_t* __restrict__ out) {
    const int3 d = g.dims(s);
    const int64_t n = (int64_t)d.x * d.y * d.z;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int fl[SCAN_ITEMS];
    int64_t cc[SCAN_ITEMS];
    int cnt = 0;
    int3 q = base < n ? orderToIjk(d, g.order, base) : make_int3(0, 0, 0);
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t t = base + i;
        fl[i] = 0;
        cc[i] = 0;
        if (t < n) {
            if (i) orderAdvance(d, g.order, t, q);
            fl[i] = orderedFlag(d, mode, src, t, q, &cc[i]);
        }
        cnt += fl[i];
    }
    int tot;
    int off = blockExclusiveScan(cnt, &tot) + blockOffs[blockIdx.x];
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (fl[i]) out[cc[i]] = off++;
}
// single-block exclusive scan of a (short) int array, carry across tiles of 1024; total -> *total
__global__ void k_scan_single(int32_t* __restrict__ a, int64_t n, int32_t* __restrict__ total) {
    __shared__ int sm[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = i < n ? a[i] : 0;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = 0;
            if ((int)threadIdx.x >= o) t = sm[threadIdx.x - o];
            __syncthreads();
            sm[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = sm[threadIdx.x];
        if (i < n) a[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void k_scan_blocksum(const int32_t* __restrict__ a, int64_t n, int32_t* __restrict__ blockSums) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int cnt = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) cnt += a[base + i];
    int tot;
    blockExclusiveScan(cnt, &tot);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}
__global__ void k_scan_apply(int32_t* __restrict__ a, int64_t n, const int32_t* __restrict__ blockOffs) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int cnt = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? a[base + i] : 0; cnt += v[i]; }
    int tot;
    int off = blockExclusiveScan(cnt, &tot) + blockOffs[blockIdx.x];
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) { a[base + i] = off; off += v[i]; }
}


// ---- internal (solver) numbering: interleaved by 16^3 spatial block ------------------------------------
// Virtual sequence: for block b (lattice order), for group g: the voxels of grid sample[g] that fall in
// block b, x-fastest.  An active voxel takes weight[g] consecutive indices.
struct ILDesc {
    int ngroups;
    int sample[8];             // the sample grid a group lives on (the four cell groups p, txx, tyy, tzz share grid 0)
    int weight[8];
    int LBx, LBy, LBz;
    int SBx, SBy, SBz;         // super-block of SBx x SBy x SBz lattice blocks: consecutive in the sequence (L2 working set of one XCD)
    int NSx, NSy;              // super-blocks per axis (x, y)
    int ox, oy, oz;            // lattice origin offset: block (bx,by,bz) covers voxels [16 b - o, 16 b - o + 16)
    int64_t total;
    const int64_t* segStart;   // LB^3 * ngroups + 1
    int nseg;
    Own own;
    int ownFilter;             // 1: only owned voxels are numbered (face rows); 0: every active voxel (DOFs)
    int planeMajor;
    const int32_t* blockMap;   // sequence position -> lattice block (a decomposition numbers its owned blocks first); null: lattice order
    int64_t probe[2];          // virtual positions whose running prefix is reported in counters[10], [11]
};
__device__ inline bool ilFlag(const ILDesc& D, const Grid& g, const Set7<const int32_t>& lab, int grp, int64_t c) {
    const int s = D.sample[grp];
    if (!isActiveL(lab.p[s][c])) return false;
    if (D.ownFilter) {
        const int3 q = unlin3(g.dims(s), c);
        if (!D.own.sample(s, q.x, q.y, q.z)) return false;
    }
    return true;
}
__device__ inline bool ilDecode(const ILDesc& D, const Grid& g, int64_t u, int* grp, int64_t* lin) {
    if (u >= D.total) return false;
    // position = ((block * 4096 + local voxel) * ngroups + group): the groups of ONE voxel index are adjacent, so the
    // 3 face rows (resp. the 7 DOFs) hanging off a cell are contiguous and a row block re-uses the lines it gathers
    const int64_t per = (int64_t)4096 * D.ngroups;
    const int bs = (int)(u / per);
    const int b = D.blockMap ? D.blockMap[bs] : bs;
    const int rem = (int)(u - (int64_t)bs * per);
    int v, gg;
    if (D.planeMajor) {   // per k-plane of the block: group 0's 256 voxels, then group 1's, ... (type-major inside a plane)
        const int pl = rem / (256 * D.ngroups), r2 = rem - pl * 256 * D.ngroups;
        gg = r2 >> 8; v = (pl << 8) | (r2 & 255);
    } else { v = rem / D.ngroups; gg = rem - v * D.ngroups; }
    const int sbv = D.SBx * D.SBy * D.SBz;
    const int sb = b / sbv, wi = b - sb * sbv;
    const int bx = (sb % D.NSx) * D.SBx + wi % D.SBx, by = ((sb / D.NSx) % D.NSy) * D.SBy + (wi / D.SBx) % D.SBy,
              bz = (sb / (D.NSx * D.NSy)) * D.SBz + wi / (D.SBx * D.SBy);
    const int i = 16 * bx + (v & 15) - D.ox, j = 16 * by + ((v >> 4) & 15) - D.oy, k = 16 * bz + (v >> 8) - D.oz;
    const int3 d = g.dims(D.sample[gg]);
    if (i < 0 || j < 0 || k < 0 || i >= d.x || j >= d.y || k >= d.z) return false;
    *grp = gg;
    *lin = lin3(d, i, j, k);
    return true;
}
// The SCAN_ITEMS = 8 consecutive positions a thread takes, decoded once (kind-major order only): they are eight x-neighbours of one kind
// in one row of a lattice block — group, linear index of the first, how many of them lie inside the grid.  (One decode per position was
// a third of the numbering's 2.6 ms at 256^3: 64-bit divisions.)
__device__ inline void ilDecode8(const ILDesc& D, const Grid& g, int64_t base, int* grp, int64_t* lin0, int* nIn) {
    *nIn = 0; *grp = 0; *lin0 = 0;
    if (base >= D.total) return;
    const unsigned per = 4096u * (unsigned)D.ngroups;
    const int bs = (int)(base / per);
    const unsigned rem = (unsigned)(base - (int64_t)bs * per);
    const int b = D.blockMap ? D.blockMap[bs] : bs;
    const unsigned pl = rem / (256u * (unsigned)D.ngroups), r2 = rem - pl * 256u * (unsigned)D.ngroups;
    const int gg = (int)(r2 >> 8), v = (int)((pl << 8) | (r2 & 255u));
    const int sbv = D.SBx * D.SBy * D.SBz;
    const int sb = b / sbv, wi = b - sb * sbv;
    const int bx = (sb % D.NSx) * D.SBx + wi % D.SBx, by = ((sb / D.NSx) % D.NSy) * D.SBy + (wi / D.SBx) % D.SBy,
              bz = (sb / (D.NSx * D.NSy)) * D.SBz + wi / (D.SBx * D.SBy);
    const int i = 16 * bx + (v & 15) - D.ox, j = 16 * by + ((v >> 4) & 15) - D.oy, k = 16 * bz + (v >> 8) - D.oz;
    const int3 d = g.dims(D.sample[gg]);
    *grp = gg;
    if (i < 0 || j < 0 || k < 0 || j >= d.y || k >= d.z || i >= d.x) return;   // (i >= 0 and a multiple of 8 here: the origin offsets are 0)
    *lin0 = lin3(d, i, j, k);
    *nIn = min(SCAN_ITEMS, d.x - i);
}
__global__ void k_il_count(ILDesc D, Grid g, Set7<const int32_t> lab, int32_t* __restrict__ blockSums) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int cnt = 0;
    if (D.planeMajor) {
        int grp, nIn; int64_t lin0;
        ilDecode8(D, g, base, &grp, &lin0, &nIn);
        for (int i = 0; i < nIn; ++i) if (ilFlag(D, g, lab, grp, lin0 + i)) cnt += D.weight[grp];
    } else {
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            int grp; int64_t c;
            if (ilDecode(D, g, base + i, &grp, &c) && ilFlag(D, g, lab, grp, c)) cnt += D.weight[grp];
        }
    }
    int tot;
    blockExclusiveScan(cnt, &tot);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}
__global__ void k_il_assign(ILDesc D, Grid g, Set7<const int32_t> lab, const int32_t* __restrict__ blockOffs, Set8<int32_t> outs,
                            int32_t* __restrict__ probeOut, int32_t* __restrict__ blockStart) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int w[SCAN_ITEMS], gr[SCAN_ITEMS];
    int64_t cc[SCAN_ITEMS];
    int cnt = 0;
    if (D.planeMajor) {
        int grp, nIn; int64_t lin0;
        ilDecode8(D, g, base, &grp, &lin0, &nIn);
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            gr[i] = grp; cc[i] = lin0 + i;
            w[i] = (i < nIn && ilFlag(D, g, lab, grp, lin0 + i)) ? D.weight[grp] : 0;
            cnt += w[i];
        }
    } else {
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            w[i] = 0; gr[i] = 0; cc[i] = 0;
            if (ilDecode(D, g, base + i, &gr[i], &cc[i]) && ilFlag(D, g, lab, gr[i], cc[i])) w[i] = D.weight[gr[i]];
            cnt += w[i];
        }
    }
    int tot;
    int off = blockExclusiveScan(cnt, &tot) + blockOffs[blockIdx.x];
    // (block starts and the probes are multiples of 4096 ngroups, a thread's first position is a multiple of 8: only that one can be either)
    if (base == D.probe[0]) probeOut[0] = off;
    if (base == D.probe[1]) probeOut[1] = off;
    if (base < D.total && base % ((int64_t)4096 * D.ngroups) == 0) blockStart[base / ((int64_t)4096 * D.ngroups)] = off;
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (w[i]) { outs.p[gr[i]][cc[i]] = off; off += w[i]; }
}
// are all volume fractions of the array multiples of 1/8 (the 2x2x2 sampler's)?  Then every stencil value is an int8 code times
// 1 / (64 dx) (ps_blocks.hip: encodeVal) and the SpMVs will run the row-per-lane kernels on the coded stream
__global__ void k_dyadic_check(const float* __restrict__ w, int64_t n, int32_t* __restrict__ fail) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float q = w[i] * 8.f;
        bad |= !(q == rintf(q) && q >= 0.f && q <= 8.f);
    }
    if (bad) *fail = 1;
}
// permSys[reference index] = internal index  (reference layout: Solver.h:586-606)
__global__ void k_perm_cells(Grid g, const int32_t* __restrict__ act, const int32_t* __restrict__ sys, const int32_t* __restrict__ sxx,
                             const int32_t* __restrict__ syy, const int32_t* __restrict__ szz, int64_t nP, int64_t nC, int32_t* __restrict__ perm) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.count(0)) return;
    const int q = act[c];
    if (q < 0) return;
    perm[q] = sys[c]; perm[nP + q] = sxx[c]; perm[nP + nC + q] = syy[c]; perm[nP + 2 * nC + q] = szz[c];
}
__global__ void k_perm_simple(const int32_t* __restrict__ act, const int32_t* __restrict__ internal, int64_t n, int64_t refOffset,
                              int32_t* __restrict__ perm) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int q = act[c];
    if (q >= 0) perm[refOffset + q] = internal[c];
}

template <class T>
Set7<const T> cset(DevBuf<T>* b) {
    Set7<const T> s;
    for (int i = 0; i < 7; ++i) s.p[i] = b[i].p;
    return s;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
int32_t ps_context::readCounter(int idx) {
    // into page-locked memory (r05): a device-to-host copy of 4 bytes into a stack variable goes through the runtime's staging buffer — about half of the
    // ~30 us a count-only round trip of the setup costs (36 of them per setup: profiles/r05_setup_timeline_coil128.txt)
    if (!pinnedCounters && hipHostMalloc((void**)&pinnedCounters, 64 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) pinnedCounters = nullptr;
    if (pinnedCounters) {
        HIP_CHECK(hipMemcpyAsync(pinnedCounters + idx, counters.p + idx, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return pinnedCounters[idx];
    }
    int32_t v = 0;
    HIP_CHECK(hipMemcpyAsync(&v, counters.p + idx, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return v;
}
// several counters in one round trip (synchronises the stream)
void ps_context::fetchCounters(int idx, int n, int32_t* out) {
    if (!pinnedCounters && hipHostMalloc((void**)&pinnedCounters, 64 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) pinnedCounters = nullptr;
    int32_t* dst = pinnedCounters ? pinnedCounters + idx : out;
    HIP_CHECK(hipMemcpyAsync(dst, counters.p + idx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (pinnedCounters) for (int q = 0; q < n; ++q) out[q] = pinnedCounters[idx + q];
}
void ps_context::zeroCounters() { HIP_CHECK(hipMemsetAsync(counters.p, 0, 64 * sizeof(int32_t), stream)); }

// In-place exclusive scan of a device int32 array; returns the total (host).
// counterSlot >= 0: the total goes to counters[counterSlot] and the call returns -1 without synchronising (as orderedIndexAssign).  Consecutive
// unsynchronised scans share scanBlock: it must already hold gridFor(largest n, SCAN_TILE) entries (a reallocation would free memory a
// queued kernel uses) — the caller allocates it first.
int64_t ps_context::exclusiveScanI32(int32_t* data, int64_t n, int counterSlot) {
    if (n <= 0) { if (counterSlot >= 0) HIP_CHECK(hipMemsetAsync(counters.p + counterSlot, 0, sizeof(int32_t), stream)); return counterSlot >= 0 ? -1 : 0; }
    const int slot = counterSlot >= 0 ? counterSlot : 8;
    const int nb = gridFor(n, SCAN_TILE);
    if (nb == 1) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, data, n, counters.p + slot);
        return counterSlot >= 0 ? -1 : readCounter(8);
    }
    scanBlock.alloc((size_t)nb);
    hipLaunchKernelGGL(k_scan_blocksum, dim3(nb), dim3(BS), 0, stream, data, n, scanBlock.p);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, scanBlock.p, (int64_t)nb, counters.p + slot);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(BS), 0, stream, data, n, scanBlock.p);
    return counterSlot >= 0 ? -1 : readCounter(8);
}

// serialAssignFieldIndices (Classifier.cpp:1738-1770) as a two-level scan over traversal positions.
// counterSlot >= 0: the total goes to counters[counterSlot] and the call returns -1 WITHOUT synchronising (the caller reads several
// totals with one round trip: every host synchronisation of the setup costs ~30 us of idle device)
int32_t ps_context::orderedIndexAssign(int s, int mode, DevBuf<int32_t>& out, int counterSlot) {
    const int64_t n = g.count(s);
    const int nb = gridFor(n, SCAN_TILE);
    scanBlock.alloc((size_t)nb);           // (one buffer for consecutive calls: the launches are ordered on the stream; sized for the largest grid on first use)
    const int32_t* src = mode == 0 ? labels[s].p : cellScratch[0].p;
    const int slot = counterSlot >= 0 ? counterSlot : 8;
    hipLaunchKernelGGL(k_ordered_count, dim3(nb), dim3(BS), 0, stream, g, s, mode, src, scanBlock.p);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, scanBlock.p, (int64_t)nb, counters.p + slot);
    hipLaunchKernelGGL(k_ordered_assign, dim3(nb), dim3(BS), 0, stream, g, s, mode, src, scanBlock.p, out.p);
    return counterSlot >= 0 ? -1 : readCounter(8);
}

// Solver.cpp:238-289
void ps_context::buildIntegrationWeightsAlt() {
    if (haveInputWeights) return;   // uploaded by the shim (HDK's own sampler output)
    const int64_t n = (int64_t)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
    Set7<float> lw, fw;
    for (int s = 0; s < 7; ++s) { lw.p[s] = liquidW[s].p; fw.p[s] = fluidW[s].p; }
    hipLaunchKernelGGL(k_sdf_weights7, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, (const float*)surface.p, 0, lw);
    hipLaunchKernelGGL(k_sdf_weights7, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, (const float*)collision.p, P.negateCollision ? 1 : 0, fw);
}

void ps_context::classifyCells() {
    const int64_t n = g.count(0);
    hipLaunchKernelGGL(k_classify_cells, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, cset(liquidW), cset(fluidW), labels[0].p);
}

// Classifier.cpp:179-190 (constructAirBoundaryLayer :291-362, constructSolidBoundaryLayer :510-571, constructTiles)
void ps_context::constructReducedRegions() {
    const int64_t n = g.count(0);
    const dim3 gr(gridFor(n, BS)), bl(BS);
    int32_t* mark = cellScratch[0].p;
    const int L = P.activeLiquidBoundaryLayerSize, Sl = P.activeSolidBoundaryLayerSize;
    // air: layers 0..L-2 are made ACTIVE (loop bound `layer < L-1`, :328; next built only if `layer < L-2`, :355)
    if (L - 1 >= 1) {
        hipLaunchKernelGGL(k_air_layer0, gr, bl, 0, stream, g, cset(liquidW), labels[0].p, mark, 1);
        for (int layer = 0; layer < L - 2; ++layer)
            hipLaunchKernelGGL(k_air_next, gr, bl, 0, stream, g, cset(liquidW), labels[0].p, mark, layer + 1);
    }
    // solid: layers 0..S-1 (:526), next built if layer < S-1 (:563)
    if (Sl >= 1) {
        int32_t* labIn = cellScratch[1].p;   // snapshot: layer 0 reads labels other threads are rewriting
        HIP_CHECK(hipMemcpyAsync(labIn, labels[0].p, n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_solid_layer0, gr, bl, 0, stream, g, labels[0].p, mark, labIn);
        for (int layer = 0; layer < Sl - 1; ++layer)
            hipLaunchKernelGGL(k_solid_next, gr, bl, 0, stream, g, cset(liquidW), labels[0].p, mark, layer + 1);
    }
    hipLaunchKernelGGL(k_tiles_and_relabel, gr, bl, 0, stream, g, labels[0].p, P.doTile ? 1 : 0, P.tileSize, P.tilePadding);
}

void ps_context::constructOnlyActiveRegions() {
    const int64_t n = g.count(0);
    hipLaunchKernelGGL(k_relabel, dim3(gridFor(n, BS)), dim3(BS), 0, stream, labels[0].p, n, (int)PS_GENERICFLUID, (int)PS_ACTIVEFLUID);
}

void ps_context::classifyFaces() {
    for (int a = 0; a < 3; ++a) {
        const int64_t n = g.count(1 + a);
        hipLaunchKernelGGL(k_classify_faces, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, a, cset(liquidW), cset(fluidW), labels[1 + a].p);
    }
}
void ps_context::classifyEdges() {
    for (int e = 0; e < 3; ++e) {
        const int64_t n = g.count(4 + e);
        hipLaunchKernelGGL(k_classify_edges, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, e, cset(liquidW), cset(fluidW), labels[4 + e].p);
    }
}

// Classifier.cpp:217-239
// part 0: connected components + fixReducedRegionBoundaries; part 1: fixSmallReducedRegions.  Between the two the ranks of a
// decomposition replace the labels of their halo blocks by the owners' (Dist::exchangeLabels): the boundary fix looks one cell beyond a
// region's tile, i.e. beyond the halo block, and what it demotes there decides whether a thin region survives part 1.
void ps_context::constructCenterReducedIndices(int part) {
    const int64_t n = g.count(0);
    const dim3 gr(gridFor(n, BS)), bl(BS);
    if (part == 0) {
    int32_t* cc = cellScratch[0].p;
    // connected components
    uint8_t* link = (uint8_t*)cellScratch[2].p;
    hipLaunchKernelGGL(k_cc_init, gr, bl, 0, stream, g, labels[0].p, cset(liquidW), cc, link);
    // tile-local components first, in LDS (5 bytes per cell of a tile, within the 64 KB a launch gets without asking: tiles up to 23^3)
    static const bool localOff = PS_ENV("PS_CC_LOCAL") && atoi(PS_ENV("PS_CC_LOCAL")) == 0;   // A/B: the grid-wide passes alone
    const int B = P.tileSize;
    const bool local = !localOff && P.doTile && P.tilePadding >= 1 && B >= 4 && (size_t)B * B * B * 5 <= 64 * 1024;
    if (local)
        hipLaunchKernelGGL(k_cc_local, dim3((g.nx + B - 1) / B, (g.ny + B - 1) / B, (g.nz + B - 1) / B), dim3(256), (size_t)B * B * B * 5, stream, g, B, (const uint8_t*)link, cc);
    for (int guard = 0; guard < 100000; ++guard) {
        zeroCounters();
        for (int it = 0; it < ((local && guard == 0) ? 1 : 4); ++it)
            hipLaunchKernelGGL(k_cc_step, gr, bl, 0, stream, g, (const uint8_t*)link, cc, counters.p);
        if (!readCounter(0)) break;
    }
    int32_t* rootRank = cellScratch[1].p;
    {
        DevBuf<int32_t> view;   // non-owning alias for orderedIndexAssign's output
        view.p = rootRank; view.n = (size_t)n;
        regionCount = orderedIndexAssign(0, 1, view);
        view.p = nullptr; view.n = 0;
    }
    hipLaunchKernelGGL(k_cc_assign, gr, bl, 0, stream, g, labels[0].p, cc, rootRank, reducedIdx[0].p);
    // a rank keeps the components as they were before the fix: a cell the owner did NOT demote gets its region back (cellScratch[2]:
    // the link bits are no longer needed, the fix works in scratch 0 and 1)
    if (slabEnabled) HIP_CHECK(hipMemcpyAsync(cellScratch[2].p, reducedIdx[0].p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));

    // fixReducedRegionBoundaries (:1073-1172)
    if (regionCount > 1) {
        uint8_t* F0 = (uint8_t*)cellScratch[0].p;
        uint8_t* F1 = (uint8_t*)cellScratch[1].p;
        for (int sweep = 0; sweep < 100000; ++sweep) {
            HIP_CHECK(hipMemsetAsync(F0, 0, (size_t)n, stream));
            int anyF = 0;
            bool applied = false;
            for (int it = 0; it < 100000; ++it) {
                zeroCounters();
                hipLaunchKernelGGL(k_fix_eval, gr, bl, 0, stream, g, labels[0].p, reducedIdx[0].p, F0, F1, anyF, counters.p);
                int32_t fl[2];
                fetchCounters(0, 2, fl);
                std::swap(F0, F1);
                anyF = fl[1];
                if (!fl[0]) break;
            }
            if (anyF) {
                hipLaunchKernelGGL(k_fix_apply, gr, bl, 0, stream, g, labels[0].p, reducedIdx[0].p, F0);
                applied = true;
            }
            if (!applied) break;
        }
    }
    return;
    }

    // fixSmallReducedRegions (:1174-1262)
    if (regionCount > 0) {
        const int64_t R = regionCount;
        bbox.alloc((size_t)R * 6);
        DevBuf<int32_t>& keep = scrKeep; DevBuf<int32_t>& remap = scrRemap;
        keep.alloc((size_t)R);
        remap.alloc((size_t)R);
        hipLaunchKernelGGL(k_bbox_init, dim3(gridFor(R, BS)), bl, 0, stream, bbox.p, R);
        hipLaunchKernelGGL(k_bbox, gr, bl, 0, stream, g, labels[0].p, reducedIdx[0].p, bbox.p);
        hipLaunchKernelGGL(k_small_flags, dim3(gridFor(R, BS)), bl, 0, stream, bbox.p, R, keep.p);
        HIP_CHECK(hipMemcpyAsync(remap.p, keep.p, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        const int64_t newR = exclusiveScanI32(remap.p, R);
        if (newR < R) {
            hipLaunchKernelGGL(k_small_apply, gr, bl, 0, stream, labels[0].p, reducedIdx[0].p, n, keep.p, remap.p);
            DevBuf<int32_t> compact;
            compact.alloc((size_t)std::max<int64_t>(newR, 1) * 6);
            hipLaunchKernelGGL(k_bbox_compact, dim3(gridFor(R, BS)), bl, 0, stream, bbox.p, keep.p, remap.p, R, compact.p);
            HIP_CHECK(hipMemcpyAsync(bbox.p, compact.p, (size_t)newR * 6 * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
            compact.free();   // deferred (a DevBuf going out of scope calls hipFree: a device-wide synchronisation inside a step)
            regionCount = newR;
        }
        bboxValid = true;   // the boxes of the final regions are already on the device (computeRegionBoxes only downloads them)
        HIP_CHECK(hipStreamSynchronize(stream));
    }
}

void ps_context::computeRegionBoxes() {
    const int64_t R = regionCount;
    hbbox.assign((size_t)R * 6, 0);
    if (R == 0) return;
    const int64_t n = g.count(0);
    if (!bboxValid) {
        bbox.alloc((size_t)R * 6);
        hipLaunchKernelGGL(k_bbox_init, dim3(gridFor(R, BS)), dim3(BS), 0, stream, bbox.p, R);
        hipLaunchKernelGGL(k_bbox, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, labels[0].p, reducedIdx[0].p, bbox.p);
    }
    HIP_CHECK(hipMemcpyAsync(hbbox.data(), bbox.p, (size_t)R * 6 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

void ps_context::constructFacesReducedIndices() {
    for (int a = 0; a < 3; ++a) {
        const int64_t n = g.count(1 + a);
        hipLaunchKernelGGL(k_face_reduced, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, a, labels[0].p, reducedIdx[0].p,
                           labels[1 + a].p, reducedIdx[1 + a].p);
    }
}
void ps_context::constructEdgesReducedIndices() {
    for (int e = 0; e < 3; ++e) {
        const int64_t n = g.count(4 + e);
        hipLaunchKernelGGL(k_edge_reduced, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, e, cset(labels), cset(reducedIdx),
                           labels[4 + e].p, reducedIdx[4 + e].p);
    }
}

// Classifier.cpp:257-284
void ps_context::constructActiveIndices() {
    {   // the scan's block buffer must fit the largest of the seven grids BEFORE the first launch: a reallocation between the
        // unsynchronised calls would free memory a queued kernel still uses
        int64_t most = 0;
        for (int s = 0; s < 7; ++s) most = std::max<int64_t>(most, g.count(s));
        scanBlock.alloc((size_t)gridFor(most, SCAN_TILE));
    }
    for (int s = 0; s < 7; ++s) {
        const int64_t n = g.count(s);
        hipLaunchKernelGGL(k_relabel, dim3(gridFor(n, BS)), dim3(BS), 0, stream, labels[s].p, n, (int)PS_GENERICFLUID, (int)PS_ACTIVEFLUID);
        (void)orderedIndexAssign(s, 0, activeIdx[s], 48 + s);      // totals -> counters[48 .. 54], one round trip for the seven
    }
    int32_t cnt[7];
    fetchCounters(48, 7, cnt);
    nCenter = cnt[0];
    for (int a = 0; a < 3; ++a) { nFace[a] = cnt[1 + a]; nEdge[a] = cnt[4 + a]; }
}


int64_t ps_context::interleavedIndexAssign(int ngroups, const int* samples, const int* weights, int32_t* const* outs) {
    return interleavedIndexAssignEx(ngroups, samples, weights, outs, false, nullptr);
}
int64_t ps_context::interleavedIndexAssignEx(int ngroups, const int* samples, const int* weights, int32_t* const* outs, bool ownFilter,
                                             int64_t* ownedRange) {
    ILDesc D;
    D.ngroups = ngroups;
    D.own = own();
    D.ownFilter = ownFilter ? 1 : 0;
    D.planeMajor = ownedRange ? (ilPlaneMajor & 1) : ((ilPlaneMajor >> 1) & 1);   // bit 0: DOFs, bit 1: face rows
    D.probe[0] = D.probe[1] = -1;
    for (int q = 0; q < 8; ++q) { D.sample[q] = q < ngroups ? samples[q] : 0; D.weight[q] = q < ngroups ? weights[q] : 0; }
    D.ox = D.oy = D.oz = 0;
    D.LBx = (g.nx + 1 + D.ox + 15) / 16; D.LBy = (g.ny + 1 + D.oy + 15) / 16; D.LBz = (g.nz + 1 + D.oz + 15) / 16;
    D.SBx = D.SBy = D.SBz = 1;
    D.NSx = (D.LBx + D.SBx - 1) / D.SBx; D.NSy = (D.LBy + D.SBy - 1) / D.SBy;
    const int NSz = (D.LBz + D.SBz - 1) / D.SBz;
    const int64_t per = (int64_t)4096 * ngroups;
    const int64_t run = (int64_t)D.NSx * D.NSy * NSz * D.SBx * D.SBy * D.SBz * per;
    D.total = run;
    D.nseg = 0;
    D.segStart = nullptr;
    // A decomposition numbers the lattice blocks this rank owns first (both sequences, in lattice order within each class): the owned
    // DOFs are then one contiguous index range [0, owned).  A block is owned iff along every axis it lies in [lo / 16, hi / 16) — or
    // beyond, where there is no upper neighbour (the last plane of the domain).  Cuts are multiples of 16: blocks are never split.
    D.blockMap = nullptr;
    if (slabEnabled) {
        const int nB = D.LBx * D.LBy * D.LBz;
        if (blockMapFor != nB || blockMapOwned < 0) {
            std::vector<int32_t> own_, rest;
            const int LB[3] = {D.LBx, D.LBy, D.LBz};
            for (int b = 0; b < nB; ++b) {
                const int bc[3] = {b % LB[0], (b / LB[0]) % LB[1], b / (LB[0] * LB[1])};
                bool mine = true;
                for (int a = 0; a < 3; ++a) mine = mine && bc[a] >= brick.lo[a] / 16 && (bc[a] < brick.hi[a] / 16 || !brick.hasUpper[a]);
                (mine ? own_ : rest).push_back(b);
            }
            blockMapOwned = (int)own_.size(); blockMapFor = nB;
            own_.insert(own_.end(), rest.begin(), rest.end());
            blockMap.alloc((size_t)nB);
            HIP_CHECK(hipMemcpyAsync(blockMap.p, own_.data(), (size_t)nB * 4, hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
        D.blockMap = blockMap.p;
    }
    if (ownedRange) {
        D.probe[0] = 0;
        D.probe[1] = slabEnabled ? (int64_t)blockMapOwned * per : run;
    }
    const int nbk = gridFor(run, SCAN_TILE);
    scanBlock.alloc((size_t)nbk);
    Set8<int32_t> o;
    for (int q = 0; q < 8; ++q) o.p[q] = q < ngroups ? outs[q] : nullptr;
    hipLaunchKernelGGL(k_il_count, dim3(nbk), dim3(BS), 0, stream, D, g, cset(labels), scanBlock.p);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, scanBlock.p, (int64_t)nbk, counters.p + 8);
    const int nBlocks = (int)(run / per);
    DevBuf<int32_t>& bstart = scrStart4;      // scratch kept with the context (also the stream build's)
    bstart.alloc((size_t)nBlocks + 1);
    hipLaunchKernelGGL(k_il_assign, dim3(nbk), dim3(BS), 0, stream, D, g, cset(labels), scanBlock.p, o, counters.p + 10, bstart.p);
    int32_t cnt[4] = {0, 0, 0, 0};                      // counters[8 .. 11]: the total, -, the two probes — one round trip with the block starts
    std::vector<int32_t>& hs = ownedRange ? blockStartSys : blockStartRow;
    hs.assign((size_t)nBlocks + 1, 0);
    HIP_CHECK(hipMemcpyAsync(hs.data(), bstart.p, (size_t)nBlocks * 4, hipMemcpyDeviceToHost, stream));
    fetchCounters(8, 4, cnt);                            // (synchronises: the block starts have landed too)
    const int64_t total = cnt[0];
    hs[(size_t)nBlocks] = (int32_t)total;
    ilBlocks = nBlocks;
    if (ownedRange) {
        ownedRange[0] = D.probe[0] >= run ? total : cnt[2];
        ownedRange[1] = D.probe[1] >= run ? total : cnt[3];
    }
    return total;
}

// Internal numbering of system DOFs and active face rows (see ps_context.hpp).  With PS_ORDER_LINEAR the
// same interleaving is used: it is an internal layout, the reference numbering stays in activeIdx[].
void ps_context::buildInternalNumbering() {
    {
        // order inside a lattice block: 0 = voxel by voxel (the kinds of DOF / the three faces hanging off one voxel adjacent), 3 = kind
        // by kind inside every k-plane of the block (all pressures of the plane, all txx, ... / all X faces, all Y, all Z): what the
        // row-per-lane SpMV kernels want — the four lanes of a quad then gather four CONSECUTIVE entries of one kind
        // (profiles/r03_spmv_issue.md).  PS_IL = 0..3 overrides (bit 0: DOFs, bit 1: face rows).
        // Default: kind-major when the row-per-lane kernels will run — the stencil values must be codable, which the weights decide:
        // face fluid weights x cell / edge liquid weights, all multiples of 1/8 (checked here, the fill kernels verify every entry);
        // systems with other weights stream fp64 values through the 4-entries-per-lane kernels, which want the voxel-major order.
        static const bool noEll = PS_ENV("PS_NO_ELL") && atoi(PS_ENV("PS_NO_ELL")) != 0;
        static const bool forceF64 = PS_ENV("PS_FORCE_FP64_VALUES") && atoi(PS_ENV("PS_FORCE_FP64_VALUES")) != 0;
        static const bool col32 = PS_ENV("PS_COL32") && atoi(PS_ENV("PS_COL32")) != 0;
        static const bool oneShot = PS_ENV("PS_PIPE_GRID") && atoi(PS_ENV("PS_PIPE_GRID")) == 0;
        int mode = 0;
        if (!noEll && !forceF64 && !col32 && !oneShot) {
            HIP_CHECK(hipMemsetAsync(counters.p + 40, 0, sizeof(int32_t), stream));
            for (int s2 : {1, 2, 3}) hipLaunchKernelGGL(k_dyadic_check, dim3(1024), dim3(BS), 0, stream, (const float*)fluidW[s2].p, g.count(s2), counters.p + 40);
            for (int s2 : {0, 4, 5, 6}) hipLaunchKernelGGL(k_dyadic_check, dim3(1024), dim3(BS), 0, stream, (const float*)liquidW[s2].p, g.count(s2), counters.p + 40);
            if (readCounter(40) == 0) mode = 3;
        }
        const char* e = PS_ENV("PS_IL");
        ilPlaneMajor = e ? (atoi(e) & 3) : mode;
    }
    const int64_t nC = nCenter, nPq = nCenter;
    const int64_t nSys = 4 * nCenter + nEdge[0] + nEdge[1] + nEdge[2];
    const int64_t nAct = nFace[0] + nFace[1] + nFace[2];
    {   // the ten index arrays of the numbering start at -1: ONE launch (r06: ten before)
        FillMany F;
        int q = 0;
        int64_t most = 1;
        auto add = [&](int32_t* ptr, int64_t n) { F.p[q] = ptr; F.n[q] = n; ++q; most = std::max(most, n); };
        for (int s : {0, 4, 5, 6}) { sysIdx[s].alloc((size_t)g.count(s)); add(sysIdx[s].p, g.count(s)); }
        for (int a = 0; a < 3; ++a) { sysIdxT[a].alloc((size_t)g.count(0)); add(sysIdxT[a].p, g.count(0)); }
        for (int a = 0; a < 3; ++a) add(faceRow[a].p, g.count(1 + a));
        hipLaunchKernelGGL(k_fill_i32_many, dim3((unsigned)std::min<int64_t>(512, gridFor(most, BS)), (unsigned)q), dim3(BS), 0, stream, F, -1);
    }
    {
        // seven groups, one per kind of DOF: p, txx, tyy, tzz (all on the cell grid), then the YZ / XZ / XY edge stresses
        const int samples[7] = {0, 0, 0, 0, 4, 5, 6}, weights[7] = {1, 1, 1, 1, 1, 1, 1};
        int32_t* outs[7] = {sysIdx[0].p, sysIdxT[0].p, sysIdxT[1].p, sysIdxT[2].p, sysIdx[4].p, sysIdx[5].p, sysIdx[6].p};
        int64_t range[2] = {0, 0};
        const int64_t tot = interleavedIndexAssignEx(7, samples, weights, outs, false, range);
        if (tot != nSys) throw Error("internal numbering: system DOF count mismatch");
        ownLo = range[0]; ownHi = range[1];
    }
    {
        const int samples[3] = {1, 2, 3}, weights[3] = {1, 1, 1};
        int32_t* outs[3] = {faceRow[0].p, faceRow[1].p, faceRow[2].p};
        const int64_t tot = interleavedIndexAssignEx(3, samples, weights, outs, slabEnabled, nullptr);
        if (!slabEnabled && tot != nAct) throw Error("internal numbering: face row count mismatch");
        nActiveVs = tot;   // with a slab: only the owned active faces get a row
    }
    permSys.alloc((size_t)nSys);
    permRow.alloc((size_t)nAct);
    if (slabEnabled) HIP_CHECK(hipMemsetAsync(permRow.p, 0, (size_t)std::max<int64_t>(nAct, 1) * sizeof(int32_t), stream));
    hipLaunchKernelGGL(k_perm_cells, dim3(gridFor(g.count(0), BS)), dim3(BS), 0, stream, g, activeIdx[0].p, sysIdx[0].p, sysIdxT[0].p, sysIdxT[1].p,
                       sysIdxT[2].p, nPq, nC, permSys.p);
    const int64_t eoff[3] = {nPq + 3 * nC, nPq + 3 * nC + nEdge[0], nPq + 3 * nC + nEdge[0] + nEdge[1]};
    for (int e = 0; e < 3; ++e)
        hipLaunchKernelGGL(k_perm_simple, dim3(gridFor(g.count(4 + e), BS)), dim3(BS), 0, stream, activeIdx[4 + e].p, sysIdx[4 + e].p,
                           g.count(4 + e), eoff[e], permSys.p);
    const int64_t foff[3] = {0, nFace[0], nFace[0] + nFace[1]};
    for (int a = 0; a < 3; ++a)
        hipLaunchKernelGGL(k_perm_simple, dim3(gridFor(g.count(1 + a), BS)), dim3(BS), 0, stream, activeIdx[1 + a].p, faceRow[a].p,
                           g.count(1 + a), foff[a], permRow.p);
}

// Exchange lists of the decomposition, built on the host from one-layer slices of sysIdx.
namespace {
// out[q] = src at the cross-section position q of the layer `layer` along `axis` (the two other axes over [r0, r1), lower axis fastest)
__global__ void k_slice(const int32_t* __restrict__ src, int3 d, int axis, int layer, int3 r0, int3 r1, int32_t* __restrict__ out) {
    const int b = axis == 0 ? 1 : 0, c = axis == 2 ? 1 : 2;
    const int nb = comp(r1, b) - comp(r0, b), nc = comp(r1, c) - comp(r0, c);
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nb * nc) return;
    int3 p = make_int3(0, 0, 0);
    addc(p, axis, layer); addc(p, b, comp(r0, b) + q % nb); addc(p, c, comp(r0, c) + q / nb);
    out[q] = oob3(d, p.x, p.y, p.z) ? -1 : src[lin3(d, p.x, p.y, p.z)];
}
}  // namespace
void ps_context::buildHaloLists() {
    for (int a = 0; a < NLINK; ++a) { nLowHalo[a] = nLowOwn[a] = nUpHalo[a] = nUpOwn[a] = 0; hashLowHalo[a] = hashLowOwn[a] = hashUpHalo[a] = hashUpOwn[a] = 0; }
    for (auto& v : hostOwnList) v.clear();
    if (!slabEnabled) return;
    DevBuf<int32_t>& scr = scrSlice;
    // the slice of sample grid s perpendicular to `axis` at `layer`, over the positions this rank owns along the two other axes (the
    // neighbour across the cut owns the same cross-section: the bricks of a row share their ranges)
    auto sliceOf = [&](const int32_t* src, int s, int axis, int layer, std::vector<int32_t>& out) {
        const int3 d = g.dims(s);
        int r0[3], r1[3];
        // Along an axis b EARLIER than the cut's, the plane hi_b is taken even when it is the neighbour's: the exchanges run axis after axis
        // (x, y, z for values; z, y, x for contributions), so what a rank holds there is the copy it received from / will pass on to its
        // neighbour along b — that is how an edge on two cuts reaches the rank diagonally below (the skin rows of a tile in the corner of its
        // brick touch such edges, and a tile's rows belong to the tile's owner whatever plane they lie on).
        // (haloForward false: no such copies — every list holds samples of the sender's own; Dist::decideExchangeMode checks that nothing else is needed)
        for (int b = 0; b < 3; ++b) { r0[b] = brick.lo[b]; r1[b] = brick.hi[b] + ((Own::onPlane(s, b) && (!brick.hasUpper[b] || (haloForward && b < axis))) ? 1 : 0); }
        const int b = axis == 0 ? 1 : 0, c = axis == 2 ? 1 : 2;
        const size_t n = (size_t)(r1[b] - r0[b]) * (size_t)(r1[c] - r0[c]);
        out.assign(n, -1);
        if (n == 0 || layer < 0 || layer >= comp(d, axis)) return;
        scr.alloc(n);
        hipLaunchKernelGGL(k_slice, dim3(gridFor((int64_t)n, BS)), dim3(BS), 0, stream, src, d, axis, layer, make_int3(r0[0], r0[1], r0[2]),
                           make_int3(r1[0], r1[1], r1[2]), scr.p);
        HIP_CHECK(hipMemcpyAsync(out.data(), scr.p, n * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    };
    // order-sensitive hash of the keys (position in the cross-section, sample grid) of a list: local indices differ
    // between the two ranks of a cut, the keys must not (Dist::checkLists)
    auto mix = [](uint64_t& h, uint64_t key) { h = (h ^ key) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; };
    auto cellsOf = [&](int axis, int layer, std::vector<int32_t>& list, uint64_t& h) {
        std::vector<int32_t> sl, sx, sy, sz;
        sliceOf(sysIdx[0].p, 0, axis, layer, sl);
        sliceOf(sysIdxT[0].p, 0, axis, layer, sx); sliceOf(sysIdxT[1].p, 0, axis, layer, sy); sliceOf(sysIdxT[2].p, 0, axis, layer, sz);
        for (size_t q = 0; q < sl.size(); ++q) { const int32_t b = sl[q]; if (b >= 0) { list.push_back(b); list.push_back(sx[q]); list.push_back(sy[q]); list.push_back(sz[q]); mix(h, (uint64_t)q * 8); } }
    };
    auto edgesOf = [&](int axis, int layer, std::vector<int32_t>& list, uint64_t& h) {   // the two edge grids that live on the planes of `axis`
        for (int s = 4; s < 7; ++s) {
            if (!Own::onPlane(s, axis)) continue;
            std::vector<int32_t> sl;
            sliceOf(sysIdx[s].p, s, axis, layer, sl);
            for (size_t q = 0; q < sl.size(); ++q) { const int32_t b = sl[q]; if (b >= 0) { list.push_back(b); mix(h, (uint64_t)q * 8 + (uint64_t)s); } }
        }
    };
    auto up = [&](const std::vector<int32_t>& h, DevBuf<int32_t>& d, int64_t& n) {
        n = (int64_t)h.size();
        d.alloc(h.size());
        if (n) HIP_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, stream));
    };
    for (int a = 0; a < 3; ++a) {
        const int lo = brick.lo[a], hi = brick.hi[a];
        std::vector<int32_t> lowHalo, lowOwn, upHalo, upOwn;
        hashLowHalo[a] = hashLowOwn[a] = hashUpHalo[a] = hashUpOwn[a] = 0;
        if (brick.hasLower[a]) {
            cellsOf(a, lo - 1, lowHalo, hashLowHalo[a]);                                   // their last layer, touched by my faces on the plane lo
            cellsOf(a, lo, lowOwn, hashLowOwn[a]); edgesOf(a, lo, lowOwn, hashLowOwn[a]);   // mine, touched by their rows
        }
        if (brick.hasUpper[a]) {
            cellsOf(a, hi, upHalo, hashUpHalo[a]); edgesOf(a, hi, upHalo, hashUpHalo[a]);   // theirs, touched by my rows
            cellsOf(a, hi - 1, upOwn, hashUpOwn[a]);                                      // mine, touched by their faces on the plane hi
        }
        up(lowHalo, listLowHalo[a], nLowHalo[a]); up(lowOwn, listLowOwn[a], nLowOwn[a]); up(upHalo, listUpHalo[a], nUpHalo[a]); up(upOwn, listUpOwn[a], nUpOwn[a]);
        hostOwnList[2 * a] = lowOwn; hostOwnList[2 * a + 1] = upOwn;
        const size_t mx = (size_t)std::max<int64_t>(std::max(nLowHalo[a], nLowOwn[a]), std::max(nUpHalo[a], nUpOwn[a])) + 8;   // >= 8: Dist::checkLists ships counts + hashes through these buffers
        sendLo[a].alloc(mx); sendUp[a].alloc(mx); recvLo[a].alloc(mx); recvUp[a].alloc(mx);
        HIP_CHECK(hipStreamSynchronize(stream));                                           // (the host vectors go out of scope)
    }
    // The diagonal links (one-round mode only; in the forwarding mode these samples travel as copies through two axis exchanges): the edge stresses
    // on the corner line of two cuts.  Link 3 + d, d = (x, y), (x, z), (y, z): the edge grid that lives on both planes (XY = 6, XZ = 5, YZ = 4), the third
    // axis c over this rank's owned layers — the diagonal brick owns the same layers (the bricks of a row share their ranges).
    //   above: the line (plane hi_a, plane hi_b) belongs to the brick one up along both: my halo, touched by the skin rows of my tile in that corner;
    //   below: the line (plane lo_a, plane lo_b) is mine, touched by the rows of the brick one down along both.
    for (int d = 0; d < 3; ++d) {
        const int l = 3 + d, a = d == 2 ? 1 : 0, b = d == 0 ? 1 : 2, c = 3 - a - b, s = d == 0 ? 6 : (d == 1 ? 5 : 4);
        std::vector<int32_t> upHalo, lowOwn;
        auto lineOf = [&](int pa, int pb, std::vector<int32_t>& list, uint64_t& h) {
            const int3 dm = g.dims(s);
            if (pa < 0 || pa >= comp(dm, a) || pb < 0 || pb >= comp(dm, b)) return;
            int r0[3], r1[3];
            r0[a] = pa; r1[a] = pa + 1; r0[b] = pb; r1[b] = pb + 1; r0[c] = brick.lo[c]; r1[c] = brick.hi[c];
            const int n = r1[c] - r0[c];
            if (n <= 0) return;
            // k_slice cuts a layer along `axis` over the ranges of the two other axes, lower axis fastest: here the layer pa along a, one position along b
            const int o1 = a == 0 ? 1 : 0, o2 = a == 2 ? 1 : 2;     // the two other axes of a, in k_slice's order
            const size_t cnt = (size_t)(r1[o1] - r0[o1]) * (size_t)(r1[o2] - r0[o2]);
            scrSlice.alloc(cnt);
            hipLaunchKernelGGL(k_slice, dim3(gridFor((int64_t)cnt, BS)), dim3(BS), 0, stream, (const int32_t*)sysIdx[s].p, dm, a, pa, make_int3(r0[0], r0[1], r0[2]),
                               make_int3(r1[0], r1[1], r1[2]), scrSlice.p);
            std::vector<int32_t> sl(cnt);
            HIP_CHECK(hipMemcpyAsync(sl.data(), scrSlice.p, cnt * 4, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            for (size_t q = 0; q < sl.size(); ++q) if (sl[q] >= 0) { list.push_back(sl[q]); mix(h, (uint64_t)q * 8 + (uint64_t)s); }
        };
        if (!haloForward && linkUpper(l)) lineOf(brick.hi[a], brick.hi[b], upHalo, hashUpHalo[l]);
        if (!haloForward && linkLower(l)) lineOf(brick.lo[a], brick.lo[b], lowOwn, hashLowOwn[l]);
        std::vector<int32_t> none;
        up(none, listLowHalo[l], nLowHalo[l]); up(lowOwn, listLowOwn[l], nLowOwn[l]); up(upHalo, listUpHalo[l], nUpHalo[l]); up(none, listUpOwn[l], nUpOwn[l]);
        hostOwnList[2 * l] = lowOwn; hostOwnList[2 * l + 1].clear();
        const size_t mx = (size_t)std::max(nLowOwn[l], nUpHalo[l]) + 8;
        sendLo[l].alloc(mx); sendUp[l].alloc(mx); recvLo[l].alloc(mx); recvUp[l].alloc(mx);
        HIP_CHECK(hipStreamSynchronize(stream));
    }
}

void ps_context::buildValidFaces() {
    for (int a = 0; a < 3; ++a) {
        const int64_t n = g.count(1 + a);
        hipLaunchKernelGGL(k_valid, dim3(gridFor(n, BS)), dim3(BS), 0, stream, labels[1 + a].p, n, valid[a].p);
    }
}
