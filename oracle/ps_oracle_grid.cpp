// TEST INFRASTRUCTURE — see ps_oracle.hpp.  Grid stages: weights, classification, regions, indices.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <limits>

#include "ps_oracle.hpp"

namespace psoracle {

static const float kSampleOffset[7][3] = {
    // Solver.h:193-222 SamplingOffset(); order: center, faceX, faceY, faceZ, edgeYZ, edgeXZ, edgeXY
    {0.5f, 0.5f, 0.5f}, {0.f, 0.5f, 0.5f}, {0.5f, 0.f, 0.5f}, {0.5f, 0.5f, 0.f},
    {0.5f, 0.f, 0.f},   {0.f, 0.5f, 0.f},  {0.f, 0.f, 0.5f}};

Dim Oracle::centerDim() const { Dim d; d.n[0] = nx; d.n[1] = ny; d.n[2] = nz; return d; }
Dim Oracle::faceDim(int a) const { Dim d = centerDim(); d.n[a] += 1; return d; }
Dim Oracle::edgeDim(int e) const {
    Dim d = centerDim();
    for (int b = 0; b < 3; ++b) if (b != e) d.n[b] += 1;
    return d;
}

static inline bool isActive(int32_t l) { return l == PS_ACTIVEFLUID || l == PS_BOUNDARY; }   // Solver.h:708-710
static inline bool isReduced(int32_t l) { return l == PS_REDUCED || l == PS_BOUNDARY; }      // Solver.h:711-713

// SIM_RawField::getValue(pos) on a cell-centred field: trilinear between voxel centres, positions
// clamped to the voxel-centre box (streak border).  Position in voxel units, corner origin.
// fp32, lerp(a,b,t)=a+(b-a)*t, x then y then z.  (HDK out of tree — call sites Solver.cpp:1920-1924.)
float Oracle::sampleCenterField(const Field<float>& f, float px, float py, float pz) const {
    const float p[3] = {px, py, pz};
    int i0[3], i1[3];
    float t[3];
    for (int a = 0; a < 3; ++a) {
        const int n = f.d.n[a];
        float u = p[a] - 0.5f;
        if (u < 0.f) u = 0.f;
        if (u > (float)(n - 1)) u = (float)(n - 1);
        int b = (int)u;
        if (b >= n - 1) { b = n - 1; i0[a] = b; i1[a] = b; t[a] = 0.f; }
        else { i0[a] = b; i1[a] = b + 1; t[a] = u - (float)b; }
    }
    auto L = [](float a, float b, float tt) { return a + (b - a) * tt; };
    const float c00 = L(f.at(i0[0], i0[1], i0[2]), f.at(i1[0], i0[1], i0[2]), t[0]);
    const float c10 = L(f.at(i0[0], i1[1], i0[2]), f.at(i1[0], i1[1], i0[2]), t[0]);
    const float c01 = L(f.at(i0[0], i0[1], i1[2]), f.at(i1[0], i0[1], i1[2]), t[0]);
    const float c11 = L(f.at(i0[0], i1[1], i1[2]), f.at(i1[0], i1[1], i1[2]), t[0]);
    const float c0 = L(c00, c10, t[1]);
    const float c1 = L(c01, c11, t[1]);
    return L(c0, c1, t[2]);
}

float Oracle::localViscosityAtCell(int i, int j, int k) const {
    // centerLabels.indexToPos + getLocalViscosity (ConstructMatrixBlocks.cpp:780-782, Solver.cpp:1549-1551)
    return sampleCenterField(viscosity, (float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f);
}
float Oracle::localViscosityAtEdge(int e, int i, int j, int k) const {
    // edgeLabels(e)->indexToPos + getLocalViscosity (ConstructMatrixBlocks.cpp:693-695, Solver.cpp:1615-1617)
    const float* o = kSampleOffset[4 + e];
    return sampleCenterField(viscosity, (float)i + o[0], (float)j + o[1], (float)k + o[2]);
}

// SIM_RawField::computeSDFWeightsSampled(sdf, 2 samples/axis, invert=false, minweight=0)
// (Solver.cpp:292-326).  HDK out of tree; restated as: fraction of the 2x2x2 sub-sample points of the
// voxel-sized box around the sample point whose trilinear SDF value is < 0.
void Oracle::computeSDFWeightsSampled(Field<float>& dst, const Dim& d, const float off[3],
                                      const Field<float>& sdf, bool negate) const {
    dst.init(d, 0.f);
    const float sub[2] = {-0.25f, 0.25f};
#pragma omp parallel for schedule(dynamic, 1) num_threads(setupThreads > 1 ? setupThreads : 1)   // (independent samples: identical bits)
    for (int k = 0; k < d.n[2]; ++k)
        for (int j = 0; j < d.n[1]; ++j)
            for (int i = 0; i < d.n[0]; ++i) {
                const float cx = (float)i + off[0], cy = (float)j + off[1], cz = (float)k + off[2];
                int cnt = 0;
                for (int sz = 0; sz < 2; ++sz)
                    for (int sy = 0; sy < 2; ++sy)
                        for (int sx = 0; sx < 2; ++sx) {
                            float v = sampleCenterField(sdf, cx + sub[sx], cy + sub[sy], cz + sub[sz]);
                            if (negate) v = -v;
                            if (v < 0.f) ++cnt;
                        }
                dst.at(i, j, k) = (float)cnt / 8.0f;
            }
}

// Solver.cpp:238-289
void Oracle::buildIntegrationWeightsAlt(const ps_fields_in* in) {
    bool haveAll = true;
    for (int w = 0; w < 14; ++w) if (!in->weights[w]) haveAll = false;
    for (int s = 0; s < 7; ++s) {
        const Dim d = s == 0 ? centerDim() : (s <= 3 ? faceDim(s - 1) : edgeDim(s - 4));
        if (haveAll) {
            liquidW[s].init(d, 0.f);
            fluidW[s].init(d, 0.f);
            std::memcpy(liquidW[s].v.data(), in->weights[s], sizeof(float) * (size_t)d.size());
            std::memcpy(fluidW[s].v.data(), in->weights[7 + s], sizeof(float) * (size_t)d.size());
        } else {
            computeSDFWeightsSampled(liquidW[s], d, kSampleOffset[s], surface, false);
            computeSDFWeightsSampled(fluidW[s], d, kSampleOffset[s], collision, P.negateCollision != 0);
        }
    }
}

// Classifier.cpp:56-128
void Oracle::classifyCells() {
    Field<int32_t>& L = labels[0];
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                bool isInSolve = false, isInFluid = true;
                if (liquidW[0].at(i, j, k) > 0.f) isInSolve = true;
                if (!isInSolve)
                    for (int axis = 0; axis < 3; ++axis)
                        for (int dir = 0; dir < 2; ++dir) {
                            int f[3] = {i, j, k};
                            f[axis] += dir;  // cellToFaceMap
                            if (liquidW[1 + axis].at(f[0], f[1], f[2]) > 0.f) isInSolve = true;
                        }
                if (fluidW[0].at(i, j, k) == 0.f) isInFluid = false;
                if (isInSolve) L.at(i, j, k) = isInFluid ? PS_GENERICFLUID : PS_SOLID;
                else L.at(i, j, k) = PS_UNSOLVED;
            }
}

// Classifier.cpp:291-508 (+ setActiveLayerCells Solver.cpp:2022-2060).  The sorted/deduplicated cell
// lists are an implementation detail of the reference; the layers themselves are sets.
void Oracle::airBoundaryLayer() {
    Field<int32_t>& L = labels[0];
    const int n[3] = {nx, ny, nz};
    std::vector<int64_t> layer, next;
    // buildInitialAirBoundaryLayer :364-430
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                if (L.at(i, j, k) != PS_GENERICFLUID) continue;
                bool isBoundaryCell = false;
                for (int axis = 0; axis < 3; ++axis)
                    for (int dir = 0; dir < 2; ++dir) {
                        int c[3] = {i, j, k};
                        c[axis] += dir ? 1 : -1;
                        if (c[axis] < 0 || c[axis] >= n[axis]) continue;
                        int f[3] = {i, j, k};
                        f[axis] += dir;
                        if (L.at(c[0], c[1], c[2]) == PS_UNSOLVED) isBoundaryCell = true;
                        if (liquidW[1 + axis].at(f[0], f[1], f[2]) < 1.f) isBoundaryCell = true;
                    }
                if (isBoundaryCell) layer.push_back(L.d.lin(i, j, k));
            }
    const int Lsz = P.activeLiquidBoundaryLayerSize;
    for (int li = 0; li < Lsz - 1; ++li) {   // :328
        for (int64_t c : layer) L.v[(size_t)c] = PS_ACTIVEFLUID;   // setActiveLayerCells
        if (li < Lsz - 2) {                   // :355, buildNextLiquidBoundaryLayer :432-508
            next.clear();
            for (int64_t c : layer) {
                const int i = (int)(c % nx), j = (int)((c / nx) % ny), k = (int)(c / ((int64_t)nx * ny));
                for (int axis = 0; axis < 3; ++axis)
                    for (int dir = 0; dir < 2; ++dir) {
                        int a[3] = {i, j, k};
                        a[axis] += dir ? 1 : -1;
                        if (a[axis] < 0 || a[axis] >= n[axis]) continue;
                        int f[3] = {i, j, k};
                        f[axis] += dir;
                        if (liquidW[1 + axis].at(f[0], f[1], f[2]) > 0.f && L.at(a[0], a[1], a[2]) == PS_GENERICFLUID)
                            next.push_back(L.d.lin(a[0], a[1], a[2]));
                    }
            }
            std::sort(next.begin(), next.end());
            next.erase(std::unique(next.begin(), next.end()), next.end());
            layer.swap(next);
        }
    }
}

// Classifier.cpp:510-703
void Oracle::solidBoundaryLayer() {
    Field<int32_t>& L = labels[0];
    const int n[3] = {nx, ny, nz};
    Field<int32_t> visited;
    visited.init(centerDim(), PS_UNVISITED);
    std::vector<int64_t> layer, next;
    // buildInitialSolidBoundaryLayer :573-641
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const int32_t l = L.at(i, j, k);
                if (l != PS_GENERICFLUID && l != PS_ACTIVEFLUID) continue;
                bool isBoundaryCell = false;
                for (int axis = 0; axis < 3; ++axis)
                    for (int dir = 0; dir < 2; ++dir) {
                        int c[3] = {i, j, k};
                        c[axis] += dir ? 1 : -1;
                        if (c[axis] < 0 || c[axis] >= n[axis]) { isBoundaryCell = true; continue; }  // :617-621
                        if (L.at(c[0], c[1], c[2]) == PS_SOLID) isBoundaryCell = true;
                    }
                if (isBoundaryCell) layer.push_back(L.d.lin(i, j, k));
            }
    const int S = P.activeSolidBoundaryLayerSize;
    for (int li = 0; li < S; ++li) {   // :526
        for (int64_t c : layer) { L.v[(size_t)c] = PS_ACTIVEFLUID; visited.v[(size_t)c] = PS_VISITED; }
        if (li < S - 1) {              // :563, buildNextSolidBoundaryLayer :643-703
            next.clear();
            for (int64_t c : layer) {
                const int i = (int)(c % nx), j = (int)((c / nx) % ny), k = (int)(c / ((int64_t)nx * ny));
                for (int axis = 0; axis < 3; ++axis)
                    for (int dir = 0; dir < 2; ++dir) {
                        int a[3] = {i, j, k};
                        a[axis] += dir ? 1 : -1;
                        if (a[axis] < 0 || a[axis] >= n[axis]) continue;
                        int f[3] = {i, j, k};
                        f[axis] += dir;
                        if (liquidW[1 + axis].at(f[0], f[1], f[2]) > 0.f) {
                            const int32_t al = L.at(a[0], a[1], a[2]);
                            if (visited.at(a[0], a[1], a[2]) == PS_UNVISITED && (al == PS_ACTIVEFLUID || al == PS_GENERICFLUID))
                                next.push_back(L.d.lin(a[0], a[1], a[2]));
                        }
                    }
            }
            std::sort(next.begin(), next.end());
            next.erase(std::unique(next.begin(), next.end()), next.end());
            layer.swap(next);
        }
    }
}

// Classifier.cpp:705-746
void Oracle::constructTiles() {
    Field<int32_t>& L = labels[0];
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                if (L.at(i, j, k) != PS_GENERICFLUID) continue;
                for (int p = 0; p < P.tilePadding; ++p)
                    if (i % P.tileSize == p || j % P.tileSize == p || k % P.tileSize == p)
                        L.at(i, j, k) = PS_ACTIVEFLUID;
            }
}

// Classifier.cpp:179-199
void Oracle::constructReducedRegions() {
    airBoundaryLayer();
    solidBoundaryLayer();
    if (P.doTile) constructTiles();
    for (auto& l : labels[0].v) if (l == PS_GENERICFLUID) l = PS_REDUCED;  // overwriteIndices :189
}
void Oracle::constructOnlyActiveRegions() {
    for (auto& l : labels[0].v) if (l == PS_GENERICFLUID) l = PS_ACTIVEFLUID;  // :198
}

// Classifier.cpp:201-207, 752-832 (findFaceLabelFromCenter)
void Oracle::classifyFaces() {
    for (int axis = 0; axis < 3; ++axis) {
        const Dim d = faceDim(axis);
        Field<int32_t>& L = labels[1 + axis];
        for (int k = 0; k < d.n[2]; ++k)
            for (int j = 0; j < d.n[1]; ++j)
                for (int i = 0; i < d.n[0]; ++i) {
                    int32_t retval = PS_UNSOLVED;
                    bool isActiveVelocity = false;
                    for (int dir = 0; dir < 2; ++dir) {
                        int c[3] = {i, j, k};
                        c[axis] += dir - 1;  // faceToCellMap
                        if (labels[0].d.oob(c[0], c[1], c[2])) continue;
                        if (liquidW[0].at(c[0], c[1], c[2]) > 0.f) isActiveVelocity = true;
                    }
                    if (!isActiveVelocity)
                        for (int edgeAxis = 0; edgeAxis < 3 && !isActiveVelocity; ++edgeAxis) {
                            if (edgeAxis == axis) continue;
                            for (int dir = 0; dir < 2; ++dir) {
                                int e[3] = {i, j, k};
                                e[3 - axis - edgeAxis] += dir;  // faceToEdgeMap
                                if (liquidW[4 + edgeAxis].at(e[0], e[1], e[2]) > 0.f) { isActiveVelocity = true; break; }
                            }
                        }
                    if (isActiveVelocity) retval = fluidW[1 + axis].at(i, j, k) < 0.5f ? PS_SOLID : PS_GENERICFLUID;
                    L.at(i, j, k) = retval;
                }
    }
}

// Classifier.cpp:209-215, 919-949, 1021-1067 (findEdgeLabelFromFaceAlt).  Face-weight reads at the
// edge's own index can fall one past the face grid; UT_VoxelArray::getValue streaks (clamps).
void Oracle::classifyEdges() {
    for (int e = 0; e < 3; ++e) {
        const Dim d = edgeDim(e);
        Field<int32_t>& L = labels[4 + e];
        // the two face axes touching this edge, in the reference's order: XY -> (X,Y), XZ -> (X,Z), YZ -> (Y,Z)
        const int fa = e == 0 ? 1 : 0;
        const int fb = e == 2 ? 1 : 2;
        for (int k = 0; k < d.n[2]; ++k)
            for (int j = 0; j < d.n[1]; ++j)
                for (int i = 0; i < d.n[0]; ++i) {
                    bool insystem = liquidW[4 + e].at(i, j, k) != 0.f && fluidW[4 + e].at(i, j, k) != 0.f;
                    if (!insystem) { L.at(i, j, k) = PS_UNSOLVED; continue; }
                    // face fa at (i,j,k) and at (i,j,k) - e_fb ; face fb at (i,j,k) and (i,j,k) - e_fa
                    int a2[3] = {i, j, k}; a2[fb] -= 1;
                    int b2[3] = {i, j, k}; b2[fa] -= 1;
                    insystem = liquidW[1 + fa].getStreak(i, j, k) != 0.f && !faceDim(fa).oob(a2[0], a2[1], a2[2]) &&
                               liquidW[1 + fa].getStreak(a2[0], a2[1], a2[2]) != 0.f &&
                               liquidW[1 + fb].getStreak(i, j, k) != 0.f && !faceDim(fb).oob(b2[0], b2[1], b2[2]) &&
                               liquidW[1 + fb].getStreak(b2[0], b2[1], b2[2]) != 0.f;
                    L.at(i, j, k) = insystem ? PS_GENERICFLUID : PS_UNSOLVED;
                }
    }
}

// SIM_VolumetricConnectedComponentBuilder (HDK, out of tree; call site Classifier.cpp:220-229):
// REDUCED cells connected through faces with liquid weight > 0.  Numbering restated as: components
// numbered by the traversal-order rank of their first cell.
void Oracle::connectedComponents() {
    const Dim d = centerDim();
    Field<int32_t>& R = reducedIdx[0];
    const Field<int32_t>& L = labels[0];
    int32_t count = 0;
    std::vector<int64_t> stack;
    forEachOrdered(d, [&](int i, int j, int k) {
        if (L.at(i, j, k) != PS_REDUCED || R.at(i, j, k) != PS_UNASSIGNED) return;
        const int32_t id = count++;
        R.at(i, j, k) = id;
        stack.push_back(d.lin(i, j, k));
        while (!stack.empty()) {
            const int64_t c = stack.back();
            stack.pop_back();
            const int ci = (int)(c % nx), cj = (int)((c / nx) % ny), ck = (int)(c / ((int64_t)nx * ny));
            for (int axis = 0; axis < 3; ++axis)
                for (int dir = 0; dir < 2; ++dir) {
                    int a[3] = {ci, cj, ck};
                    a[axis] += dir ? 1 : -1;
                    if (d.oob(a[0], a[1], a[2])) continue;
                    int f[3] = {ci, cj, ck};
                    f[axis] += dir;
                    if (!(liquidW[1 + axis].at(f[0], f[1], f[2]) > 0.f)) continue;
                    if (L.at(a[0], a[1], a[2]) == PS_REDUCED && R.at(a[0], a[1], a[2]) == PS_UNASSIGNED) {
                        R.at(a[0], a[1], a[2]) = id;
                        stack.push_back(d.lin(a[0], a[1], a[2]));
                    }
                }
        }
    });
    regionCount = count;
}

// Classifier.cpp:1073-1172 — serial, in-place, traversal-ordered fix point.
void Oracle::fixReducedRegionBoundaries() {
    const Dim d = centerDim();
    Field<int32_t>& L = labels[0];
    Field<int32_t>& R = reducedIdx[0];
    bool done = false;
    while (!done) {
        done = true;
        forEachOrdered(d, [&](int i, int j, int k) {
            if (L.at(i, j, k) != PS_ACTIVEFLUID) return;
            bool applyFix = false, isBoundaryCell = false;
            int adjacentInteriorRegion = 0;
            for (int axis = 0; axis < 3; ++axis)
                for (int dir = 0; dir < 2; ++dir) {
                    int a[3] = {i, j, k};
                    a[axis] += dir ? 1 : -1;
                    if (isReduced(L.getConst(a[0], a[1], a[2], PS_UNASSIGNED))) {
                        if (!isBoundaryCell) {
                            isBoundaryCell = true;
                            adjacentInteriorRegion = R.at(a[0], a[1], a[2]);
                        } else if (R.at(a[0], a[1], a[2]) != adjacentInteriorRegion)
                            applyFix = true;
                    }
                }
            if (applyFix) {
                done = false;
                for (int axis = 0; axis < 3; ++axis)
                    for (int dir = 0; dir < 2; ++dir) {
                        int a[3] = {i, j, k};
                        a[axis] += dir ? 1 : -1;
                        if (isReduced(L.getConst(a[0], a[1], a[2], PS_UNASSIGNED))) {
                            L.at(a[0], a[1], a[2]) = PS_ACTIVEFLUID;
                            R.at(a[0], a[1], a[2]) = PS_UNASSIGNED;
                        }
                    }
            }
        });
    }
}

// Classifier.cpp:1174-1313 (+ buildInteriorBoundingBoxes :1418-1467)
void Oracle::fixSmallReducedRegions() {
    const Dim d = centerDim();
    Field<int32_t>& L = labels[0];
    Field<int32_t>& R = reducedIdx[0];
    const int64_t Rn = regionCount;
    std::vector<int64_t> bbmin((size_t)Rn * 3, std::numeric_limits<int64_t>::max());
    std::vector<int64_t> bbmax((size_t)Rn * 3, std::numeric_limits<int64_t>::min());
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                if (!isReduced(L.at(i, j, k))) continue;
                const int64_t r = R.at(i, j, k);
                const int c[3] = {i, j, k};
                for (int a = 0; a < 3; ++a) {
                    bbmin[(size_t)r * 3 + a] = std::min<int64_t>(bbmin[(size_t)r * 3 + a], c[a]);
                    bbmax[(size_t)r * 3 + a] = std::max<int64_t>(bbmax[(size_t)r * 3 + a], c[a]);
                }
            }
    std::vector<char> doRemove((size_t)Rn, 0);
    for (int64_t r = 0; r < Rn; ++r)
        for (int a = 0; a < 3; ++a) {
            const int64_t lo = bbmin[(size_t)r * 3 + a], hi = bbmax[(size_t)r * 3 + a];
            // a region emptied by fixReducedRegionBoundaries keeps its +max/-min sentinels; the
            // reference's `min > max-3` (:1239) overflows there (UB) — restated as "removed".
            if (hi < lo) { doRemove[(size_t)r] = 1; continue; }
            if (hi == lo) doRemove[(size_t)r] = 1;          // :1236
            if (lo > hi - 3) doRemove[(size_t)r] = 1;       // :1239
        }
    std::vector<int64_t> remap((size_t)Rn, -1);
    int64_t regionMap = 0;
    for (int64_t r = 0; r < Rn; ++r) if (!doRemove[(size_t)r]) remap[(size_t)r] = regionMap++;
    if (Rn > regionMap) {
        regionCount = regionMap;
        for (size_t c = 0; c < L.v.size(); ++c) {   // remapInteriorRegions :1264-1313
            if (!isReduced(L.v[c])) continue;
            const int64_t r = R.v[c];
            if (doRemove[(size_t)r]) { L.v[c] = PS_ACTIVEFLUID; R.v[c] = PS_UNASSIGNED; }
            else R.v[c] = (int32_t)remap[(size_t)r];
        }
    }
    (void)d;
}

// Classifier.cpp:217-239
void Oracle::constructCenterReducedIndices() {
    connectedComponents();
    fixReducedRegionBoundaries();
    fixSmallReducedRegions();
}

// Classifier.cpp:241-247, 1473-1528
void Oracle::constructFacesReducedIndices() {
    const Dim cd = centerDim();
    for (int axis = 0; axis < 3; ++axis) {
        const Dim d = faceDim(axis);
        for (int k = 0; k < d.n[2]; ++k)
            for (int j = 0; j < d.n[1]; ++j)
                for (int i = 0; i < d.n[0]; ++i) {
                    int32_t idx = PS_UNASSIGNED;
                    int m[3] = {i, j, k};
                    m[axis] -= 1;
                    if (!cd.oob(i, j, k) && labels[0].at(i, j, k) == PS_REDUCED) idx = reducedIdx[0].at(i, j, k);
                    else if (!cd.oob(m[0], m[1], m[2]) && labels[0].at(m[0], m[1], m[2]) == PS_REDUCED)
                        idx = reducedIdx[0].at(m[0], m[1], m[2]);
                    if (idx != PS_UNASSIGNED) {
                        labels[1 + axis].at(i, j, k) = PS_REDUCED;
                        reducedIdx[1 + axis].at(i, j, k) = idx;
                    }
                }
    }
}

// Classifier.cpp:249-255, 1534-1659
void Oracle::constructEdgesReducedIndices() {
    for (int e = 0; e < 3; ++e) {
        const Dim d = edgeDim(e);
        const int fa = e == 0 ? 1 : 0;   // first face axis in the reference's priority order
        const int fb = e == 2 ? 1 : 2;   // second
        const Dim da = faceDim(fa), db = faceDim(fb);
        auto red = [&](int fax, const Dim& fd, const int* c) {
            return !fd.oob(c[0], c[1], c[2]) && labels[1 + fax].at(c[0], c[1], c[2]) == PS_REDUCED;
        };
        for (int k = 0; k < d.n[2]; ++k)
            for (int j = 0; j < d.n[1]; ++j)
                for (int i = 0; i < d.n[0]; ++i) {
                    const int a1[3] = {i, j, k};
                    int a2[3] = {i, j, k}; a2[fb] -= 1;   // face fa, shifted along the other face axis
                    const int b1[3] = {i, j, k};
                    int b2[3] = {i, j, k}; b2[fa] -= 1;
                    int32_t label = PS_UNASSIGNED, idx = PS_UNASSIGNED;
                    const bool ra1 = red(fa, da, a1), ra2 = red(fa, da, a2), rb1 = red(fb, db, b1), rb2 = red(fb, db, b2);
                    if (ra1 && ra2 && rb1 && rb2) {
                        // XY,XZ: faceX(i,j,k); YZ: faceY(i,j-1,k)  (:1569,:1599,:1629)
                        if (e == 0) { int q[3] = {i, j - 1, k}; idx = reducedIdx[1 + 1].at(q[0], q[1], q[2]); }
                        else idx = reducedIdx[1 + fa].at(i, j, k);
                        label = PS_REDUCED;
                    } else if (ra1) { idx = reducedIdx[1 + fa].at(a1[0], a1[1], a1[2]); label = PS_BOUNDARY; }
                    else if (ra2) { idx = reducedIdx[1 + fa].at(a2[0], a2[1], a2[2]); label = PS_BOUNDARY; }
                    else if (rb1) { idx = reducedIdx[1 + fb].at(b1[0], b1[1], b1[2]); label = PS_BOUNDARY; }
                    else if (rb2) { idx = reducedIdx[1 + fb].at(b2[0], b2[1], b2[2]); label = PS_BOUNDARY; }
                    if (idx != PS_UNASSIGNED) {
                        labels[4 + e].at(i, j, k) = label;
                        reducedIdx[4 + e].at(i, j, k) = idx;
                    }
                }
    }
}

// Classifier.cpp:257-284, 1738-1770
void Oracle::constructActiveIndices() {
    for (int s = 0; s < 7; ++s) {
        for (auto& l : labels[s].v) if (l == PS_GENERICFLUID) l = PS_ACTIVEFLUID;
        int32_t idx = 0;
        Field<int32_t>& A = activeIdx[s];
        const Field<int32_t>& L = labels[s];
        forEachOrdered(L.d, [&](int i, int j, int k) {
            if (isActive(L.at(i, j, k))) A.at(i, j, k) = idx++;
        });
        if (s == 0) nCenter = idx;
        else if (s <= 3) nFace[s - 1] = idx;
        else nEdge[s - 4] = idx;
    }
}

// Classifier.cpp:4-54
void Oracle::buildValidFaces() {
    for (int a = 0; a < 3; ++a) {
        valid[a].init(faceDim(a), 1.f);
        for (size_t c = 0; c < valid[a].v.size(); ++c) {
            const int32_t l = labels[1 + a].v[c];
            valid[a].v[c] = (l == PS_UNSOLVED || l == PS_UNASSIGNED) ? 0.f : 1.f;
        }
    }
}

}  // namespace psoracle
