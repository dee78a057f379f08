// TEST INFRASTRUCTURE — see ps_oracle.hpp.  Per-tile dense blocks, stencil blocks, system assembly.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <limits>
#include <parallel/algorithm>   // __gnu_parallel::stable_sort (libstdc++ parallel mode: a stable multiway mergesort, the same result as std::stable_sort)

#include "ps_oracle.hpp"

namespace psoracle {

static inline bool isActive(int32_t l) { return l == PS_ACTIVEFLUID || l == PS_BOUNDARY; }
static inline bool isReduced(int32_t l) { return l == PS_REDUCED || l == PS_BOUNDARY; }
static inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }  // SYSclamp
static inline Trip mkT(int64_t r, int64_t c, double v) { return Trip{(int32_t)r, (int32_t)c, v}; }

// ---------------------------------------------------------------------------------------------
// CSR
// ---------------------------------------------------------------------------------------------
void CSR::fromTriplets(int64_t rows_, int64_t cols_, std::vector<Trip>& t, int threads) {
    rows = rows_; cols = cols_;
    auto less = [](const Trip& a, const Trip& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; };
    if (threads > 1) __gnu_parallel::stable_sort(t.begin(), t.end(), less, __gnu_parallel::multiway_mergesort_tag(threads));
    else std::stable_sort(t.begin(), t.end(), less);
    ptr.assign((size_t)rows + 1, 0);
    col.clear(); val.clear();
    col.reserve(t.size()); val.reserve(t.size());
    size_t p = 0;
    for (int64_t r = 0; r < rows; ++r) {
        ptr[(size_t)r] = (int64_t)val.size();
        while (p < t.size() && t[p].r == r) {
            const int64_t c = t[p].c;
            double s = 0;
            while (p < t.size() && t[p].r == r && t[p].c == c) { s += t[p].v; ++p; }
            col.push_back((int32_t)c);
            val.push_back(s);
        }
    }
    ptr[(size_t)rows] = (int64_t)val.size();
}
void CSR::mul(const double* x, double* y) const {
    for (int64_t r = 0; r < rows; ++r) {
        double s = 0;
        for (int64_t p = ptr[(size_t)r]; p < ptr[(size_t)r + 1]; ++p) s += val[(size_t)p] * x[col[(size_t)p]];
        y[r] = s;
    }
}
void CSR::mulT_add(const double* x, double* y) const {
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t p = ptr[(size_t)r]; p < ptr[(size_t)r + 1]; ++p) y[col[(size_t)p]] += val[(size_t)p] * x[r];
}
CSR CSR::transposed() const {
    CSR t;
    t.rows = cols; t.cols = rows;
    t.ptr.assign((size_t)cols + 1, 0);
    for (int32_t c : col) t.ptr[(size_t)c + 1]++;
    for (int64_t c = 0; c < cols; ++c) t.ptr[(size_t)c + 1] += t.ptr[(size_t)c];
    t.col.resize(val.size()); t.val.resize(val.size());
    std::vector<int64_t> pos(t.ptr.begin(), t.ptr.end() - 1);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t p = ptr[(size_t)r]; p < ptr[(size_t)r + 1]; ++p) {
            const int64_t q = pos[(size_t)col[(size_t)p]]++;
            t.col[(size_t)q] = (int32_t)r;
            t.val[(size_t)q] = val[(size_t)p];
        }
    return t;
}

// ---------------------------------------------------------------------------------------------
// Polynomial basis — Solver.cpp:2105-2149 (QUADRATIC_REGIONS)
// ---------------------------------------------------------------------------------------------
void buildConversionCoefficients(const double o[3], int axis, double v[RD]) {
    for (int n = 0; n < RD; ++n) v[n] = 0.;
#ifdef PS_AFFINE_REGIONS
    switch (axis) {   // AFFINE_REGIONS, exec/HDK_PolyStokesSolver.cpp:2153-2184
        case 0: v[0] = 1.; v[3] = o[0]; v[4] = o[1]; v[5] = o[2]; break;
        case 1: v[1] = 1.; v[6] = o[0]; v[7] = o[1]; v[8] = o[2]; break;
        case 2: v[2] = 1.; v[3] = -o[2]; v[7] = -o[2]; v[9] = o[0]; v[10] = o[1]; break;
    }
    return;
#endif
    switch (axis) {
        case 0:
            v[0] = 1.;
            v[3] = o[0]; v[4] = o[1]; v[5] = o[2];
            v[6] = o[0] * o[0]; v[7] = o[0] * o[1]; v[8] = o[0] * o[2];
            v[9] = o[1] * o[1]; v[10] = o[1] * o[2]; v[11] = o[2] * o[2];
            break;
        case 1:
            v[1] = 1.;
            v[12] = o[0]; v[13] = o[1]; v[14] = o[2];
            v[15] = o[0] * o[0]; v[16] = o[0] * o[1]; v[17] = o[0] * o[2];
            v[18] = o[1] * o[1]; v[19] = o[1] * o[2]; v[20] = o[2] * o[2];
            break;
        case 2:
            v[2] = 1.;
            v[3] = -o[2];
            v[6] = -2. * o[0] * o[2]; v[7] = -1. * o[1] * o[2]; v[8] = -0.5 * o[2] * o[2];
            v[13] = -o[2];
            v[16] = -1. * o[0] * o[2];
            v[18] = -2. * o[1] * o[2]; v[19] = -0.5 * o[2] * o[2];
            v[21] = o[0]; v[22] = o[1]; v[23] = o[0] * o[0];
            v[24] = o[0] * o[1]; v[25] = o[1] * o[1];
            break;
    }
}

// Eigen FullPivLU<Matrix<double,26,26>>::compute + solve (used at Solver.cpp:415).  Column-major
// max search with first-maximum tie-breaking, rank threshold eps*26*|maxpivot|.
bool fullPivLuSolve(const double* N, const double* rhs, double* x) {
    const int n = RD;
    double lu[RD][RD];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) lu[i][j] = N[i * n + j];
    int rt[RD], ct[RD];
    int nonzeroPivots = n;
    double maxpivot = 0.;
    for (int k = 0; k < n; ++k) {
        int pr = k, pc = k;
        double biggest = -1.;
        for (int j = k; j < n; ++j)
            for (int i = k; i < n; ++i) {
                const double a = std::fabs(lu[i][j]);
                if (a > biggest) { biggest = a; pr = i; pc = j; }
            }
        if (biggest == 0.) {
            nonzeroPivots = k;
            for (int i = k; i < n; ++i) { rt[i] = i; ct[i] = i; }
            break;
        }
        if (biggest > maxpivot) maxpivot = biggest;
        rt[k] = pr; ct[k] = pc;
        if (k != pr) for (int j = 0; j < n; ++j) std::swap(lu[k][j], lu[pr][j]);
        if (k != pc) for (int i = 0; i < n; ++i) std::swap(lu[i][k], lu[i][pc]);
        if (k < n - 1) for (int i = k + 1; i < n; ++i) lu[i][k] /= lu[k][k];
        if (k < n - 1)
            for (int i = k + 1; i < n; ++i)
                for (int j = k + 1; j < n; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
    }
    const double thresh = std::fabs(maxpivot) * (std::numeric_limits<double>::epsilon() * (double)n);
    int rank = 0;
    for (int i = 0; i < nonzeroPivots; ++i) rank += (std::fabs(lu[i][i]) > thresh);
    for (int i = 0; i < n; ++i) x[i] = 0.;
    if (rank == 0) return false;
    double c[RD];
    for (int i = 0; i < n; ++i) c[i] = rhs[i];
    for (int k = 0; k < n; ++k) std::swap(c[k], c[rt[k]]);
    for (int i = 0; i < n; ++i) { double s = c[i]; for (int j = 0; j < i; ++j) s -= lu[i][j] * c[j]; c[i] = s; }
    for (int i = rank - 1; i >= 0; --i) {
        double s = c[i];
        for (int j = i + 1; j < rank; ++j) s -= lu[i][j] * c[j];
        c[i] = s / lu[i][i];
    }
    int colperm[RD];
    for (int i = 0; i < n; ++i) colperm[i] = i;
    for (int k = 0; k < n; ++k) std::swap(colperm[k], colperm[ct[k]]);
    for (int i = 0; i < rank; ++i) x[colperm[i]] = c[i];
    return rank == n;
}

// Eigen Matrix<double,26,26>::inverse() = PartialPivLU(...).inverse() (AssembleBlocks.cpp:209).
// Same pivot choice; Eigen's blocked update order is not reproduced (tolerance-level parity).
bool partialPivInverse(const double* B, double* Binv) {
    const int n = RD;
    double lu[RD][RD];
    int perm[RD];
    for (int i = 0; i < n; ++i) { perm[i] = i; for (int j = 0; j < n; ++j) lu[i][j] = B[i * n + j]; }
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        int pr = k; double biggest = std::fabs(lu[k][k]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(lu[i][k]) > biggest) { biggest = std::fabs(lu[i][k]); pr = i; }
        if (biggest == 0.) { ok = false; continue; }
        if (pr != k) { for (int j = 0; j < n; ++j) std::swap(lu[k][j], lu[pr][j]); std::swap(perm[k], perm[pr]); }
        for (int i = k + 1; i < n; ++i) {
            lu[i][k] /= lu[k][k];
            const double f = lu[i][k];
            for (int j = k + 1; j < n; ++j) lu[i][j] -= f * lu[k][j];
        }
    }
    for (int c0 = 0; c0 < n; ++c0) {
        double y[RD];
        for (int i = 0; i < n; ++i) y[i] = (perm[i] == c0) ? 1. : 0.;
        for (int i = 0; i < n; ++i) { double s = y[i]; for (int j = 0; j < i; ++j) s -= lu[i][j] * y[j]; y[i] = s; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < n; ++j) s -= lu[i][j] * y[j]; y[i] = s / lu[i][i]; }
        for (int i = 0; i < n; ++i) Binv[i * n + c0] = y[i];
    }
    return ok;
}

int64_t Oracle::stressDOF(int64_t idx, int type) const {  // Solver.h:586-606: XX,YY,ZZ,YZ,XZ,XY
    switch (type) {
        case 0: return idx;
        case 1: return idx + nCenter;
        case 2: return idx + 2 * nCenter;
        case 3: return idx + 3 * nCenter;
        case 4: return idx + 3 * nCenter + nEdge[0];
        default: return idx + 3 * nCenter + nEdge[0] + nEdge[1];
    }
}
int64_t Oracle::faceVelocityDOF(int64_t idx, int axis) const {  // Solver.h:628-642
    return axis == 0 ? idx : (axis == 1 ? idx + nFace[0] : idx + nFace[0] + nFace[1]);
}

// Solver.cpp:328-372, 1274-1324
void Oracle::computeCenterOfMasses() {
    const int64_t R = regionCount;
    COM.assign((size_t)R * 3, 0.);
    std::vector<double> count((size_t)R, 0.);
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                if (!isReduced(labels[0].at(i, j, k))) continue;
                const int64_t r = reducedIdx[0].at(i, j, k);
                count[(size_t)r] += 1.;
                COM[(size_t)r * 3 + 0] += (double)i; COM[(size_t)r * 3 + 1] += (double)j; COM[(size_t)r * 3 + 2] += (double)k;
            }
    for (int64_t r = 0; r < R; ++r) {
        const double s = dx / count[(size_t)r];   // :370  COM *= dx / count
        for (int a = 0; a < 3; ++a) COM[(size_t)r * 3 + a] *= s;
    }
}

// Solver.cpp:374-417, 1330-1399
void Oracle::regionCellBoxes(std::vector<int32_t>& box) const {
    const int64_t R = regionCount;
    box.assign((size_t)R * 6, 0);
    for (int64_t r = 0; r < R; ++r) { box[(size_t)r * 6] = box[(size_t)r * 6 + 1] = box[(size_t)r * 6 + 2] = INT32_MAX; box[(size_t)r * 6 + 3] = box[(size_t)r * 6 + 4] = box[(size_t)r * 6 + 5] = -1; }
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const int64_t r = reducedIdx[0].at(i, j, k);
                if (r < 0) continue;
                int32_t* b = &box[(size_t)r * 6];
                b[0] = std::min(b[0], i); b[1] = std::min(b[1], j); b[2] = std::min(b[2], k);
                b[3] = std::max(b[3], i); b[4] = std::max(b[4], j); b[5] = std::max(b[5], k);
            }
}

// The per-tile sums below (least-squares systems, Mr, K) visit the grid in the reference's order and add into the tile a sample belongs to.
// Restricted to ONE tile that is a walk over the tile's box in the same nesting order — so the tiles are independent and each tile's sum
// sees its terms in the serial order: the loops run tile by tile over the tile's box (threads: Oracle::setupThreads), bit-identical to the
// single sweep over the whole grid they restate.
void Oracle::computeLeastSquaresFits() {
    const int64_t R = regionCount;
    std::vector<double> N((size_t)R * RD * RD, 0.), rhs((size_t)R * RD, 0.);
    std::vector<int32_t> box;
    regionCellBoxes(box);
#pragma omp parallel for schedule(dynamic, 4) num_threads(setupThreads > 1 ? setupThreads : 1)
    for (int64_t r = 0; r < R; ++r) {
        const int32_t* b = &box[(size_t)r * 6];
        for (int k = b[2]; k <= b[5]; ++k)
            for (int j = b[1]; j <= b[4]; ++j)
                for (int i = b[0]; i <= b[3]; ++i) {
                    if (reducedIdx[0].at(i, j, k) != r) continue;
                    for (int axis = 0; axis < 3; ++axis)
                        for (int dir = 0; dir < 2; ++dir) {
                            int a[3] = {i, j, k};
                            a[axis] += dir ? 1 : -1;
                            if (!isActive(labels[0].getConst(a[0], a[1], a[2], PS_UNASSIGNED))) continue;  // :1375
                            double off[3] = {(double)i, (double)j, (double)k};
                            off[axis] += dir == 0 ? -.5 : .5;
                            for (int q = 0; q < 3; ++q) { off[q] *= dx; off[q] -= COM[(size_t)r * 3 + q]; }
                            double C[RD];
                            buildConversionCoefficients(off, axis, C);
                            int f[3] = {i, j, k};
                            f[axis] += dir;
                            const double uval = (double)vel[axis].at(f[0], f[1], f[2]);
                            double* Nr = &N[(size_t)r * RD * RD];
                            for (int m = 0; m < RD; ++m) {
                                for (int n = 0; n < RD; ++n) Nr[m * RD + n] += C[m] * C[n];
                                rhs[(size_t)r * RD + m] += uval * C[m];
                            }
                        }
                }
    }
    cfit.assign((size_t)R * RD, 0.);
#pragma omp parallel for schedule(dynamic, 4) num_threads(setupThreads > 1 ? setupThreads : 1)
    for (int64_t r = 0; r < R; ++r) fullPivLuSolve(&N[(size_t)r * RD * RD], &rhs[(size_t)r * RD], &cfit[(size_t)r * RD]);
    fitN = N; fitRhs = rhs;   // kept for diagnostics (conditioning of the per-tile normal systems)
}

// Solver.cpp:419-441, 1405-1482
void Oracle::computeReducedMassMatrices() {
    const int64_t R = regionCount;
    Mr.assign((size_t)R * RD * RD, 0.);
    std::vector<int32_t> box;
    regionCellBoxes(box);
#pragma omp parallel for schedule(dynamic, 4) num_threads(setupThreads > 1 ? setupThreads : 1)
    for (int64_t r = 0; r < R; ++r) {
        const int32_t* b = &box[(size_t)r * 6];
        for (int k = b[2]; k <= b[5]; ++k)
            for (int j = b[1]; j <= b[4]; ++j)
                for (int i = b[0]; i <= b[3]; ++i) {
                    if (reducedIdx[0].at(i, j, k) != r) continue;
                    for (int axis = 0; axis < 3; ++axis)
                        for (int dir = 0; dir < 2; ++dir) {
                            bool doApplyFace = false;
                            if (dir == 0) doApplyFace = true;
                            else {
                                int a[3] = {i, j, k};
                                a[axis] += 1;
                                if (isActive(labels[0].getConst(a[0], a[1], a[2], PS_UNASSIGNED))) doApplyFace = true;
                            }
                            if (!doApplyFace) continue;
                            double off[3] = {(double)i, (double)j, (double)k};
                            off[axis] += dir == 0 ? -.5 : .5;
                            for (int q = 0; q < 3; ++q) { off[q] *= dx; off[q] -= COM[(size_t)r * 3 + q]; }
                            double C[RD];
                            buildConversionCoefficients(off, axis, C);
                            double* M = &Mr[(size_t)r * RD * RD];
                            for (int m = 0; m < RD; ++m)
                                for (int n = 0; n < RD; ++n) M[m * RD + n] += rho * C[m] * C[n];  // :1471 (rho*col)*row
                        }
                }
    }
}

// Solver.cpp:468-490, 1484-1694
void Oracle::computeReducedViscosityMatricesInteriorOnly() {
    const int64_t R = regionCount;
    K.assign((size_t)R * RD * RD, 0.);
    const Dim cd = centerDim();
    std::vector<int32_t> box;
    regionCellBoxes(box);
    // (a face carries the tile of the cell at the face or of the cell below it along its axis, Classifier.cpp:1473-1528: the tile's faces lie in
    // the cells' box grown by one — walked per tile, axis by axis, in the serial nesting order: see computeLeastSquaresFits)
#pragma omp parallel for schedule(dynamic, 1) num_threads(setupThreads > 1 ? setupThreads : 1)
    for (int64_t tile = 0; tile < R; ++tile)
    for (int faceAxis = 0; faceAxis < 3; ++faceAxis) {
        const Dim fd = faceDim(faceAxis);
        const Field<int32_t>& tileFaceIndex = reducedIdx[1 + faceAxis];
        const int32_t* tb = &box[(size_t)tile * 6];
        if (tb[3] < 0) continue;
        for (int k = std::max(0, tb[2] - 1); k <= std::min(fd.n[2] - 1, tb[5] + 1); ++k)
            for (int j = std::max(0, tb[1] - 1); j <= std::min(fd.n[1] - 1, tb[4] + 1); ++j)
                for (int i = std::max(0, tb[0] - 1); i <= std::min(fd.n[0] - 1, tb[3] + 1); ++i) {
                    const int64_t self = tileFaceIndex.at(i, j, k);
                    if (self != tile) continue;
                    double selfOff[3] = {(double)i, (double)j, (double)k};
                    selfOff[faceAxis] -= 0.5;
                    for (int q = 0; q < 3; ++q) { selfOff[q] *= dx; selfOff[q] -= COM[(size_t)self * 3 + q]; }
                    double colVec[RD];
                    buildConversionCoefficients(selfOff, faceAxis, colVec);
                    double* Kr = &K[(size_t)self * RD * RD];
                    // cell-centred stress terms :1538-1603
                    for (int divDir = 0; divDir < 2; ++divDir) {
                        int c[3] = {i, j, k};
                        c[faceAxis] += divDir - 1;
                        if (!isReduced(labels[0].getConst(c[0], c[1], c[2], PS_UNASSIGNED))) continue;
                        if (c[faceAxis] < 0 || c[faceAxis] >= fd.n[faceAxis]) continue;
                        const double divSign = divDir == 0 ? -1. : 1.;
                        const double visc = (double)localViscosityAtCell(c[0], c[1], c[2]);
                        for (int gradDir = 0; gradDir < 2; ++gradDir) {
                            int af[3] = {c[0], c[1], c[2]};
                            af[faceAxis] += gradDir;
                            const double gradSign = gradDir == 0 ? -1. : 1.;
                            const double contribution = -1. * divSign * gradSign * visc / (dx * dx);
                            const int64_t adj = tileFaceIndex.getConst(af[0], af[1], af[2], PS_UNASSIGNED);
                            if (adj < 0) continue;
                            double adjOff[3] = {(double)af[0], (double)af[1], (double)af[2]};
                            adjOff[faceAxis] -= 0.5;
                            for (int q = 0; q < 3; ++q) { adjOff[q] *= dx; adjOff[q] -= COM[(size_t)adj * 3 + q]; }
                            double rowVec[RD];
                            buildConversionCoefficients(adjOff, faceAxis, rowVec);
                            for (int m = 0; m < RD; ++m)
                                for (int n = 0; n < RD; ++n) Kr[m * RD + n] += contribution * colVec[m] * rowVec[n];
                        }
                    }
                    // edge-centred stress terms :1606-1683
                    for (int edgeAxis = 0; edgeAxis < 3; ++edgeAxis) {
                        if (edgeAxis == faceAxis) continue;
                        for (int divDir = 0; divDir < 2; ++divDir) {
                            const double divSign = divDir == 0 ? -1. : 1.;
                            int e[3] = {i, j, k};
                            e[3 - faceAxis - edgeAxis] += divDir;
                            const float visc = localViscosityAtEdge(edgeAxis, e[0], e[1], e[2]);  // float, :1617
                            if (labels[4 + edgeAxis].getConst(e[0], e[1], e[2], PS_UNASSIGNED) != PS_REDUCED) continue;
                            for (int gradAxis = 0; gradAxis < 3; ++gradAxis) {
                                if (gradAxis == edgeAxis) continue;
                                const int adjFaceAxis = 3 - gradAxis - edgeAxis;
                                const Field<int32_t>& adjIndex = reducedIdx[1 + adjFaceAxis];
                                for (int gradDir = 0; gradDir < 2; ++gradDir) {
                                    int af[3] = {e[0], e[1], e[2]};
                                    af[gradAxis] += gradDir - 1;  // edgeToFaceMap
                                    const double gradSign = gradDir == 0 ? -1. : 1.;
                                    const double contribution = -0.5 * divSign * gradSign * visc / (dx * dx);
                                    const int64_t adj = adjIndex.getConst(af[0], af[1], af[2], PS_UNASSIGNED);
                                    if (adj < 0) continue;
                                    double adjOff[3] = {(double)af[0], (double)af[1], (double)af[2]};
                                    adjOff[adjFaceAxis] -= 0.5;
                                    for (int q = 0; q < 3; ++q) { adjOff[q] *= dx; adjOff[q] -= COM[(size_t)adj * 3 + q]; }
                                    double rowVec[RD];
                                    buildConversionCoefficients(adjOff, adjFaceAxis, rowVec);
                                    for (int m = 0; m < RD; ++m)
                                        for (int n = 0; n < RD; ++n) Kr[m * RD + n] += contribution * colVec[m] * rowVec[n];
                                }
                            }
                        }
                    }
                }
    }
    (void)cd;
}

// ConstructMatrixBlocks.cpp:9-292 (sizes, triplets -> CSR) and :294-868 (the sweeps)
void Oracle::constructMatrixBlocks() {
    nActiveVs = nFace[0] + nFace[1] + nFace[2];
    nReducedVs = regionCount * RD;
    nPressures = nCenter;
    nStresses = 3 * nCenter + nEdge[0] + nEdge[1] + nEdge[2];
    nReducedStresses = regionCount * 6;
    nTotalDOFs = nActiveVs + nReducedVs + nPressures + nStresses;

    std::vector<Trip> tMc, tMcInv, tRhsA, tOld, tG, tJG, tDt, tJDt, tRhsP, tRhsT, tUInv, tU;
    const double MINWEIGHT = 0.1;
    const Dim cd = centerDim();
    // Threads (setupThreads > 1): every sweep below is cut into `pieces` contiguous ranges of the serial traversal; a piece fills its own triplet
    // lists and the lists are appended in piece order — the global lists hold the triplets in EXACTLY the serial order, so the sums over
    // duplicates (toVec, fromTriplets: stable sort) add in the serial order too.
    struct TripSet { std::vector<Trip> tMc, tMcInv, tRhsA, tOld, tG, tJG, tDt, tJDt, tRhsP, tRhsT, tUInv, tU; };
    const int pieces = setupThreads > 1 ? setupThreads * 4 : 1;
    auto gatherSets = [&](std::vector<TripSet>& sets) {
        std::vector<Trip> TripSet::*member[12] = {&TripSet::tMc, &TripSet::tMcInv, &TripSet::tRhsA, &TripSet::tOld, &TripSet::tG, &TripSet::tJG, &TripSet::tDt,
                                                  &TripSet::tJDt, &TripSet::tRhsP, &TripSet::tRhsT, &TripSet::tUInv, &TripSet::tU};
        std::vector<Trip>* dst[12] = {&tMc, &tMcInv, &tRhsA, &tOld, &tG, &tJG, &tDt, &tJDt, &tRhsP, &tRhsT, &tUInv, &tU};
        for (int q = 0; q < 12; ++q) {
            size_t add = 0;
            for (TripSet& S : sets) add += (S.*member[q]).size();
            dst[q]->reserve(dst[q]->size() + add);
            for (TripSet& S : sets) {
                std::vector<Trip>& v = S.*member[q];
                dst[q]->insert(dst[q]->end(), v.begin(), v.end());
                std::vector<Trip>().swap(v);
            }
        }
    };

    for (int faceAxis = 0; faceAxis < 3; ++faceAxis) {   // :320-648
        const Dim fd = faceDim(faceAxis);
        std::vector<TripSet> sets((size_t)pieces);
        forEachOrderedPieces(fd, pieces, [&](int pc, int i, int j, int k) {
            TripSet& T = sets[(size_t)pc];
            const int32_t selfLabel = labels[1 + faceAxis].at(i, j, k);
            const int64_t selfActiveIndex = faceVelocityDOF(activeIdx[1 + faceAxis].at(i, j, k), faceAxis);
            const int64_t selfReducedIndex = reducedIdx[1 + faceAxis].at(i, j, k);
            const double localDensity = rho;
            const double wF = (double)fluidW[1 + faceAxis].at(i, j, k);
            double volume = wF * (double)liquidW[1 + faceAxis].at(i, j, k);
            volume = clampd(volume, MINWEIGHT * MINWEIGHT, 1.0);
            const double localVelocity = (double)vel[faceAxis].at(i, j, k);
            if (isActive(selfLabel)) {   // :369-391
                T.tMc.push_back(mkT(selfActiveIndex, selfActiveIndex, volume * localDensity));
                T.tMcInv.push_back(mkT(selfActiveIndex, selfActiveIndex, 1. / (volume * localDensity)));
                T.tRhsA.push_back(mkT(selfActiveIndex, 0, localVelocity * volume * localDensity));
                T.tOld.push_back(mkT(selfActiveIndex, 0, localVelocity));
            }
            if (!(isActive(selfLabel) || isReduced(selfLabel))) return;
            double colVec[RD];
            if (!isActive(selfLabel)) {
                double off[3] = {(double)i, (double)j, (double)k};
                off[faceAxis] -= 0.5;
                for (int q = 0; q < 3; ++q) { off[q] *= dx; off[q] -= COM[(size_t)selfReducedIndex * 3 + q]; }
                buildConversionCoefficients(off, faceAxis, colVec);
            }
            const double svel = (double)collisionvel[faceAxis].at(i, j, k);
            // pressure stencils :394-460
            for (int gradDir = 0; gradDir < 2; ++gradDir) {
                const double gradSign = gradDir == 0 ? -1. : 1.;
                int c[3] = {i, j, k};
                c[faceAxis] += gradDir - 1;
                if (c[faceAxis] < 0 || c[faceAxis] >= cd.n[faceAxis]) continue;
                const int64_t cellPressureIndex = activeIdx[0].at(c[0], c[1], c[2]);
                if (cellPressureIndex < 0) continue;
                const double coeff = wF * (double)liquidW[0].at(c[0], c[1], c[2]) * invDx;
                const double contribution = gradSign * coeff;
                if (coeff <= 0.) continue;
                if (isActive(selfLabel)) {
                    T.tG.push_back(mkT(selfActiveIndex, cellPressureIndex, contribution));
                    if (fluidW[0].at(c[0], c[1], c[2]) < 1.f) {   // :424-432 (solidCoeff unused)
                        const double solidContribution = gradSign * coeff;
                        T.tRhsP.push_back(mkT(cellPressureIndex, 0, -1. * solidContribution * svel));
                    }
                    if (fluidW[1 + faceAxis].at(i, j, k) < 1.f) {  // :433-441
                        const double solidContribution = gradSign * coeff;
                        T.tRhsP.push_back(mkT(cellPressureIndex, 0, solidContribution * svel));
                    }
                } else {
                    for (int n = 0; n < RD; ++n) T.tJG.push_back(mkT(RD * selfReducedIndex + n, cellPressureIndex, contribution * colVec[n]));
                }
            }
            // stress stencils, centres :466-550
            for (int divDir = 0; divDir < 2; ++divDir) {
                const double divSign = divDir == 0 ? -1. : 1.;
                int c[3] = {i, j, k};
                c[faceAxis] += divDir - 1;
                if (c[faceAxis] < 0 || c[faceAxis] >= cd.n[faceAxis]) continue;
                const int32_t cellLabel = labels[0].at(c[0], c[1], c[2]);
                const int64_t cellStressIndex = stressDOF(activeIdx[0].at(c[0], c[1], c[2]), faceAxis);
                if (!isActive(cellLabel)) continue;
                const double coeff = wF * (double)liquidW[0].at(c[0], c[1], c[2]) * invDx;
                const double contribution = -1. * divSign * coeff;
                if (coeff <= 0.) continue;
                if (isActive(selfLabel)) {
                    T.tDt.push_back(mkT(selfActiveIndex, cellStressIndex, contribution));
                    if (fluidW[0].at(c[0], c[1], c[2]) < 1.f) {
                        const double solidContribution = divSign * coeff;
                        T.tRhsT.push_back(mkT(cellStressIndex, 0, -1. * solidContribution * svel));
                    }
                    if (fluidW[1 + faceAxis].at(i, j, k) < 1.f) {
                        const double solidContribution = divSign * coeff;
                        T.tRhsT.push_back(mkT(cellStressIndex, 0, solidContribution * svel));
                    }
                } else {
                    for (int n = 0; n < RD; ++n) T.tJDt.push_back(mkT(RD * selfReducedIndex + n, cellStressIndex, contribution * colVec[n]));
                }
            }
            // stress stencils, edges :553-639
            for (int edgeAxis = 0; edgeAxis < 3; ++edgeAxis) {
                if (edgeAxis == faceAxis) continue;
                for (int divDir = 0; divDir < 2; ++divDir) {
                    const double divSign = divDir == 0 ? -1. : 1.;
                    int e[3] = {i, j, k};
                    e[3 - faceAxis - edgeAxis] += divDir;
                    const int32_t edgeLabel = labels[4 + edgeAxis].at(e[0], e[1], e[2]);
                    const int64_t edgeStressIndex = stressDOF(activeIdx[4 + edgeAxis].at(e[0], e[1], e[2]), 3 + edgeAxis);
                    if (!isActive(edgeLabel)) continue;
                    const double coeff = wF * (double)liquidW[4 + edgeAxis].at(e[0], e[1], e[2]) * invDx;
                    const double contribution = -1. * divSign * coeff;
                    if (coeff <= 0.) continue;
                    if (isActive(selfLabel)) {
                        T.tDt.push_back(mkT(selfActiveIndex, edgeStressIndex, contribution));
                        if (fluidW[4 + edgeAxis].at(e[0], e[1], e[2]) < 1.f) {
                            const double solidContribution = divSign * coeff;
                            T.tRhsT.push_back(mkT(edgeStressIndex, 0, -1. * solidContribution * svel));
                        }
                        if (fluidW[1 + faceAxis].at(i, j, k) < 1.f) {
                            const double solidContribution = divSign * coeff;
                            T.tRhsT.push_back(mkT(edgeStressIndex, 0, solidContribution * svel));
                        }
                    } else {
                        for (int n = 0; n < RD; ++n) T.tJDt.push_back(mkT(RD * selfReducedIndex + n, edgeStressIndex, contribution * colVec[n]));
                    }
                }
            }
        });
        gatherSets(sets);
    }
    // edge stress diagonal :651-735
    for (int edgeAxis = 0; edgeAxis < 3; ++edgeAxis) {
        const Dim ed = edgeDim(edgeAxis);
        std::vector<TripSet> sets((size_t)pieces);
        forEachOrderedPieces(ed, pieces, [&](int pc, int i, int j, int k) {
            TripSet& T = sets[(size_t)pc];
            const int32_t edgeLabel = labels[4 + edgeAxis].at(i, j, k);
            if (!isActive(edgeLabel)) return;
            const int64_t edgeStressIndex = stressDOF(activeIdx[4 + edgeAxis].at(i, j, k), 3 + edgeAxis);
            const double volumeWeight = clampd((double)fluidW[4 + edgeAxis].at(i, j, k), MINWEIGHT, 1.0) * (double)liquidW[4 + edgeAxis].at(i, j, k);
            const double localViscosity = (double)localViscosityAtEdge(edgeAxis, i, j, k);
            const double invLocalViscosity = clampd(1. / localViscosity, 0., 1e10);
            T.tUInv.push_back(mkT(edgeStressIndex, edgeStressIndex, 2. * invLocalViscosity * volumeWeight));
            T.tU.push_back(mkT(edgeStressIndex, edgeStressIndex, 0.5 * localViscosity * clampd(1. / volumeWeight, 0., 1.e2)));
        });
        gatherSets(sets);
    }
    // centre stress diagonal :737-867
    std::vector<TripSet> csets((size_t)pieces);
    forEachOrderedPieces(cd, pieces, [&](int pc, int i, int j, int k) {
        TripSet& T = csets[(size_t)pc];
        if (!isActive(labels[0].at(i, j, k))) return;
        const int64_t ci = activeIdx[0].at(i, j, k);
        const double volumeWeight = clampd((double)fluidW[0].at(i, j, k), MINWEIGHT, 1.0) * (double)liquidW[0].at(i, j, k);
        const double localViscosity = (double)localViscosityAtCell(i, j, k);
        const double invLocalViscosity = clampd(1. / localViscosity, 0., 1.e10);
        for (int t = 0; t < 3; ++t) {
            T.tUInv.push_back(mkT(stressDOF(ci, t), stressDOF(ci, t), invLocalViscosity * clampd(volumeWeight, 1.e-2, 1.)));
            T.tU.push_back(mkT(stressDOF(ci, t), stressDOF(ci, t), localViscosity * clampd(1. / volumeWeight, 0., 1.e2)));
        }
    });
    gatherSets(csets);

    auto toVec = [](std::vector<Trip>& t, int64_t n, std::vector<double>& out, bool diag) {
        out.assign((size_t)n, 0.);
        for (const Trip& e : t) out[(size_t)e.r] += e.v;
        (void)diag;
    };
    toVec(tMc, nActiveVs, Mc, true);
    toVec(tMcInv, nActiveVs, McInv, true);
    toVec(tRhsA, nActiveVs, activeRHS, false);
    toVec(tOld, nActiveVs, oldActiveVs, false);
    toVec(tRhsP, nPressures, pressureRHS, false);
    toVec(tRhsT, nStresses, stressRHS, false);
    toVec(tUInv, nStresses, uInv, true);
    toVec(tU, nStresses, u, true);
    G.fromTriplets(nActiveVs, nPressures, tG, setupThreads);
    Dt.fromTriplets(nActiveVs, nStresses, tDt, setupThreads);
    JG.fromTriplets(nReducedVs, nPressures, tJG, setupThreads);
    JDt.fromTriplets(nReducedVs, nStresses, tJDt, setupThreads);
}

// AssembleBlocks.cpp:147-244,356-367 + AssembleSystem.cpp:432-470
void Oracle::assembleSystemPressureStressFactored() {
    nSystemSize = nPressures + nStresses;
    const int64_t R = regionCount;
    Binv.assign((size_t)R * RD * RD, 0.);
    reducedRHS.assign((size_t)R * RD, 0.);
    for (int64_t r = 0; r < R; ++r) {
        double B[RD * RD];
        for (int q = 0; q < RD * RD; ++q) B[q] = invDt * Mr[(size_t)r * RD * RD + q] + 2. * K[(size_t)r * RD * RD + q];
        partialPivInverse(B, &Binv[(size_t)r * RD * RD]);
        for (int m = 0; m < RD; ++m) {   // assembleReducedRHSVector: Mr * c_fit
            double s = 0;
            for (int n = 0; n < RD; ++n) s += Mr[(size_t)r * RD * RD + m * RD + n] * cfit[(size_t)r * RD + n];
            reducedRHS[(size_t)r * RD + m] = s;
        }
    }
    Gt = G.transposed();
    D = Dt.transposed();
    // b1 = -G^T McInv rhs_a - invDt JG^T BInv rhs_r ; b2 likewise (AssembleSystem.cpp:448-459)
    std::vector<double> t1((size_t)nActiveVs), t2((size_t)nReducedVs, 0.);
    for (int64_t f = 0; f < nActiveVs; ++f) t1[(size_t)f] = McInv[(size_t)f] * activeRHS[(size_t)f];
    for (int64_t r = 0; r < R; ++r)
        for (int m = 0; m < RD; ++m) {
            double s = 0;
            for (int n = 0; n < RD; ++n) s += Binv[(size_t)r * RD * RD + m * RD + n] * reducedRHS[(size_t)r * RD + n];
            t2[(size_t)r * RD + m] = s;
        }
    std::vector<double> gp((size_t)nPressures, 0.), gt((size_t)nStresses, 0.), jp((size_t)nPressures, 0.), jt((size_t)nStresses, 0.);
    G.mulT_add(t1.data(), gp.data());
    Dt.mulT_add(t1.data(), gt.data());
    JG.mulT_add(t2.data(), jp.data());
    JDt.mulT_add(t2.data(), jt.data());
    b.assign((size_t)nSystemSize, 0.);
    for (int64_t i = 0; i < nPressures; ++i) b[(size_t)i] = (-gp[(size_t)i] - invDt * jp[(size_t)i]) + pressureRHS[(size_t)i];
    for (int64_t i = 0; i < nStresses; ++i) b[(size_t)(nPressures + i)] = (-gt[(size_t)i] - invDt * jt[(size_t)i]) + stressRHS[(size_t)i];
    solution.assign((size_t)nSystemSize, 0.);
}

// AssembleSystem.cpp:351-430: explicit A = -dt [G Dt]^T McInv [G Dt] - [JG JDt]^T BInv [JG JDt] - 1/2 uInv.
// Small cases only (the reduced term is a dense clique per tile).
void Oracle::assembleSystemPressureStress() {
    const int64_t n = nPressures + nStresses;
    std::vector<std::map<int32_t, double>> rowsA((size_t)n);
    for (int64_t f = 0; f < nActiveVs; ++f) {
        std::vector<std::pair<int32_t, double>> ent;
        for (int64_t p = G.ptr[(size_t)f]; p < G.ptr[(size_t)f + 1]; ++p) ent.push_back({G.col[(size_t)p], G.val[(size_t)p]});
        for (int64_t p = Dt.ptr[(size_t)f]; p < Dt.ptr[(size_t)f + 1]; ++p) ent.push_back({(int32_t)(nPressures + Dt.col[(size_t)p]), Dt.val[(size_t)p]});
        for (auto& a : ent) for (auto& c : ent) rowsA[(size_t)a.first][c.first] += -dt * a.second * McInv[(size_t)f] * c.second;
    }
    for (int64_t r = 0; r < regionCount; ++r) {
        std::vector<std::vector<std::pair<int32_t, double>>> ent(RD);
        for (int m = 0; m < RD; ++m) {
            const int64_t row = r * RD + m;
            for (int64_t p = JG.ptr[(size_t)row]; p < JG.ptr[(size_t)row + 1]; ++p) ent[m].push_back({JG.col[(size_t)p], JG.val[(size_t)p]});
            for (int64_t p = JDt.ptr[(size_t)row]; p < JDt.ptr[(size_t)row + 1]; ++p) ent[m].push_back({(int32_t)(nPressures + JDt.col[(size_t)p]), JDt.val[(size_t)p]});
        }
        for (int m = 0; m < RD; ++m)
            for (int q = 0; q < RD; ++q) {
                const double bi = Binv[(size_t)r * RD * RD + m * RD + q];
                if (bi == 0.) continue;
                for (auto& a : ent[m]) for (auto& c : ent[q]) rowsA[(size_t)a.first][c.first] += -a.second * bi * c.second;
            }
    }
    for (int64_t i = 0; i < nStresses; ++i) rowsA[(size_t)(nPressures + i)][(int32_t)(nPressures + i)] += -0.5 * uInv[(size_t)i];
    A.rows = A.cols = n;
    A.ptr.assign((size_t)n + 1, 0);
    A.col.clear(); A.val.clear();
    for (int64_t i = 0; i < n; ++i) {
        A.ptr[(size_t)i] = (int64_t)A.val.size();
        for (auto& kv : rowsA[(size_t)i]) { A.col.push_back(kv.first); A.val.push_back(kv.second); }
    }
    A.ptr[(size_t)n] = (int64_t)A.val.size();
}

// Jacobi extension (reference stub Preconditioners.cpp:37-41): diag(A) of the product-form operator.
void Oracle::buildJacobiDiagonal() {
    const int64_t n = nPressures + nStresses;
    diagA.assign((size_t)n, 0.);
    for (int64_t f = 0; f < nActiveVs; ++f) {
        for (int64_t p = G.ptr[(size_t)f]; p < G.ptr[(size_t)f + 1]; ++p)
            diagA[(size_t)G.col[(size_t)p]] += -dt * McInv[(size_t)f] * G.val[(size_t)p] * G.val[(size_t)p];
        for (int64_t p = Dt.ptr[(size_t)f]; p < Dt.ptr[(size_t)f + 1]; ++p)
            diagA[(size_t)(nPressures + Dt.col[(size_t)p])] += -dt * McInv[(size_t)f] * Dt.val[(size_t)p] * Dt.val[(size_t)p];
    }
    for (int64_t r = 0; r < regionCount; ++r) {
        std::map<int32_t, std::vector<double>> q;   // column -> 26-vector of [JG JDt](:,col) restricted to region r
        for (int m = 0; m < RD; ++m) {
            const int64_t row = r * RD + m;
            for (int64_t p = JG.ptr[(size_t)row]; p < JG.ptr[(size_t)row + 1]; ++p) {
                auto& v = q[JG.col[(size_t)p]]; if (v.empty()) v.assign(RD, 0.); v[(size_t)m] += JG.val[(size_t)p];
            }
            for (int64_t p = JDt.ptr[(size_t)row]; p < JDt.ptr[(size_t)row + 1]; ++p) {
                auto& v = q[(int32_t)(nPressures + JDt.col[(size_t)p])]; if (v.empty()) v.assign(RD, 0.); v[(size_t)m] += JDt.val[(size_t)p];
            }
        }
        for (auto& kv : q) {
            double s = 0;
            for (int m = 0; m < RD; ++m) {
                double t = 0;
                for (int k2 = 0; k2 < RD; ++k2) t += Binv[(size_t)r * RD * RD + m * RD + k2] * kv.second[(size_t)k2];
                s += kv.second[(size_t)m] * t;
            }
            diagA[(size_t)kv.first] -= s;
        }
    }
    for (int64_t i = 0; i < nStresses; ++i) diagA[(size_t)(nPressures + i)] += -0.5 * uInv[(size_t)i];
}

}  // namespace psoracle
