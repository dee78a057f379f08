// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU oracle: a literal, single-threaded restatement of the hot path of
// panuelosj/polystokes (the per-step reduced-viscosity Stokes solve) on dense arrays,
// with no Eigen / HDK / TBB.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, link, import or call anything in this directory.
//
// PARITY PINNING: the reference ships no tests, golden vectors or exported systems
//  and cannot be compiled here (needs Houdini HDK + TBB; vendored Eigen lacks
// Eigen/Core).  The oracle is therefore pinned by the analytic known-answer tests listed in
// SURVEY.md §8c (tests/test_oracle_kat.py) — "parity unpinned" with respect to reference
// fixtures, and with respect to the out-of-tree HDK pieces it has to restate from their call
// sites: computeSDFWeightsSampled, SIM_VolumetricConnectedComponentBuilder numbering,
// UT_VoxelArray tile traversal order and border modes, SIM_RawField::getValue trilinear.
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
#pragma once
#include <cstdint>
#include <cstddef>
#include <map>
#include <string>
#include <vector>

#include "../include/polystokes.h"

namespace psoracle {

constexpr int RD = PS_REDUCED_DOF;

struct Dim {
    int n[3] = {0, 0, 0};
    int64_t size() const { return (int64_t)n[0] * n[1] * n[2]; }
    int64_t lin(int i, int j, int k) const { return i + (int64_t)n[0] * (j + (int64_t)n[1] * k); }
    bool oob(int i, int j, int k) const {
        return i < 0 || i >= n[0] || j < 0 || j >= n[1] || k < 0 || k >= n[2];
    }
};

template <class T>
struct Field {
    Dim d;
    std::vector<T> v;
    void init(const Dim& dd, T c) { d = dd; v.assign((size_t)dd.size(), c); }
    T& at(int i, int j, int k) { return v[(size_t)d.lin(i, j, k)]; }
    const T& at(int i, int j, int k) const { return v[(size_t)d.lin(i, j, k)]; }
    // UT_VOXELBORDER_CONSTANT (label/index fields, Solver.cpp:101-152)
    T getConst(int i, int j, int k, T border) const { return d.oob(i, j, k) ? border : at(i, j, k); }
    // UT_VOXELBORDER_STREAK (UT_VoxelArray default; weight fields)
    T getStreak(int i, int j, int k) const {
        i = i < 0 ? 0 : (i >= d.n[0] ? d.n[0] - 1 : i);
        j = j < 0 ? 0 : (j >= d.n[1] ? d.n[1] - 1 : j);
        k = k < 0 ? 0 : (k >= d.n[2] ? d.n[2] - 1 : k);
        return at(i, j, k);
    }
};

struct Trip {          // 16 bytes (r06: int32 indices — the 256^3 system has ~0.9 G of them before duplicates are summed)
    int32_t r, c;
    double v;
};

struct CSR {
    int64_t rows = 0, cols = 0;
    std::vector<int64_t> ptr;
    std::vector<int32_t> col;
    std::vector<double> val;
    int64_t nnz() const { return (int64_t)val.size(); }
    // Eigen setFromTriplets semantics (SparseMatrix.h:1025-1054,1122-1160): duplicates summed,
    // compressed, inner indices sorted.
    void fromTriplets(int64_t rows_, int64_t cols_, std::vector<Trip>& t, int threads = 1);
    void mul(const double* x, double* y) const;            // y = M x
    void mulT_add(const double* x, double* y) const;       // y += M^T x
    CSR transposed() const;
};

struct ArrayRef {
    const void* ptr;
    int64_t count;
    int32_t elem;
};

struct Oracle {
    // ---- inputs ----
    ps_params P;
    int nx = 0, ny = 0, nz = 0;
    double dx = 0, invDx = 0, dt = 0, invDt = 0;
    double rho = 0;
    Field<float> surface, collision, viscosity;
    Field<float> vel[3], collisionvel[3];
    std::string err;

    // ---- weights (Solver.h:316-322) index: 0 center, 1..3 face X/Y/Z, 4..6 edge YZ/XZ/XY (edge axis 0,1,2)
    Field<float> liquidW[7], fluidW[7];
    // ---- labels / indices (Solver.h:327-335), same indexing
    Field<int32_t> labels[7], activeIdx[7], reducedIdx[7];

    // counts (Solver.h:272-285)
    int64_t nCenter = 0, nFace[3] = {0, 0, 0}, nEdge[3] = {0, 0, 0};  // nEdge[edgeAxis]: 0=YZ 1=XZ 2=XY
    int64_t nActiveVs = 0, nReducedVs = 0, nPressures = 0, nStresses = 0, nReducedStresses = 0;
    int64_t nTotalDOFs = 0, nSystemSize = 0;
    int64_t regionCount = 0;

    // per-region data
    std::vector<double> COM;        // R*3
    std::vector<double> cfit;       // R*26
    std::vector<double> Mr, K, Binv; // R*26*26 row-major
    std::vector<double> fitN, fitRhs; // the per-tile least-squares normal systems (diagnostics)
    std::vector<double> reducedRHS; // R*26

    // blocks
    std::vector<double> Mc, McInv, uInv, u;
    std::vector<double> activeRHS, pressureRHS, stressRHS, oldActiveVs;
    CSR G, Dt, JG, JDt;
    CSR Gt, D;  // explicit transposes (ApplyPressureStressMatrix.h:42-45)
    CSR A;      // explicit operator (AssembleSystem.cpp:351-430), optional
    std::vector<double> diagA; // Jacobi extension
    // Threads of the SETUP sweeps (po_set_setup_threads; default 1 = the literal serial code).  The reference's own setup fans out over all cores
    // (exec/HDK_PolyStokesSolver.cpp:154); bench.py's CPU leg uses this to build the 256^3 system in seconds instead of minutes.  Every threaded
    // sweep produces the SAME bits as the serial one (tests/test_oracle_kat.py asserts it): per-tile sums run whole on one thread in the serial
    // order; triplets are generated per contiguous piece of the serial traversal and concatenated in order; the sort is a stable parallel sort.
    int setupThreads = 1;
    bool exactDiagonal = false; // Jacobi / Chebyshev extensions on 1 / A_jj in fp64 instead of the product's 16-bit storage form (ps_oracle_solve.cpp: storedDinv)
    std::vector<double> b, solution, recovered, guess;
    Field<float> velOut[3], valid[3];

    ps_stats stats;
    int solveIterations = -1;
    double solveError = -1;

    std::map<std::string, ArrayRef> arrays;

    // ---- pipeline (names = reference Solver methods) ----
    int load(const ps_params* p, const ps_fields_in* in);
    void buildIntegrationWeightsAlt(const ps_fields_in* in);
    void classifyCells();
    void constructReducedRegions();
    void constructOnlyActiveRegions();
    void classifyFaces();
    void classifyEdges();
    void constructCenterReducedIndices();
    void constructFacesReducedIndices();
    void constructEdgesReducedIndices();
    void constructActiveIndices();
    void computeCenterOfMasses();
    void computeLeastSquaresFits();
    void computeReducedMassMatrices();
    void computeReducedViscosityMatricesInteriorOnly();
    void constructMatrixBlocks();
    void assembleSystemPressureStressFactored();
    void assembleSystemPressureStress();  // explicit A
    void buildJacobiDiagonal();
    int solve();
    void buildValidFaces();
    void recoverVelocityFromPressureStress();
    void applySolutionToVelocity();
    int setup(const ps_params* p, const ps_fields_in* in);
    int run(const ps_params* p, const ps_fields_in* in, bool doSolveStage);
    void registerArrays();

    // operator
    void applyOperator(const double* x, double* y) const;        // reference-shaped (ApplyPressureStressMatrix.h:102-179)
    void constructGuessVectors();                                 // Solver.cpp:512-531
    // Chebyshev-Jacobi polynomial preconditioner (extension PS_PRE_CHEBYSHEV, include/polystokes.h)
    double chebLmax = 0;
    void estimateLambdaMax();
    void chebyshev(const std::vector<double>& r, std::vector<double>& z) const;
    void chebyshev32(const std::vector<double>& r, std::vector<double>& z) const;   // PS_PRE_CHEBYSHEV_F32: inner vectors stored as fp32
    void applyOperatorInner32(const double* x, double* y) const;
    void precondition(const std::vector<double>& r, std::vector<double>& z) const;   // z = M^-1 r of the configured preconditioner
    void applySection1(const double* x, std::vector<double>& A11_1, std::vector<double>& A21_1) const;
    void applySection2(const double* x, std::vector<double>& tp, std::vector<double>& tt) const;
    void applySection3(const double* x, std::vector<double>& A12_1, std::vector<double>& A22_1) const;
    void applyCombine(const double* x, const std::vector<double>& A11_1, const std::vector<double>& A21_1, const std::vector<double>& tp,
                      const std::vector<double>& tt, const std::vector<double>& A12_1, const std::vector<double>& A22_1, double* y) const;
    void applyOperatorFair(const double* x, double* y) const;    // same math, fused passes ("fair CPU")
    int pcg(std::vector<double>& x, const std::vector<double>& rhs, double tol, int maxit, double& rre) const;
    int bicgstab(std::vector<double>& x, const std::vector<double>& rhs, double tol, int maxit, double& rre) const;
    int eigenCG(std::vector<double>& x, const std::vector<double>& rhs, double tol, int maxit, double& tolError) const;

    // helpers
    Dim centerDim() const;
    Dim faceDim(int a) const;
    Dim edgeDim(int e) const;
    float sampleCenterField(const Field<float>& f, float px, float py, float pz) const;
    float localViscosityAtCell(int i, int j, int k) const;
    float localViscosityAtEdge(int edgeAxis, int i, int j, int k) const;
    void computeSDFWeightsSampled(Field<float>& dst, const Dim& d, const float off[3],
                                  const Field<float>& sdf, bool negate) const;
    template <class F> void forEachOrdered(const Dim& d, F f) const;
    // the same traversal cut into `pieces` contiguous ranges: f(piece, i, j, k); pieces run concurrently, each in the serial order
    template <class F> void forEachOrderedPieces(const Dim& d, int pieces, F f) const;
    void regionCellBoxes(std::vector<int32_t>& box) const;   // per region: min / max cell index (6 ints), from reducedIdx[0]
    int64_t stressDOF(int64_t idx, int type) const;  // Solver.h:586-606 (XX,YY,ZZ,YZ,XZ,XY)
    int64_t faceVelocityDOF(int64_t idx, int axis) const; // Solver.h:628-642
    void connectedComponents();
    void fixReducedRegionBoundaries();
    void fixSmallReducedRegions();
    void airBoundaryLayer();
    void solidBoundaryLayer();
    void constructTiles();
};

// Walk a field in the order UT_VoxelArrayIterator does (whole 16^3 voxel tiles in tile-linear order,
// x-fastest inside a tile) — the order that defines DOF numbering in serialAssignFieldIndices
// (Classifier.cpp:1738-1770).  HDK internals are out of tree: restated, switchable (ps_params.indexOrder).
template <class F>
void Oracle::forEachOrdered(const Dim& d, F f) const {
    if (P.indexOrder == PS_ORDER_LINEAR) {
        for (int k = 0; k < d.n[2]; ++k)
            for (int j = 0; j < d.n[1]; ++j)
                for (int i = 0; i < d.n[0]; ++i) f(i, j, k);
        return;
    }
    const int T = 16;
    for (int z0 = 0; z0 < d.n[2]; z0 += T)
        for (int y0 = 0; y0 < d.n[1]; y0 += T)
            for (int x0 = 0; x0 < d.n[0]; x0 += T) {
                const int z1 = (z0 + T < d.n[2] ? z0 + T : d.n[2]), y1 = (y0 + T < d.n[1] ? y0 + T : d.n[1]), x1 = (x0 + T < d.n[0] ? x0 + T : d.n[0]);
                for (int k = z0; k < z1; ++k)
                    for (int j = y0; j < y1; ++j)
                        for (int i = x0; i < x1; ++i) f(i, j, k);
            }
}

template <class F>
void Oracle::forEachOrderedPieces(const Dim& d, int pieces, F f) const {
    // blocks of the traversal in order: k-planes (linear order) or 16^3 voxel tiles (tile-linear)
    struct Blk { int x0, x1, y0, y1, z0, z1; };
    std::vector<Blk> blocks;
    if (P.indexOrder == PS_ORDER_LINEAR) {
        for (int k = 0; k < d.n[2]; ++k) blocks.push_back(Blk{0, d.n[0], 0, d.n[1], k, k + 1});
    } else {
        const int T = 16;
        for (int z0 = 0; z0 < d.n[2]; z0 += T)
            for (int y0 = 0; y0 < d.n[1]; y0 += T)
                for (int x0 = 0; x0 < d.n[0]; x0 += T)
                    blocks.push_back(Blk{x0, (x0 + T < d.n[0] ? x0 + T : d.n[0]), y0, (y0 + T < d.n[1] ? y0 + T : d.n[1]), z0, (z0 + T < d.n[2] ? z0 + T : d.n[2])});
    }
    const int64_t nb = (int64_t)blocks.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(setupThreads > 1 ? setupThreads : 1)
    for (int pc = 0; pc < pieces; ++pc) {
        const int64_t b0 = nb * pc / pieces, b1 = nb * (pc + 1) / pieces;
        for (int64_t b = b0; b < b1; ++b) {
            const Blk& q = blocks[(size_t)b];
            for (int k = q.z0; k < q.z1; ++k)
                for (int j = q.y0; j < q.y1; ++j)
                    for (int i = q.x0; i < q.x1; ++i) f(pc, i, j, k);
        }
    }
}

void buildConversionCoefficients(const double off[3], int axis, double out[RD]);  // Solver.cpp:2105-2149
bool fullPivLuSolve(const double* N, const double* rhs, double* x);                // Eigen FullPivLU semantics
bool partialPivInverse(const double* B, double* Binv);                              // Eigen PartialPivLU inverse semantics

}  // namespace psoracle
