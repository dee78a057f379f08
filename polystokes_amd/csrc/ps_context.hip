// C ABI (include/polystokes.h) and stage orchestration — the solveGasSubclass() sequence of
// exec/HDK_PolyStokes.C:222-609 driving HIP kernels.  No CPU fallback exists: every entry point that
// computes requires a HIP device and fails loudly otherwise.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>

#include "ps_context.hpp"

using namespace ps;

namespace {

struct StageTimer {
    hipEvent_t ev[PS_STAGE_COUNT + 1];
    hipStream_t s;
    int n = 0;
    explicit StageTimer(hipStream_t st) : s(st) {
        for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    }
    ~StageTimer() { for (auto& e : ev) (void)hipEventDestroy(e); }
    void mark(int i) { HIP_CHECK(hipEventRecord(ev[i], s)); }
    double ms(int a, int b) { float f = 0; HIP_CHECK(hipEventElapsedTime(&f, ev[a], ev[b])); return f; }
};

void uploadField(DevBuf<float>& d, const float* src, int64_t n, hipStream_t s) {
    d.alloc((size_t)n);
    if (src) HIP_CHECK(hipMemcpyAsync(d.p, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
    else HIP_CHECK(hipMemsetAsync(d.p, 0, (size_t)n * sizeof(float), s));
}

// several arrays filled by one launch: blockIdx.y = the array, grid-stride over its entries
struct Fill21 { int32_t* p[21]; int64_t n[21]; };
__global__ void k_fill32_multi(Fill21 F, int32_t v) {
    int32_t* __restrict__ a = F.p[blockIdx.y];
    const int64_t n = F.n[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void k_fill32(int32_t* a, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

__global__ void k_gather_perm8(double* __restrict__ dst, const double* __restrict__ src, const int32_t* __restrict__ perm, int64_t off, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[off + i]];
}
__global__ void k_scatter_perm8(double* __restrict__ dst, const double* __restrict__ src, const int32_t* __restrict__ perm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[perm[i]] = src[i];
}

const char* kSampleName[7] = {"center", "faceX", "faceY", "faceZ", "edgeYZ", "edgeXZ", "edgeXY"};

}  // namespace

void ps_context::upload(const ps_params* p, const ps_fields_in* in) {
    if (!p || !in) throw Error("null params/fields");
    if (in->nx <= 0 || in->ny <= 0 || in->nz <= 0) throw Error("bad resolution");
    if (in->nx > 1022 || in->ny > 1022 || in->nz > 1022) throw Error("resolution above 1022 per axis is not supported");
    if (!in->vel[0] || !in->vel[1] || !in->vel[2]) throw Error("Velocity field is missing.");
    if (!in->surface) throw Error("Surface field is missing.");
    if (!in->collision) throw Error("Collision field is missing.");
    if (!in->viscosity) throw Error("Viscosity field is missing.");
    if (p->matrixSetup != PS_PRESSURE_STRESS) throw Error("Unsupported matrix setup.");
    // numeric sanity the node's UI ranges guarantee (HDK_PolyStokes.C:88-208); a raw ABI caller gets an error instead of a
    // division by zero in `cell % tileSize` or a non-finite operator
    if (!(in->dx > 0.f) || !(in->dt > 0.f) || !std::isfinite(in->dx) || !std::isfinite(in->dt)) throw Error("dx and dt must be positive and finite");
    if (!(in->density > 0.f) || !std::isfinite(in->density)) throw Error("density must be positive and finite");
    if (p->doReducedRegions && p->doTile && p->tileSize < 1) throw Error("tileSize must be at least 1");
    if (p->tilePadding < 0 || p->activeLiquidBoundaryLayerSize < 0 || p->activeSolidBoundaryLayerSize < 0) throw Error("layer sizes and tilePadding must not be negative");
    if (!(p->tolerance >= 0.) || p->maxSolverIterations < 0) throw Error("tolerance and maxSolverIterations must not be negative");
    if (p->preconditioner != PS_PRE_IDENTITY && p->preconditioner != PS_PRE_DIAGONAL && p->preconditioner != PS_PRE_CHEBYSHEV && p->preconditioner != PS_PRE_CHEBYSHEV_F32)
        throw Error("Unsupported preconditioner.");
    if (p->preconditionerDegree < 0 || p->preconditionerDegree > 64) throw Error("preconditionerDegree must lie in 0..64");
    P = *p;
    // PS_PRE_CHEBYSHEV_F32 is PS_PRE_CHEBYSHEV with permission to keep the polynomial's inner vectors in fp32 where the kernels for it run
    chebInner32Req = p->preconditioner == PS_PRE_CHEBYSHEV_F32;
    if (chebInner32Req) P.preconditioner = PS_PRE_CHEBYSHEV;
    chebInner32 = false;
    g.nx = in->nx; g.ny = in->ny; g.nz = in->nz; g.order = p->indexOrder;
    dx = in->dx; invDx = 1. / dx; dt = in->dt; invDt = 1. / dt; rho = (double)in->density;
    HIP_CHECK(hipSetDevice(device));
    const int64_t nc = g.count(0);
    uploadField(surface, in->surface, nc, stream);
    uploadField(collision, in->collision, nc, stream);
    uploadField(viscosity, in->viscosity, nc, stream);
    {   // a constant viscosity field (the usual case: a scalar parameter) needs no sampling: trilinear interpolation of a constant returns
        // it bit for bit (a + (b - a) t with a == b), so the setup kernels skip the 8 loads per sample (ps_tiles.hip, ps_blocks.hip)
        const float v0 = in->viscosity[0];
        bool same = true;
        for (int64_t i = 1; i < nc && same; ++i) same = in->viscosity[i] == v0;
        viscUniform = same && std::isfinite(v0);
        viscUniformValue = v0;
    }
    for (int a = 0; a < 3; ++a) {
        uploadField(vel[a], in->vel[a], g.count(1 + a), stream);
        uploadField(cvel[a], in->collisionvel[a], g.count(1 + a), stream);
        velOut[a].alloc((size_t)g.count(1 + a));
        valid[a].alloc((size_t)g.count(1 + a));
        faceRow[a].alloc((size_t)g.count(1 + a));
    }
    haveInputWeights = true;
    for (int w = 0; w < 14; ++w) if (!in->weights[w]) haveInputWeights = false;
    for (int s = 0; s < 7; ++s) {
        const int64_t n = g.count(s);
        liquidW[s].alloc((size_t)n); fluidW[s].alloc((size_t)n);
        labels[s].alloc((size_t)n); activeIdx[s].alloc((size_t)n); reducedIdx[s].alloc((size_t)n);
        if (haveInputWeights) {
            HIP_CHECK(hipMemcpyAsync(liquidW[s].p, in->weights[s], (size_t)n * 4, hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(fluidW[s].p, in->weights[7 + s], (size_t)n * 4, hipMemcpyHostToDevice, stream));
        }
    }
    for (int q = 0; q < 3; ++q) cellScratch[q].alloc((size_t)nc);
    counters.alloc(64);
    HIP_CHECK(hipStreamSynchronize(stream));
    uploaded = true; isSetup = false; isSolved = false;
    arrays.clear();          // the registered device pointers may have been re-allocated above
    slabEnabled = false;     // a decomposition describes ONE grid: set it again after every upload (ps_set_slab / ps_set_brick)
    deviceShareRows = 0; nXseg[0] = nXseg[1] = 0;
    blockMapOwned = -1;
    gOff[0] = gOff[1] = gOff[2] = 0;
}

void ps_context::fillDimData(ps_stats* st) const {   // Solver.cpp:578-593
    double* dd = st->dimData;
    dd[0] = (double)nCenter; dd[1] = (double)nFace[0]; dd[2] = (double)nFace[1]; dd[3] = (double)nFace[2];
    dd[4] = (double)nEdge[0]; dd[5] = (double)nEdge[1]; dd[6] = (double)nEdge[2];
    dd[7] = (double)nActiveVs; dd[8] = (double)nFace[0]; dd[9] = (double)nFace[1]; dd[10] = (double)nFace[2];
    dd[11] = (double)nReducedVs; dd[12] = (double)nPressures; dd[13] = (double)nStresses;
    dd[14] = dd[15] = dd[16] = (double)nCenter;
    dd[17] = (double)nEdge[0]; dd[18] = (double)nEdge[1]; dd[19] = (double)nEdge[2];
    dd[20] = (double)nTotalDOFs; dd[21] = (double)nSystem; dd[22] = 1.; dd[23] = 0.;
    dd[24] = (double)regionCount; dd[25] = dx; dd[26] = dt;
}

// HDK_PolyStokes.C:344-476: everything between setupClockStart() and setupClockEnd().
// Three phases, so that the ranks of a decomposition can exchange the cell labels of their halo blocks at the two points where the
// reference's classification looks further than a halo block reaches (ps_dist.hpp: Dist::exchangeLabels):
//   0: weights, classifyCells, constructReducedRegions                  -> labels before the regions exist
//   1: classifyFaces / Edges, connected components, fixReducedRegionBoundaries   -> labels after the boundary fix
//   2: fixSmallReducedRegions and everything after it
// A single domain runs them back to back.
namespace {
struct SetupState {
    StageTimer T;
    std::clock_t c0;
    std::chrono::high_resolution_clock::time_point w0;
    explicit SetupState(hipStream_t s) : T(s), c0(std::clock()), w0(std::chrono::high_resolution_clock::now()) {}
};
}
int ps_context::setup(ps_stats* stats) {
    for (int ph = 0; ph < 3; ++ph) setupPhase(ph);
    if (stats) *stats = lastStats;
    return PS_SUCCESS;
}
void ps_context::setupPhase(int phase) {
    if (phase == 0) {
        if (!uploaded) throw Error("ps_upload_fields has not been called");
        isSetup = false; isSolved = false;   // a setup that throws must not leave the previous step's system looking valid
        setupPhaseDone = -1;
        arrays.clear();
        HIP_CHECK(hipSetDevice(device));
        setupState = std::make_shared<SetupState>(stream);
    }
    if (!setupState || setupPhaseDone != phase - 1) throw Error("setup phases out of order");
    HIP_CHECK(hipSetDevice(device));
    SetupState& Z = *std::static_pointer_cast<SetupState>(setupState);
    StageTimer& T = Z.T;
    if (phase == 0) {
        bboxValid = false;
        // Solver ctor: labels / indices start UNASSIGNED (Solver.cpp:86-152)
        {   // the 21 label / index arrays in ONE launch (r06: 21 launches of 5 - 20 us each before)
            Fill21 F;
            int64_t most = 1;
            for (int s = 0; s < 7; ++s) {
                const int64_t n = g.count(s);
                F.p[3 * s] = labels[s].p; F.p[3 * s + 1] = activeIdx[s].p; F.p[3 * s + 2] = reducedIdx[s].p;
                F.n[3 * s] = F.n[3 * s + 1] = F.n[3 * s + 2] = n;
                most = std::max(most, n);
            }
            hipLaunchKernelGGL(k_fill32_multi, dim3((unsigned)std::min<int64_t>(512, gridFor(most, 256)), 21), dim3(256), 0, stream, F, (int32_t)PS_UNASSIGNED);
        }
        regionCount = 0;
        T.mark(0);
        buildIntegrationWeightsAlt();
        T.mark(1);
        classifyCells();
        if (P.doReducedRegions) constructReducedRegions(); else constructOnlyActiveRegions();
        setupPhaseDone = 0;
        return;
    }
    if (phase == 1) {
        classifyFaces();
        classifyEdges();
        T.mark(2);
        if (P.doReducedRegions) constructCenterReducedIndices(0);
        setupPhaseDone = 1;
        return;
    }
    if (P.doReducedRegions) {
        constructCenterReducedIndices(1);
        constructFacesReducedIndices();
        constructEdgesReducedIndices();
    }
    T.mark(3);
    constructActiveIndices();
    T.mark(4);
    if (P.doReducedRegions) {
        computeCenterOfMasses();
        computeLeastSquaresFits();
        computeReducedMassMatrices();
        computeReducedViscosityMatricesInteriorOnly();
    } else {
        fbItems = 0;
    }
    T.mark(5);
    constructMatrixBlocks();
    T.mark(6);
    assembleSystemPressureStressFactored();
    constructGuessVectors();   // HDK_PolyStokes.C:462-467 (runs before assemble() there; it only needs the blocks)
    T.mark(7);
    constructPreconditioner();
    T.mark(8);
    HIP_CHECK(hipStreamSynchronize(stream));
    const auto w1 = std::chrono::high_resolution_clock::now();
    std::memset(&lastStats, 0, sizeof(lastStats));
    lastStats.result = PS_INCOMPLETE;
    fillDimData(&lastStats);
    lastStats.solveData[0] = -1; lastStats.solveData[1] = -1; lastStats.solveData[2] = -1; lastStats.solveData[3] = -1;
    lastStats.solveData[4] = 1000.0 * (double)(std::clock() - Z.c0) / CLOCKS_PER_SEC;
    lastStats.solveData[5] = std::chrono::duration<double, std::milli>(w1 - Z.w0).count();
    lastStats.stage_ms[PS_STAGE_WEIGHTS] = T.ms(0, 1);
    lastStats.stage_ms[PS_STAGE_CLASSIFY] = T.ms(1, 2);
    lastStats.stage_ms[PS_STAGE_REGIONS] = T.ms(2, 3);
    lastStats.stage_ms[PS_STAGE_INDICES] = T.ms(3, 4);
    lastStats.stage_ms[PS_STAGE_TILE_MATRICES] = T.ms(4, 5);
    lastStats.stage_ms[PS_STAGE_BLOCKS] = T.ms(5, 6);
    lastStats.stage_ms[PS_STAGE_ASSEMBLE] = T.ms(6, 7);
    lastStats.stage_ms[PS_STAGE_PRECOND] = T.ms(7, 8);
    setupState.reset();
    setupPhaseDone = 2;
    isSetup = true; isSolved = false;
    registerArrays();
}

// HDK_PolyStokes.C:509-583: solve(), buildValidFaces, recoverVelocityFromPressureStress, applySolutionToVelocity
int ps_context::solveStage(ps_stats* stats) {
    if (!isSetup) throw Error("ps_setup_device has not been called");
    HIP_CHECK(hipSetDevice(device));
    StageTimer T(stream);
    int result = PS_INCOMPLETE;
    const std::clock_t c0 = std::clock();
    const auto w0 = std::chrono::high_resolution_clock::now();
    T.mark(0);
    if (P.doSolve) {
        result = solve();
        HIP_CHECK(hipStreamSynchronize(stream));
        const auto w1 = std::chrono::high_resolution_clock::now();
        lastStats.solveData[0] = solveError;
        lastStats.solveData[1] = solveIterations;
        lastStats.solveData[2] = 1000.0 * (double)(std::clock() - c0) / CLOCKS_PER_SEC;
        lastStats.solveData[3] = std::chrono::duration<double, std::milli>(w1 - w0).count();
    }
    T.mark(1);
    buildValidFaces();
    // HDK_PolyStokes.C:566-583: recovery and write-back run when the solve succeeded OR keepNonConvergedResults is set — also with
    // doSolve off (result INCOMPLETE; the solution vector then still holds the zeros of assemble(), AssembleSystem.cpp:469, and the
    // velocities come out as u = McInv rhs_a on active faces, the smoothed fit on reduced ones).  "Unsupported Solver." returns
    // before any of it (:530-535); an interrupted solve (this library's extension) leaves the field alone.
    const bool apply = result != PS_UNSUPPORTED_SOLVER && !(P.doSolve && interrupted) && (result == PS_SUCCESS || P.keepNonConvergedResults);
    if (apply) {
        recoverVelocityFromPressureStress();
        T.mark(2);
        applySolutionToVelocity();
    } else {
        T.mark(2);
        for (int a = 0; a < 3; ++a)
            HIP_CHECK(hipMemcpyAsync(velOut[a].p, vel[a].p, (size_t)g.count(1 + a) * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    T.mark(3);
    HIP_CHECK(hipStreamSynchronize(stream));
    lastStats.stage_ms[PS_STAGE_SOLVE] = T.ms(0, 1);
    lastStats.stage_ms[PS_STAGE_RECOVER] = T.ms(1, 2);
    lastStats.stage_ms[PS_STAGE_WRITEBACK] = T.ms(2, 3);
    lastStats.result = result;
    lastStats.usedBiCGStab = usedBiCGStab;
    isSolved = true;
    registerArrays();
    if (stats) *stats = lastStats;
    return result;
}

void ps_context::registerArrays() {
    arrays.clear();
    auto reg = [&](const std::string& n, const void* p, int64_t c, int e) { arrays[n] = ArrayInfo{p, c, e}; };
    for (int s = 0; s < 7; ++s) {
        const int64_t n = g.count(s);
        reg(std::string(kSampleName[s]) + "LiquidWeights", liquidW[s].p, n, 4);
        reg(std::string(kSampleName[s]) + "FluidWeights", fluidW[s].p, n, 4);
        reg(std::string(kSampleName[s]) + "Labels", labels[s].p, n, 4);
        reg(std::string(kSampleName[s]) + "ActiveIndices", activeIdx[s].p, n, 4);
        reg(std::string(kSampleName[s]) + "ReducedIndices", reducedIdx[s].p, n, 4);
    }
    const int64_t R = regionCount;
    reg("reducedRegionCOM", COM.p, R * 3, 8);
    reg("reducedRegionCenterOfMass", COM.p, R * 3, 8);   // the reference's output-geometry name (HDK_PolyStokes.h:62-102)
    reg("reducedRegionBestFitVectors", cfit.p, R * PS_RD, 8);
    reg("reducedMassMatrices", Mr.p, R * PS_RD * PS_RD, 8);
    reg("reducedViscosityMatrices", Kv.p, R * PS_RD * PS_RD, 8);
    reg("Inv_Mr_plus_2JDtuDJ", Binv.p, R * PS_RD * PS_RD, 8);
    reg("reducedRHSVector", rhsR.p, R * PS_RD, 8);
    auto regp = [&](const std::string& n, const void* p, int64_t c, const int32_t* perm, int64_t off) {
        ArrayInfo a{p, c, 8};
        a.perm = perm; a.permOffset = off;
        arrays[n] = a;
    };
    // vectors are stored in the internal (block-interleaved) numbering; these views are in reference order
    regp("McInv", McInv.p, nActiveVs, permRow.p, 0);
    regp("activeRHSVector", rhsA.p, nActiveVs, permRow.p, 0);
    regp("oldActiveVs", oldVs.p, nActiveVs, permRow.p, 0);
    regp("uInv", uInv.p, nStresses, permSys.p, nPressures);
    if (P.exportComponentMatrices) {
        regp("Mc", Mc.p, nActiveVs, permRow.p, 0);
        regp("u", uDiag.p, nStresses, permSys.p, nPressures);
    }
    regp("pressureRHSVector", rhsPT.p, nPressures, permSys.p, 0);
    regp("stressRHSVector", rhsPT.p, nStresses, permSys.p, nPressures);
    regp("b", b.p, nSystem, permSys.p, 0);
    regp("solutionVector", x.p, nSystem, permSys.p, 0);
    regp("guessVector", guess.p, nSystem, permSys.p, 0);
    if (P.preconditioner == PS_PRE_DIAGONAL) regp("dinv", dinv.p, nSystem, permSys.p, 0);
    if (isSolved) {
        regp("recoveredActiveVelocity", recovered.p, nActiveVs, permRow.p, 0);
        reg("recoveredReducedVelocity", recovered.p ? recovered.p + nActiveVs : nullptr, nReducedVs, 8);
    }
    reg("valuesCoded", counters.p + 21, 1, 4);
    diagFlagsHost = (uCoded ? 1 : 0) | (mcCoded ? 2 : 0);
    HIP_CHECK(hipMemcpyAsync(counters.p + 27, &diagFlagsHost, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    reg("diagonalsCoded", counters.p + 27, 1, 4);
    reg("columns16", counters.p + 24, 1, 4);
    // 1: the last PCG solve ran the four-kernel step (residual update inside the St kernel, ps_solve.hip)
    HIP_CHECK(hipMemcpyAsync(counters.p + 28, &fusedStepHost, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    reg("fusedStep", counters.p + 28, 1, 4);
    // 1: the last solve / preconditioner apply kept the Chebyshev polynomial's inner vectors in fp32 (PS_PRE_CHEBYSHEV_F32 where its kernels run)
    chebInner32Host = chebInner32 ? 1 : 0;
    HIP_CHECK(hipMemcpyAsync(counters.p + 36, &chebInner32Host, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    reg("chebInner32", counters.p + 36, 1, 4);
    // entries of the distinct runs of the compressed streams / all entries: S, then St (equal without sharing; 0 without a stream)
    streamRunsHost[0] = S.col16ok ? (int32_t)S.uniqueLen : 0; streamRunsHost[1] = S.col16ok ? (int32_t)S.streamLen : 0;
    streamRunsHost[2] = St.col16ok ? (int32_t)St.uniqueLen : 0; streamRunsHost[3] = St.col16ok ? (int32_t)St.streamLen : 0;
    HIP_CHECK(hipMemcpyAsync(counters.p + 29, streamRunsHost, 4 * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    reg("streamRuns", counters.p + 29, 4, 4);
    // bit 0 / 1: S / St run the row-per-lane kernels (DevCSR::ecol); [1]: the numbering mode (ilPlaneMajor)
    rowPerLaneHost[0] = ((S.ellok && S.packed) ? 1 : 0) | ((St.ellok && St.packed) ? 2 : 0); rowPerLaneHost[1] = ilPlaneMajor;
    HIP_CHECK(hipMemcpyAsync(counters.p + 34, rowPerLaneHost, 2 * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    reg("rowPerLane", counters.p + 34, 2, 4);
    reg("sysPerm", permSys.p, nSystem, 4);
    reg("rowPerm", permRow.p, nActiveVs, 4);
    reg("S.ptr", S.ptr.p, S.rows + 1, 4); reg("S.col", S.col.p, S.nnz, 4); reg("S.val", S.val.p, S.nnz, 8);
    reg("St.ptr", St.ptr.p, St.rows + 1, 4); reg("St.col", St.col.p, St.nnz, 4); reg("St.val", St.val.p, St.nnz, 8);
    // the chunk tables of the compressed streams (4 int32 per chunk: run begin, entries | rows << 16, first row, run owner's first row),
    // the coded values and the chunk each chunk shares its run with: for the layout studies of scripts/
    if (S.col16ok) { reg("S.chunkInfo", S.chunkInfo.p, (int64_t)S.nChunks * 4, 4); reg("S.chunkRep", S.chunkRep.p, S.nChunks, 4); }
    if (St.col16ok) { reg("St.chunkInfo", St.chunkInfo.p, (int64_t)St.nChunks * 4, 4); reg("St.chunkRep", St.chunkRep.p, St.nChunks, 4); }
    if (S.packed) reg("S.code", S.code.p, S.nnz, 1);
    if (St.packed) reg("St.code", St.code.p, St.nnz, 1);
    reg("reducedRowFace", rrowFace.p, nReducedRows, 4);
    reg("reducedRowRegion", rrowRegion.p, nReducedRows, 4);
    static const char* ax[3] = {"X", "Y", "Z"};
    for (int a = 0; a < 3; ++a) {
        reg(std::string("vel") + ax[a], velOut[a].p, g.count(1 + a), 4);
        reg(std::string("valid") + ax[a], valid[a].p, g.count(1 + a), 4);
        reg(std::string("faceRow") + ax[a], faceRow[a].p, g.count(1 + a), 4);
        if (slabEnabled && isSolved) reg(std::string("owned") + ax[a], ownedFace[a].p, g.count(1 + a), 4);
    }
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
// Every entry point makes the context's deferred-free list current for the calling thread (ps_common.hpp: SinkScope) and, on an error
// path, releases what the failed call dropped if the stream is idle (the successful paths release at their end: drainDeferred(true)).
#define PS_TRY(ctx, ...)                                                     \
    ps::SinkScope sinkScope_((ctx) ? &(ctx)->deferred : nullptr);            \
    try { __VA_ARGS__ } catch (const ps::Error& e) {                         \
        if (ctx) { (ctx)->err = e.msg; (ctx)->drainDeferred(false); }        \
        return PS_FAILED;                                                    \
    } catch (const std::exception& e) {                                      \
        if (ctx) { (ctx)->err = e.what(); (ctx)->drainDeferred(false); }     \
        return PS_FAILED;                                                    \
    }

static std::string g_createError;

// ---- MatrixMarket export (Solver.cpp:533-606; MarketIO.h:310-380) --------------------------------
static bool writeMarketVector(const std::string& fn, const std::vector<double>& v) {
    std::ofstream out(fn.c_str(), std::ios::out);
    if (!out) return false;
    out.flags(std::ios_base::scientific);
    out.precision(17);   // digits10 + 2
    out << "%%MatrixMarket matrix array real general\n";
    out << v.size() << " " << 1 << "\n";
    for (double d : v) out << d << "\n";
    return true;
}
static bool writeMarketSparse(const std::string& fn, int64_t rows, int64_t cols, const std::vector<int64_t>& ptr,
                              const std::vector<int32_t>& col, const std::vector<double>& val) {
    std::ofstream out(fn.c_str(), std::ios::out);
    if (!out) return false;
    out.flags(std::ios_base::scientific);
    out.precision(17);
    out << "%%MatrixMarket matrix coordinate  real general" << std::endl;
    out << rows << " " << cols << " " << val.size() << "\n";
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t p = ptr[(size_t)r]; p < ptr[(size_t)r + 1]; ++p) out << r + 1 << " " << col[(size_t)p] + 1 << " " << val[(size_t)p] << "\n";
    return true;
}
template <class T>
static std::vector<T> fetch(ps_context* c, const T* dptr, int64_t n) {
    std::vector<T> h((size_t)std::max<int64_t>(n, 0));
    if (n > 0) {
        HIP_CHECK(hipMemcpyAsync(h.data(), dptr, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return h;
}

// assembleSystemPressureStress (AssembleSystem.cpp:351-430) from the device-side blocks: export tooling for solverType EIGEN
// (the solve itself applies the same operator in factored form).  Triplets -> sorted, duplicates summed (setFromTriplets).
void ps_context::buildExplicitA(std::vector<int64_t>& aptr, std::vector<int32_t>& acol, std::vector<double>& aval) {
    const int64_t n = nSystem, nA = nActiveVs, nP = nPressures;
    auto sp = fetch(this, S.ptr.p, nRows + 1);
    auto sc = fetch(this, S.col.p, S.nnz);
    ensureValues(S);
    auto sv = fetch(this, S.val.p, S.nnz);
    auto mc = fetch(this, McInv.p, nA);
    auto ui = fetch(this, uInv.p, n);
    auto perm = fetch(this, permSys.p, n);              // reference -> internal
    std::vector<int32_t> inv((size_t)n);
    for (int64_t i = 0; i < n; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
    struct T { int32_t r, c; double v; };
    std::vector<T> t;
    double est = 0;
    for (int64_t f = 0; f < nA; ++f) { const double l = (double)(sp[(size_t)f + 1] - sp[(size_t)f]); est += l * l; }
    std::vector<std::vector<int32_t>> regCols((size_t)regionCount);
    std::vector<int32_t> rreg;
    std::vector<uint32_t> rface;
    if (regionCount > 0) {
        rreg = fetch(this, rrowRegion.p, nReducedRows);
        rface = fetch(this, rrowFace.p, nReducedRows);
        for (int64_t q = 0; q < nReducedRows; ++q)
            for (int32_t p = sp[(size_t)(nA + q)]; p < sp[(size_t)(nA + q) + 1]; ++p) regCols[(size_t)rreg[(size_t)q]].push_back(sc[(size_t)p]);
        for (auto& v : regCols) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); est += (double)v.size() * (double)v.size(); }
    }
    if (est > 4.0e8) throw Error("the explicit matrix A is too large to export (its reduced part is dense per tile): export the component matrices instead");
    t.reserve((size_t)est + (size_t)n);
    for (int64_t f = 0; f < nA; ++f) {                   // -dt [G Dt]^T McInv [G Dt]
        const double d = -dt * mc[(size_t)f];
        for (int32_t a = sp[(size_t)f]; a < sp[(size_t)f + 1]; ++a)
            for (int32_t b2 = sp[(size_t)f]; b2 < sp[(size_t)f + 1]; ++b2)
                t.push_back({inv[(size_t)sc[(size_t)a]], inv[(size_t)sc[(size_t)b2]], d * sv[(size_t)a] * sv[(size_t)b2]});
    }
    if (regionCount > 0) {                               // -[JG JDt]^T BInv [JG JDt], JS_r = sum_f C_f (x) S_f
        auto com = fetch(this, COM.p, regionCount * 3);
        auto bi = fetch(this, Binv.p, regionCount * PS_RD * PS_RD);
        std::vector<int64_t> first((size_t)regionCount + 1, 0);
        for (int64_t q = 0; q < nReducedRows; ++q) first[(size_t)rreg[(size_t)q] + 1]++;
        for (int64_t r = 0; r < regionCount; ++r) first[(size_t)r + 1] += first[(size_t)r];   // rows are region-contiguous
        for (int64_t r = 0; r < regionCount; ++r) {
            const std::vector<int32_t>& cols = regCols[(size_t)r];
            const size_t m = cols.size();
            if (m == 0) continue;
            std::vector<double> JS((size_t)PS_RD * m, 0.), W((size_t)PS_RD * m, 0.);
            for (int64_t q = first[(size_t)r]; q < first[(size_t)r + 1]; ++q) {
                int i, j, k, axis;
                unpackFace(rface[(size_t)q], i, j, k, axis);
                double pos[3] = {(double)(i + gOff[0]), (double)(j + gOff[1]), (double)(k + gOff[2])};
                pos[axis] -= 0.5;
                double cf[PS_RD];
                basisRow(pos[0] * dx - com[(size_t)r * 3], pos[1] * dx - com[(size_t)r * 3 + 1], pos[2] * dx - com[(size_t)r * 3 + 2], axis, cf);
                for (int32_t p = sp[(size_t)(nA + q)]; p < sp[(size_t)(nA + q) + 1]; ++p) {
                    const size_t lc = (size_t)(std::lower_bound(cols.begin(), cols.end(), sc[(size_t)p]) - cols.begin());
                    for (int e = 0; e < PS_RD; ++e) JS[(size_t)e * m + lc] += cf[e] * sv[(size_t)p];
                }
            }
            for (int e = 0; e < PS_RD; ++e)
                for (int g2 = 0; g2 < PS_RD; ++g2) {
                    const double bv = bi[(size_t)r * PS_RD * PS_RD + (size_t)e * PS_RD + g2];
                    if (bv == 0.) continue;
                    for (size_t c2 = 0; c2 < m; ++c2) W[(size_t)e * m + c2] += bv * JS[(size_t)g2 * m + c2];
                }
            for (size_t a = 0; a < m; ++a)
                for (size_t b2 = 0; b2 < m; ++b2) {
                    double s2 = 0;
                    for (int e = 0; e < PS_RD; ++e) s2 += JS[(size_t)e * m + a] * W[(size_t)e * m + b2];
                    t.push_back({inv[(size_t)cols[a]], inv[(size_t)cols[b2]], -s2});
                }
        }
    }
    for (int64_t i = nP; i < n; ++i) t.push_back({(int32_t)i, (int32_t)i, -0.5 * ui[(size_t)perm[(size_t)i]]});   // -1/2 uInv on the stress block
    std::stable_sort(t.begin(), t.end(), [](const T& a, const T& b2) { return a.r != b2.r ? a.r < b2.r : a.c < b2.c; });
    aptr.assign((size_t)n + 1, 0); acol.clear(); aval.clear();
    size_t p = 0;
    for (int64_t r = 0; r < n; ++r) {
        aptr[(size_t)r] = (int64_t)aval.size();
        while (p < t.size() && t[p].r == r) {
            const int32_t cc = t[p].c;
            double s2 = 0;
            while (p < t.size() && t[p].r == r && t[p].c == cc) { s2 += t[p].v; ++p; }
            acol.push_back(cc); aval.push_back(s2);
        }
    }
    aptr[(size_t)n] = (int64_t)aval.size();
}

extern "C" {

int32_t ps_abi_version(void) { return 1; }
int32_t ps_reduced_dof(void) { return PS_RD; }

ps_context* ps_context_create(int32_t device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_createError = "polystokes: no HIP device available (the product path has no CPU fallback)";
        std::fprintf(stderr, "%s\n", g_createError.c_str());
        return nullptr;
    }
    if (device < 0 || device >= count) {
        g_createError = "polystokes: device index out of range";
        std::fprintf(stderr, "%s\n", g_createError.c_str());
        return nullptr;
    }
    ps_context* c = new ps_context();
    c->device = device;
    // a NON-BLOCKING stream: the host application's work on the legacy default stream neither waits for ours nor makes ours wait (every input
    // and output of the ABI is a host pointer; the entry points synchronise this stream themselves before they return results)
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        g_createError = "polystokes: cannot initialise HIP device";
        std::fprintf(stderr, "%s\n", g_createError.c_str());
        return nullptr;
    }
    ps_params_default(&c->P);
    return c;
}

void ps_context_destroy(ps_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    ps_dist_release(c);   // communicator / sockets first, then the stream they use
    hipStream_t s = c->ownsStream ? c->stream : nullptr;
    if (c->pinnedCounters) (void)hipHostFree(c->pinnedCounters);
    delete c;                              // (releases the context's deferred list, then its buffers)
    ps::releaseDeferred(ps::orphanFrees());
    if (s) (void)hipStreamDestroy(s);
}

const char* ps_last_error(const ps_context* c) { return c ? c->err.c_str() : g_createError.c_str(); }

void ps_params_default(ps_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->mindensity = 1; p->maxdensity = 100000;
    p->matrixSetup = PS_PRESSURE_STRESS; p->solverType = PS_PCG_MATRIX_VECTOR_PRODUCTS;
    p->doSolve = 1; p->keepNonConvergedResults = 1; p->useWarmStart = 1;
    p->tolerance = 1e-3; p->maxSolverIterations = 5000;
    p->useInputSurfaceWeights = 1; p->useInputCollisionWeights = 1;
    p->activeLiquidBoundaryLayerSize = 2; p->activeSolidBoundaryLayerSize = 2;
    p->doReducedRegions = 1; p->doTile = 1; p->tileSize = 16; p->tilePadding = 2;
    p->preconditioner = PS_PRE_IDENTITY; p->indexOrder = PS_ORDER_VOXEL_TILES; p->negateCollision = 1;
}

int32_t ps_upload_fields(ps_context* c, const ps_params* p, const ps_fields_in* in) {
    if (!c) return PS_FAILED;
    PS_TRY(c, { c->upload(p, in); return PS_SUCCESS; })
}
// A context with a slab holds one rank's part of a distributed system (b lacks the neighbours' contributions, the diagonal
// is not completed, only owned faces have rows): the split entry points would solve that fragment on its own.
static void refuseSlab(const ps_context* c, const char* fn) {
    if (c->slabEnabled) throw Error(std::string(fn) + ": a slab is set — use ps_step_device (one process per GPU) or ps_group_step");
}
int32_t ps_setup_device(ps_context* c, ps_stats* st) {
    if (!c) return PS_FAILED;
    PS_TRY(c, { refuseSlab(c, "ps_setup_device"); const int rc = c->setup(st); c->drainDeferred(true); return rc; })   // (setup ends with the stream synchronised)
}
int32_t ps_solve_device(ps_context* c, ps_stats* st) {
    if (!c) return PS_FAILED;
    PS_TRY(c, { refuseSlab(c, "ps_solve_device"); const int result = c->solveStage(st); c->drainDeferred(true); return result; })
}
int32_t ps_step_device(ps_context* c, ps_stats* st) {
    if (!c) return PS_FAILED;
    PS_TRY(c, {
        if (c->slabEnabled) {
            if (!c->rcclComm && !c->hostComm) throw Error("a slab is set but no communicator: call ps_comm_init_rccl / ps_comm_init_tcp (or use ps_group_step)");
            const int result = ps_dist_step_single(c, st);
            c->drainDeferred(true);
            return result;
        }
        const int rc = c->setup(nullptr);
        if (rc != PS_SUCCESS) { c->drainDeferred(true); return rc; }
        const int result = c->solveStage(st);
        c->drainDeferred(true);             // (the solve stage ends with the stream synchronised)
        return result;
    })
}
int32_t ps_memory_stats(const ps_context* c, int64_t* out4) {
    if (!out4) return PS_FAILED;
    ps::MemState& M = ps::memState();
    std::lock_guard<std::mutex> lk(M.m);
    out4[0] = M.liveBytes; out4[1] = M.peakBytes; out4[2] = 0;
    if (c) { ps::DeferredFrees& d = const_cast<ps_context*>(c)->deferred; std::lock_guard<std::mutex> l2(d.m); out4[2] = (int64_t)d.bytes; }
    out4[3] = M.contexts;
    return PS_SUCCESS;
}
int32_t ps_set_interrupt(ps_context* c, ps_interrupt_fn cb, void* user) {
    if (!c) return PS_FAILED;
    c->interruptCb = cb;
    c->interruptUser = user;
    return PS_SUCCESS;
}
int32_t ps_download_fields(ps_context* c, ps_fields_out* out) {
    if (!c || !out) return PS_FAILED;
    PS_TRY(c, {
        HIP_CHECK(hipSetDevice(c->device));
        for (int a = 0; a < 3; ++a) {
            const size_t nb = (size_t)c->g.count(1 + a) * sizeof(float);
            if (out->vel[a]) HIP_CHECK(hipMemcpyAsync(out->vel[a], c->velOut[a].p, nb, hipMemcpyDeviceToHost, c->stream));
            if (out->valid[a]) HIP_CHECK(hipMemcpyAsync(out->valid[a], c->valid[a].p, nb, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return PS_SUCCESS;
    })
}
int32_t polystokes_step(ps_context* c, const ps_params* p, const ps_fields_in* in, ps_fields_out* out, ps_stats* st) {
    if (!c) return PS_FAILED;
    PS_TRY(c, {
        c->upload(p, in);
        const int rc = c->setup(nullptr);
        if (rc != PS_SUCCESS) { c->drainDeferred(true); return rc; }
        const int result = c->solveStage(st);
        c->drainDeferred(true);
        if (out) {
            const int rc2 = ps_download_fields(c, out);
            if (rc2 != PS_SUCCESS) return rc2;
        }
        if (p->exportMatrices && p->exportDataPrefix) ps_export_matrices(c, p->exportDataPrefix);
        if (p->exportComponentMatrices && p->exportDataPrefix) ps_export_component_matrices(c, p->exportDataPrefix);
        if (p->exportStats && p->exportDataPrefix) ps_export_stats(c, st, p->exportDataPrefix);
        return result;
    })
}

int32_t ps_apply_operator(ps_context* c, const double* x, double* y) {
    if (!c) return PS_FAILED;
    PS_TRY(c, {
        if (!c->isSetup) throw Error("not set up");
        HIP_CHECK(hipSetDevice(c->device));
        const size_t n = (size_t)c->nSystem;
        c->tmp1.alloc(n); c->tmp2.alloc(n);
        c->tmp3.alloc(n);
        HIP_CHECK(hipMemcpyAsync(c->tmp3.p, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_scatter_perm8, dim3(gridFor((int64_t)n, 256)), dim3(256), 0, c->stream, c->tmp1.p, c->tmp3.p, c->permSys.p, (int64_t)n);
        c->applyOperator(c->tmp1.p, c->tmp2.p, c->dotPartials.p);
        hipLaunchKernelGGL(k_gather_perm8, dim3(gridFor((int64_t)n, 256)), dim3(256), 0, c->stream, c->tmp3.p, c->tmp2.p, c->permSys.p, (int64_t)0, (int64_t)n);
        HIP_CHECK(hipMemcpyAsync(y, c->tmp3.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return PS_SUCCESS;
    })
}

int32_t ps_apply_preconditioner(ps_context* c, const double* rin, double* zout) {
    if (!c || !rin || !zout) return PS_FAILED;
    PS_TRY(c, {
        if (!c->isSetup) throw Error("not set up");
        if (c->slabEnabled) throw Error("ps_apply_preconditioner is a single-domain call");
        HIP_CHECK(hipSetDevice(c->device));
        const size_t n = (size_t)c->nSystem;
        c->tmp3.alloc(n); c->tmp4.alloc(n); c->tmp1.alloc(n); c->tmp2.alloc(n);
        HIP_CHECK(hipMemcpyAsync(c->tmp3.p, rin, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_scatter_perm8, dim3(gridFor((int64_t)n, 256)), dim3(256), 0, c->stream, c->tmp4.p, c->tmp3.p, c->permSys.p, (int64_t)n);   // tmp4 = r (internal order)
        c->applyPreconditionerDevice(c->tmp4.p, c->tmp1.p, c->tmp2.p);                                                                                  // tmp1 = z
        hipLaunchKernelGGL(k_gather_perm8, dim3(gridFor((int64_t)n, 256)), dim3(256), 0, c->stream, c->tmp3.p, c->tmp1.p, c->permSys.p, (int64_t)0, (int64_t)n);
        HIP_CHECK(hipMemcpyAsync(zout, c->tmp3.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return PS_SUCCESS;
    })
}

int64_t ps_query_array(ps_context* c, const char* name, int32_t* elem_bytes) {
    if (!c || !name) return -1;
    auto it = c->arrays.find(name);
    if (it == c->arrays.end()) return -1;
    if (elem_bytes) *elem_bytes = it->second.elem;
    return it->second.count;
}
int32_t ps_read_array(ps_context* c, const char* name, void* dst, int64_t dst_bytes) {
    if (!c || !name) return PS_FAILED;
    PS_TRY(c, {
        if (c->isSetup && (std::strcmp(name, "S.val") == 0 || std::strcmp(name, "St.val") == 0)) {   // decoded on demand (ps_context::ensureValues)
            ps::DevCSR& M = name[1] == '.' ? c->S : c->St;
            c->ensureValues(M);
            c->arrays[name].dptr = M.val.p;
        }
        auto it = c->arrays.find(name);
        if (it == c->arrays.end()) throw Error(std::string("unknown array ") + name);
        const int64_t need = it->second.count * it->second.elem;
        if (dst_bytes < need) throw Error("destination too small");
        if (need > 0) {
            HIP_CHECK(hipSetDevice(c->device));
            const void* src = it->second.dptr;
            if (it->second.perm) {
                const int64_t n = it->second.count;
                c->tmp3.alloc((size_t)n);
                hipLaunchKernelGGL(k_gather_perm8, dim3(gridFor(n, 256)), dim3(256), 0, c->stream, c->tmp3.p, (const double*)it->second.dptr,
                                   it->second.perm, it->second.permOffset, n);
                src = c->tmp3.p;
            }
            HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)need, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        return PS_SUCCESS;
    })
}

// kernel micro-benchmarks for bench.py's roofline object (HIP events on the solver stream)
int32_t ps_bench_kernel(ps_context* c, const char* kernel, int32_t iters, double* avg_ms, double* algorithmic_bytes) {
    if (!c || !kernel) return PS_FAILED;
    PS_TRY(c, {
        if (!c->isSetup) throw Error("not set up");
        HIP_CHECK(hipSetDevice(c->device));
        std::string k(kernel);
        // "seq:<name>": the kernel timed in its place in the CG iteration (S, tiles, St, update_r, update_xp): its predecessor
        // of the loop runs, untimed, before every timed launch.  Back-to-back replays of ONE kernel see another cache / DRAM
        // page state than the solve does (St: 0.42 ms replayed, 0.39 ms in sequence at 256^3); rocprof's per-kernel average
        // over a real solve matches the in-sequence figure.
        const bool seq = k.compare(0, 4, "seq:") == 0;
        if (seq) k = k.substr(4);
        // "_fp64": the streams as a system with arbitrary weights has them — no shared runs (equal codes there, not equal values)
        const bool unshared = k.size() > 5 && k.compare(k.size() - 5, 5, "_fp64") == 0 && c->S.col16ok && c->St.col16ok && c->S.packed;
        struct Reshare { ps_context* c; bool on; ~Reshare() { if (on) { try { c->buildStreams(true); } catch (...) {} } } } reshare{c, unshared};
        if (unshared) c->buildStreams(false);
        std::vector<std::string> pred;
        if (seq) {
            if (k == "spmv_St" || k == "spmv_St_r") pred = {"spmv_S", "tiles"};
            else if (k == "cg_update_xp_u") pred = {"spmv_St_r"};
            else if (k == "spmv_S") pred = {"cg_update_xp"};
            else if (k == "tiles") pred = {"spmv_S"};
            else if (k == "cg_update_r") pred = {"spmv_St"};
            else if (k == "cg_update_xp") pred = {"cg_update_r"};
            else throw Error("seq: unknown kernel " + k);
        }
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        const size_t n = (size_t)c->nSystem;
        c->tmp1.alloc(n); c->tmp2.alloc(n);
        HIP_CHECK(hipMemcpyAsync(c->tmp1.p, c->b.p, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        for (int w = 0; w < 3; ++w) { for (const std::string& q : pred) ps_bench_launch(c, q, c->tmp1.p, c->tmp2.p); ps_bench_launch(c, k, c->tmp1.p, c->tmp2.p); }
        float ms = 0;
        if (!seq) {
            HIP_CHECK(hipEventRecord(e0, c->stream));
            for (int i = 0; i < iters; ++i) ps_bench_launch(c, k, c->tmp1.p, c->tmp2.p);
            HIP_CHECK(hipEventRecord(e1, c->stream));
            HIP_CHECK(hipEventSynchronize(e1));
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        } else {
            for (int i = 0; i < iters; ++i) {
                for (const std::string& q : pred) ps_bench_launch(c, q, c->tmp1.p, c->tmp2.p);
                HIP_CHECK(hipEventRecord(e0, c->stream));
                ps_bench_launch(c, k, c->tmp1.p, c->tmp2.p);
                HIP_CHECK(hipEventRecord(e1, c->stream));
                HIP_CHECK(hipEventSynchronize(e1));
                float one = 0;
                HIP_CHECK(hipEventElapsedTime(&one, e0, e1));
                ms += one;
            }
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (avg_ms) *avg_ms = (double)ms / (double)iters;
        if (algorithmic_bytes) {
            // CSR with fp64 values, int32 columns, int32 row pointers:
            // 12 nnz + 4 (rows+1) + 8 rows (y) + 8 cols (x read once) + fused diagonal / x reads of the epilogue
            const double nnz = (double)c->S.nnz, rowsS = (double)c->nRows, rowsT = (double)c->nSystem;
            auto endsWith = [&](const char* suf) { const size_t m = std::strlen(suf); return k.size() > m && k.compare(k.size() - m, m, suf) == 0; };
            const bool fp64 = endsWith("_fp64"), csr = endsWith("_csr");
            const std::string kb = fp64 ? k.substr(0, k.size() - 5) : (csr ? k.substr(0, k.size() - 4) : k);
            // bytes per stored entry of the form the named kernel streams: 16-bit windowed column + int8 code (3) | the same
            // column + fp64 value (10) | int32 column + int8 code (5) | int32 column + fp64 value (12)
            auto perNnz = [&](const ps::DevCSR& M) { return csr ? 12. : (M.col16ok ? ((M.packed && !fp64) ? 3. : 10.) : (M.packed ? 5. : 12.)); };
            auto c16 = [&](const ps::DevCSR& M) { return !csr && M.col16ok; };
            const double perNnzS = perNnz(c->S), perNnzT = perNnz(c->St);
            // every entry of the matrix counts once per launch: the kernel loads all of them; how many of those loads are served from
            // cache because chunks share their runs (DevCSR::uniqueLen, array "streamRuns") shows in the measured HBM traffic
            const double entS = nnz, entT = nnz;
            const double winS = c16(c->S) ? 80. * (double)c->S.nChunks : 0., winT = c16(c->St) ? 80. * (double)c->St.nChunks : 0.;   // 64 B of window bases + 16 B of ranges per chunk
            // row length byte | row pointer; the row-per-lane kernels read neither (rows padded to the unit's width: the padded slots are
            // NOT counted as bytes moved — 3 B per real entry, as above)
            auto ell = [&](const ps::DevCSR& M) { return c16(M) && M.packed && M.ellok && !fp64; };
            const double ptrS = c16(c->S) ? (ell(c->S) ? 0. : 1.) : 4., ptrT = c16(c->St) ? (ell(c->St) ? 0. : 1.) : 4.;
            // the fused diagonals: fp64 array, or a 1-byte value-set code per row on the pipelined kernels (ps_context.hpp: uCode / mcCode)
            const double dMc = (c16(c->S) && c->mcCoded) ? 1. : 8., dU = (c16(c->St) && c->uCoded) ? 1. : 8.;
            const double bS = winS + perNnzS * entS + ptrS * (rowsS + 1) + 8. * rowsS + 8. * rowsT + dMc * (double)c->nActiveVs;
            const double bT = winT + perNnzT * entT + ptrT * (rowsT + 1) + 8. * rowsT + 8. * rowsS + 8. * rowsT + dU * rowsT;
            const double dg = c->P.preconditioner == PS_PRE_DIAGONAL ? (double)sizeof(ps::diag_t) : 0.;   // the stored Jacobi diagonal (ps_common.hpp)
            if (kb == "spmv_S") *algorithmic_bytes = bS;
            else if (kb == "spmv_St") *algorithmic_bytes = bT;
            // fused residual update: r read and written in place of the A p store, + the stored Jacobi diagonal
            else if (kb == "spmv_St_r") *algorithmic_bytes = bT + (8. + dg) * rowsT;
            else if (kb == "cg_update_xp_u") *algorithmic_bytes = (40. + dg + (c->uCoded ? 1. : 8.)) * rowsT;
            else if (kb == "apply") *algorithmic_bytes = bS + bT + (double)c->nReducedRows * (8. + 4. + 8. + 8. + 4.);
            else if (kb == "tiles") *algorithmic_bytes = (double)c->nReducedRows * (8. + 4. + 8. + 4.);   // s in, t out, packed face x2
            else if (kb == "cg_update_r") *algorithmic_bytes = (24. + dg) * rowsT;
            else if (kb == "cg_update_xp") *algorithmic_bytes = (40. + dg) * rowsT;
            else if (kb == "cg_update_xr") *algorithmic_bytes = (c->P.preconditioner == PS_PRE_DIAGONAL ? 56. : 48.) * rowsT;
            else if (kb == "cg_update_p") *algorithmic_bytes = (c->P.preconditioner == PS_PRE_DIAGONAL ? 32. : 24.) * rowsT;
            else *algorithmic_bytes = 0;
        }
        return PS_SUCCESS;
    })
}

int32_t ps_export_stats(ps_context* c, const ps_stats* st, const char* prefix) {
    if (!c || !st || !prefix) return PS_FAILED;
    PS_TRY(c, {
        std::vector<double> dd(st->dimData, st->dimData + 27), sd(st->solveData, st->solveData + 6);
        if (!writeMarketVector(std::string(prefix) + "dimData.mtx", dd)) throw Error("cannot write dimData.mtx");
        if (!writeMarketVector(std::string(prefix) + "solveData.mtx", sd)) throw Error("cannot write solveData.mtx");
        return PS_SUCCESS;
    })
}

// exportComponentMatrices (Solver.cpp:543-566): G, Dt, JG, JDt are materialised on the host from S and the
// per-row basis (the device never stores JG/JDt), diagonals as sparse diagonal matrices.
int32_t ps_export_component_matrices(ps_context* c, const char* prefix) {
    if (!c || !prefix) return PS_FAILED;
    PS_TRY(c, {
        if (!c->isSetup) throw Error("not set up");
        HIP_CHECK(hipSetDevice(c->device));
        const std::string pre(prefix);
        const int64_t nA = c->nActiveVs, nP = c->nPressures, nT = c->nStresses, R = c->regionCount;
        auto sp = fetch(c, c->S.ptr.p, c->S.rows + 1);
        auto sc = fetch(c, c->S.col.p, c->S.nnz);
        c->ensureValues(c->S);
        auto sv = fetch(c, c->S.val.p, c->S.nnz);
        auto rface = fetch(c, c->rrowFace.p, c->nReducedRows);
        auto rreg = fetch(c, c->rrowRegion.p, c->nReducedRows);
        auto com = fetch(c, c->COM.p, R * 3);
        // device storage is in the internal block-interleaved numbering; files are written in reference order
        auto permRow = fetch(c, c->permRow.p, nA);
        auto permSys = fetch(c, c->permSys.p, nP + nT);
        std::vector<int32_t> invSys((size_t)(nP + nT));
        for (int64_t i = 0; i < nP + nT; ++i) invSys[(size_t)permSys[(size_t)i]] = (int32_t)i;
        // G, Dt
        for (int which = 0; which < 2; ++which) {
            std::vector<int64_t> ptr((size_t)nA + 1, 0);
            std::vector<int32_t> col;
            std::vector<double> val;
            std::vector<std::pair<int32_t, double>> ent;
            for (int64_t r = 0; r < nA; ++r) {
                ptr[(size_t)r] = (int64_t)val.size();
                const int row = permRow[(size_t)r];
                ent.clear();
                for (int p = sp[(size_t)row]; p < sp[(size_t)row + 1]; ++p) {
                    const int32_t rc = invSys[(size_t)sc[(size_t)p]];
                    const bool isP = rc < nP;
                    if ((which == 0) == isP) ent.push_back({isP ? rc : (int32_t)(rc - nP), sv[(size_t)p]});
                }
                std::sort(ent.begin(), ent.end());
                for (auto& e : ent) { col.push_back(e.first); val.push_back(e.second); }
            }
            ptr[(size_t)nA] = (int64_t)val.size();
            writeMarketSparse(pre + (which == 0 ? "Mat_G.mtx" : "Mat_Dt.mtx"), nA, which == 0 ? nP : nT, ptr, col, val);
        }
        // JG, JDt: row 26 r + n collects C_f[n] * S_f,j over the region's reduced rows
        for (int which = 0; which < 2; ++which) {
            std::vector<std::map<int32_t, double>> rowsM((size_t)R * PS_RD);
            for (int64_t rr = 0; rr < c->nReducedRows; ++rr) {
                int i, j, k, a;
                unpackFace(rface[(size_t)rr], i, j, k, a);
                const int reg = rreg[(size_t)rr];
                double pnt[3] = {(double)(i + c->gOff[0]), (double)(j + c->gOff[1]), (double)(k + c->gOff[2])};
                pnt[a] -= 0.5;
                double C[PS_RD];
                basisRow(pnt[0] * c->dx - com[(size_t)reg * 3 + 0], pnt[1] * c->dx - com[(size_t)reg * 3 + 1],
                         pnt[2] * c->dx - com[(size_t)reg * 3 + 2], a, C);
                const int64_t row = nA + rr;
                for (int p = sp[(size_t)row]; p < sp[(size_t)row + 1]; ++p) {
                    const int32_t rc = invSys[(size_t)sc[(size_t)p]];
                    const bool isP = rc < nP;
                    if ((which == 0) != isP) continue;
                    const int32_t cc = isP ? rc : (int32_t)(rc - nP);
                    for (int n = 0; n < PS_RD; ++n) rowsM[(size_t)reg * PS_RD + n][cc] += sv[(size_t)p] * C[n];
                }
            }
            std::vector<int64_t> ptr((size_t)R * PS_RD + 1, 0);
            std::vector<int32_t> col;
            std::vector<double> val;
            for (size_t r = 0; r < rowsM.size(); ++r) {
                ptr[r] = (int64_t)val.size();
                for (auto& kv : rowsM[r]) { col.push_back(kv.first); val.push_back(kv.second); }
            }
            ptr[rowsM.size()] = (int64_t)val.size();
            writeMarketSparse(pre + (which == 0 ? "Mat_JG.mtx" : "Mat_JDt.mtx"), R * PS_RD, which == 0 ? nP : nT, ptr, col, val);
        }
        auto refRows = [&](const double* dptr) {   // active-row vector -> reference order
            auto v = fetch(c, dptr, nA);
            std::vector<double> o((size_t)nA);
            for (int64_t i = 0; i < nA; ++i) o[(size_t)i] = v[(size_t)permRow[(size_t)i]];
            return o;
        };
        auto refSys = [&](const double* dptr, int64_t off, int64_t n) {   // system vector slice -> reference order
            auto v = fetch(c, dptr, nP + nT);
            std::vector<double> o((size_t)n);
            for (int64_t i = 0; i < n; ++i) o[(size_t)i] = v[(size_t)permSys[(size_t)(off + i)]];
            return o;
        };
        auto diagOut = [&](const char* name, const std::vector<double>& v) {
            const int64_t n = (int64_t)v.size();
            std::vector<int64_t> ptr((size_t)n + 1);
            std::vector<int32_t> col((size_t)n);
            for (int64_t i = 0; i <= n; ++i) ptr[(size_t)i] = i;
            for (int64_t i = 0; i < n; ++i) col[(size_t)i] = (int32_t)i;
            writeMarketSparse(pre + name, n, n, ptr, col, v);
        };
        diagOut("Mat_McInv.mtx", refRows(c->McInv.p));
        diagOut("Mat_uInv.mtx", refSys(c->uInv.p, nP, nT));
        if (c->P.exportComponentMatrices) { diagOut("Mat_Mc.mtx", refRows(c->Mc.p)); diagOut("Mat_u.mtx", refSys(c->uDiag.p, nP, nT)); }
        auto blockOut = [&](const char* name, const double* dptr) {
            auto v = fetch(c, dptr, R * PS_RD * PS_RD);
            std::vector<int64_t> ptr((size_t)R * PS_RD + 1);
            std::vector<int32_t> col((size_t)R * PS_RD * PS_RD);
            for (int64_t r = 0; r <= R * PS_RD; ++r) ptr[(size_t)r] = r * PS_RD;
            for (int64_t r = 0; r < R; ++r)
                for (int m = 0; m < PS_RD; ++m)
                    for (int n = 0; n < PS_RD; ++n) col[(size_t)((r * PS_RD + m) * PS_RD + n)] = (int32_t)(r * PS_RD + n);
            writeMarketSparse(pre + name, R * PS_RD, R * PS_RD, ptr, col, v);
        };
        blockOut("Mat_Mr.mtx", c->Mr.p);
        blockOut("Mat_JDtuDJ.mtx", c->Kv.p);
        blockOut("Mat_Inv_Mr_plus_2JDtuDJ.mtx", c->Binv.p);
        {   // Mr_plus_2JDtuDJ = Mr/dt + 2 K (assembleReducedCombinedBlock, AssembleBlocks.cpp:147-205), formed like k_binv does
            auto mr = fetch(c, c->Mr.p, R * PS_RD * PS_RD), kv = fetch(c, c->Kv.p, R * PS_RD * PS_RD);
            std::vector<double> bsum(mr.size());
            for (size_t i = 0; i < mr.size(); ++i) bsum[i] = mr[i] * c->invDt + 2. * kv[i];
            std::vector<int64_t> ptr((size_t)R * PS_RD + 1);
            std::vector<int32_t> col((size_t)R * PS_RD * PS_RD);
            for (int64_t r = 0; r <= R * PS_RD; ++r) ptr[(size_t)r] = r * PS_RD;
            for (int64_t r = 0; r < R; ++r)
                for (int m = 0; m < PS_RD; ++m)
                    for (int n = 0; n < PS_RD; ++n) col[(size_t)((r * PS_RD + m) * PS_RD + n)] = (int32_t)(r * PS_RD + n);
            writeMarketSparse(pre + "Mat_Mr_plus_2JDtuDJ.mtx", R * PS_RD, R * PS_RD, ptr, col, bsum);
        }
        {   // a member the live path of the reference never fills but still writes: MrInv (assembleReducedMassBlockInverse is only
            // called from the unreachable Eq-14 preconditioner, Preconditioners.cpp:48) is a default 0x0 matrix.
            // uRed / uInvRed / JDtRed (dead in the live path) are not built here.
            const std::vector<int64_t> p0(1, 0);
            writeMarketSparse(pre + "Mat_MrInv.mtx", 0, 0, p0, {}, {});
        }
        writeMarketVector(pre + "Vec_activeRHS.mtx", refRows(c->rhsA.p));
        writeMarketVector(pre + "Vec_reducedRHS.mtx", fetch(c, c->rhsR.p, R * PS_RD));
        writeMarketVector(pre + "Vec_pressureRHS.mtx", refSys(c->rhsPT.p, 0, nP));
        writeMarketVector(pre + "Vec_stressRHS.mtx", refSys(c->rhsPT.p, nP, nT));
        writeMarketVector(pre + "Vec_b.mtx", refSys(c->b.p, 0, nP + nT));
        if (c->isSolved) writeMarketVector(pre + "solutionVector.mtx", refSys(c->x.p, 0, nP + nT));
        return PS_SUCCESS;
    })
}

// exportMatrices + exportMatricesPostSolve (Solver.cpp:533-572): Mat_A, Vec_b, Vec_guess (the warm-start vector of
// constructGuessVectors, :461-467; zero with useWarmStart off) and, after a solve, solutionVector.
// Mat_A: with solverType PCG_MATRIX_VECTOR_PRODUCTS the reference resizes A to n x n and leaves it empty
// (assembleSystemPressureStressFactored, AssembleSystem.cpp:445) — so does this; with solverType EIGEN it assembles the
// explicit operator (assembleSystemPressureStress, :351-430) and so does this, on the host from the device blocks:
//   A = -dt S_a^T McInv S_a - (J^T S_r)^T BInv (J^T S_r) - 1/2 diag(0, uInv)
// (setFromTriplets semantics: duplicates summed, explicit zeros kept out).  The reduced part is dense per tile; export tooling.
int32_t ps_export_matrices(ps_context* c, const char* prefix) {
    if (!c || !prefix) return PS_FAILED;
    PS_TRY(c, {
        if (!c->isSetup) throw Error("not set up");
        HIP_CHECK(hipSetDevice(c->device));
        const std::string pre(prefix);
        const int64_t n = c->nPressures + c->nStresses;
        auto permSys = fetch(c, c->permSys.p, n);
        auto refSys = [&](const double* dptr) {
            auto v = fetch(c, dptr, n);
            std::vector<double> o((size_t)n);
            for (int64_t i = 0; i < n; ++i) o[(size_t)i] = v[(size_t)permSys[(size_t)i]];
            return o;
        };
        if (c->P.solverType != PS_EIGEN) {
            const std::vector<int64_t> pn((size_t)n + 1, 0);
            if (!writeMarketSparse(pre + "Mat_A.mtx", n, n, pn, {}, {})) throw Error("cannot write Mat_A.mtx");
        } else {
            std::vector<int64_t> ap; std::vector<int32_t> ac; std::vector<double> av;
            c->buildExplicitA(ap, ac, av);
            if (!writeMarketSparse(pre + "Mat_A.mtx", n, n, ap, ac, av)) throw Error("cannot write Mat_A.mtx");
        }
        writeMarketVector(pre + "Vec_b.mtx", refSys(c->b.p));
        writeMarketVector(pre + "Vec_guess.mtx", refSys(c->guess.p));
        if (c->isSolved) writeMarketVector(pre + "solutionVector.mtx", refSys(c->x.p));
        return PS_SUCCESS;
    })
}

}  // extern "C"
