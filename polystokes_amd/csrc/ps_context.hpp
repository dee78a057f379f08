// The host-side mirror of the reference's `HDK_PolyStokes::Solver` (exec/HDK_PolyStokesSolver.h:27-375):
// same stage methods, same state, but every field lives in HBM and every stage is a HIP kernel launch.
#pragma once
#include <map>
#include <memory>

#include "ps_common.hpp"

namespace ps {

struct DevCSR {
    int64_t rows = 0, cols = 0, nnz = 0;
    DevBuf<int32_t> ptr;   // rows+1 (nnz < 2^31 is enforced)
    DevBuf<int32_t> col;
    DevBuf<double> val;
    // Lossless value coding: every stencil value is code * scale with an integer |code| <= 127 whenever the volume
    // fractions are multiples of 1/8 (the 2x2x2 sampler): val = +-(wF*wL)/dx = +-(8wF * 8wL) * (1/(64 dx)).
    // The fill kernels verify code*scale == val bit for bit; if any entry fails, the SpMV streams `val` instead.
    DevBuf<int8_t> code;
    bool packed = false;
    // Compressed stream of the persistent SpMV kernels (with `packed`, or with val4; only if EVERY chunk fits: col16ok).
    // A chunk = up to 256 consecutive rows [chunkInfo.z, chunkInfo.z + chunkInfo.w) and a 4-entry-aligned run [chunkInfo.x,
    // chunkInfo.y) of (col16, code4) entries in CSR order; col = winBase[chunk*16 + (c16 >> 12)] + (c16 & 4095) (up to 16 windows
    // of 4096 columns per chunk); len8 = entries per row.  nv = ceil(fullest chunk / 1024) = 4-entry groups per lane.
    // Chunks start at the boundaries of the numbering's lattice blocks (and of the tiles' skin-row ranges): equivalent blocks then
    // produce byte-identical runs, and a chunk whose run equals an earlier chunk's points at THAT run (ps_blocks.hip:
    // dedupChunks) — a periodic tile structure streams one copy of each distinct run, from cache.
    DevBuf<uint16_t> col16;
    DevBuf<int8_t> code4;
    DevBuf<double> val4;         // the fp64 values in the same aligned chunk layout (only when the values are not coded)
    DevBuf<int32_t> winBase;
    DevBuf<int4> chunkInfo;      // (run begin, entries | rows << 16, first row, first row of the run's owner) — ps_kernels_spmv.hpp:Chunk
    DevBuf<uint8_t> len8;
    DevBuf<int32_t> chunkRep;    // the chunk whose run a chunk points at (itself unless shared)
    int nChunks = 0;
    int64_t uniqueLen = 0;       // entries of the runs some chunk actually points at (== streamLen without sharing)
    bool col16ok = false;
    int nv = 2;
    int64_t streamLen = 0;       // entries of col16 / code4 (multiple of 4)
    // Row-per-lane form of the same stream (coded values only; ps_blocks.hip:buildEll, ps_kernels_spmv.hpp:k_spmv_*_ell).  A chunk's
    // rows are cut into four units of 64 consecutive rows, one per wave; a unit is padded to the width W (even, <= 8) of its longest
    // row and stored lane-major: lane l of the unit finds its row's W windowed 16-bit columns at ecol[unitBegin + l*W ..] and its W
    // value codes (padded to 4 or 8 bytes) at ecode[...].  Gather instruction k of a wave then covers entry k of 64 CONSECUTIVE rows:
    // the lanes of a quad address neighbouring columns, which the CU's L1 pipeline serves per distinct line, not per lane
    // (profiles/r03_spmv_issue.md).  Each lane keeps its row's sum in a register: no LDS round trip, no row-length scan, no barrier.
    // echunk: (col begin [u16 units], code begin [bytes], first row, rows | W0 << 12 | W1 << 16 | W2 << 20 | W3 << 24).
    DevBuf<uint16_t> ecol;
    DevBuf<int8_t> ecode;
    DevBuf<int4> echunk;         // (the 16 window bases per chunk are winBase: same greedy cover, same columns)
    bool ellok = false;
    int64_t ellCols = 0, ellCodes = 0;     // sizes of ecol (u16 units) / ecode (bytes)
    int64_t ellUniqueCols = 0;             // u16 units of the runs some chunk actually points at (== ellCols without sharing)
};

// device-resident CG scalars (no host round trip inside the iteration)
struct CGScalars {
    double rsold, pAp, alpha, beta, rr, xx, rz, rre;
    int iter;        // index of the iteration that converged (pcg.h:322-325)
    int done;        // 1 once the stop rule fired (or rsold == 0)
    int maxit;
    int pend;        // the stop test of iteration `pendIter` waits for ||x||^2 of the x it updated (deferred-x step)
    double tol2;
    int pendIter;
    int vecNT;       // 1: the vector kernels use non-temporal loads / stores (ps_context::ntLevel() == 2)
    double rsold2[2];   // r.z of the previous iteration, double-buffered by iteration parity (fused-scalar step kernels)
};

struct ArrayInfo {
    const void* dptr;
    int64_t count;
    int elem;
    const int32_t* perm = nullptr;   // optional: out[i] = src[perm[permOffset + i]] (reference order view)
    int64_t permOffset = 0;
};

}  // namespace ps

struct ps_context {
    // buffers this context dropped or grew inside a step, waiting for hipFree (ps_common.hpp: DeferredFrees).  First member: alive until the
    // last DevBuf below has gone.
    ps::DeferredFrees deferred;
    ps_context() { ps::MemState& M = ps::memState(); std::lock_guard<std::mutex> lk(M.m); M.lists.push_back(&deferred); ++M.contexts; }
    ~ps_context() {
        { ps::MemState& M = ps::memState(); std::lock_guard<std::mutex> lk(M.m); --M.contexts; for (size_t q = 0; q < M.lists.size(); ++q) if (M.lists[q] == &deferred) { M.lists.erase(M.lists.begin() + (long)q); break; } }
        ps::releaseDeferred(deferred);      // (ps_context_destroy has synchronised the stream)
    }
    ps_context(const ps_context&) = delete;
    ps_context& operator=(const ps_context&) = delete;
    // Release the dropped buffers: called where the stream has just been synchronised (end of setup / solve / step: streamIsIdle) and on
    // error paths (then only if the stream really is idle).  Never inside a step of threaded ranks (ps_common.hpp).
    void drainDeferred(bool streamIsIdle) {
        if (ps::threadedRanks()) return;
        if (!streamIsIdle && (!stream || hipStreamQuery(stream) != hipSuccess)) { (void)hipGetLastError(); return; }
        ps::releaseDeferred(deferred);
        ps::releaseDeferred(ps::orphanFrees());
    }
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- parameters / geometry (Solver.cpp:18-68) ----
    ps_params P{};
    ps::Grid g{0, 0, 0, 0};
    double dx = 0, invDx = 0, dt = 0, invDt = 0, rho = 0;
    double valScale = 0;      // invDx / 64
    bool forceFp64Values = false;   // env PS_FORCE_FP64_VALUES=1: never use the coded values (A/B and fallback testing)
    bool haveInputWeights = false;
    bool viscUniform = false;       // the viscosity field is one value everywhere (checked at upload): samplers return it without loads
    float viscUniformValue = 0.f;
    bool uploaded = false, isSetup = false, isSolved = false;

    // ---- inputs (fp32 Houdini voxel arrays, HDK_PolyStokes.C:235-246) ----
    ps::DevBuf<float> surface, collision, viscosity, vel[3], cvel[3];
    ps::DevBuf<float> velOut[3], valid[3];

    // ---- weights, labels, indices (Solver.h:316-335) ----
    ps::DevBuf<float> liquidW[7], fluidW[7];
    ps::DevBuf<int32_t> labels[7], activeIdx[7], reducedIdx[7];
    ps::DevBuf<int32_t> faceRow[3];          // face -> internal row of S (-1: none)
    // Internal (solver) numbering: DOFs and face rows interleaved by 16^3 spatial block so that the SpMV
    // gathers stay in L2.  sysIdx[0]: cell -> base (p, txx, tyy, tzz = base+0..3); sysIdx[4..6]: edge -> index.
    // permSys[ref system index] = internal index; permRow[ref active-face index] = internal row.
    ps::DevBuf<int32_t> sysIdx[7];
    ps::DevBuf<int32_t> sysIdxT[3];          // cell -> internal index of its txx / tyy / tzz (sysIdx[0]: its pressure)
    int ilPlaneMajor = 0;                    // bit 0: DOFs, bit 1: face rows numbered kind-major inside every k-plane of a lattice block (buildInternalNumbering)
    ps::DevBuf<int32_t> permSys, permRow;
    int ilBlocks = 0;                        // lattice blocks in sequence order (incl. the padding blocks of partial super-blocks)
    std::vector<int32_t> blockStartSys, blockStartRow;   // first DOF / first active row of every lattice block (+ total): host copies
    ps::DevBuf<int32_t> cellScratch[3];      // layer marks / CC labels / fix flags
    ps::DevBuf<int32_t> scanBlock;           // block sums for scans
    ps::DevBuf<int32_t> counters;            // small device counters (flags, totals)

    // ---- counts (Solver.h:272-285) ----
    int64_t nCenter = 0, nFace[3] = {0, 0, 0}, nEdge[3] = {0, 0, 0};
    int64_t nActiveVs = 0, nReducedVs = 0, nPressures = 0, nStresses = 0, nSystem = 0, nTotalDOFs = 0;
    int64_t regionCount = 0;
    int64_t nReducedRows = 0;   // reduced faces that carry at least one stencil entry
    int64_t nRows = 0;          // nActiveVs + nReducedRows

    // ---- per-region data (Solver.h:339, 705-706) ----
    ps::DevBuf<int32_t> bbox;                // R*6: min xyz, max xyz (cells)
    std::vector<int32_t> hbbox;
    ps::DevBuf<double> COM, cfit, Mr, Kv, Binv, rhsR, regionScratch;
    // work tables for per-region reductions over the region's face box (host built, small)
    ps::DevBuf<int32_t> fbItemRegion, fbItemAxis, fbItemStart;   // one item = <=FB_CHUNK face-box positions
    ps::DevBuf<int32_t> fbRegionItemPtr;                         // R+1
    int64_t fbItems = 0;
    // classes of identical tiles (ps_tiles.hip:buildTileClasses): tileRep[r] = the region whose Mr / K region r copies (itself: sums its own);
    // the face-box items of the representatives alone
    ps::DevBuf<unsigned long long> tileSig; ps::DevBuf<int32_t> tileUnique, tileRep, repItemRegion, repItemAxis, repItemStart, repRegionItemPtr, repList;
    int64_t tileReps = 0, repItems = 0;
    void buildTileClasses();
    // skin-row enumeration: items of <= FB_CHUNK positions of a region's UNION face box (ex+1)(ey+1)(ez+1); a position yields up
    // to three rows (its X, Y, Z face) — rows are ordered (region, position x-fastest, axis), so the faces hanging off one
    // voxel are adjacent rows like the active ones
    ps::DevBuf<int32_t> sbItemRegion, sbItemStart, sbItemCount, sbItemLong;   // sbItemLong: long (> 2 entries) skin rows of the item, numbered first
    std::vector<int32_t> skinCutsHost;                           // skin-row offsets where a chunk of S's stream may start (item starts, ends of the long rows)
    std::vector<int32_t> sbRegionItemPtrHost;                    // R+1
    int64_t sbItems = 0;
    int64_t maxRegionRows = 0;                                   // fullest region: decides fused / three-kernel tile apply
    ps::DevBuf<double> partials;                                  // items * 676 (or rows chunks * 26)

    // reduced rows of S: region-contiguous, deterministic order
    ps::DevBuf<uint32_t> rrowFace;           // packed (i,j,k,axis)
    ps::DevBuf<int32_t> rrowRegion;
    ps::DevBuf<int32_t> regionRowPtr;        // R+1, offsets into reduced rows
    ps::DevBuf<int32_t> rchunkRegion, rchunkStart, rchunkEnd, regionChunkPtr;  // <= RC_ROWS rows of ONE region per chunk (three-kernel tile apply)
    int64_t nRChunks = 0;
    bool bboxValid = false;      // bbox[] holds the boxes of the final regions (set by the small-region fix, cleared per setup)

    // ---- blocks (Solver.h:337-369): S = [G Dt ; Ghat Dhat] by face row, St its transpose ----
    ps::DevCSR S, St;
    ps::DevBuf<double> McInv, rhsA, uInv, rhsPT, Mc, uDiag, oldVs;
    // Value-set coding of the two diagonals the SpMV epilogues read (ps_blocks.hip:buildDiagonalCodes): when a diagonal takes at
    // most 256 distinct values (constant viscosity: uInv = invVisc * {volume fractions}; McInv = 1 / (rho * k/64)) the kernels
    // read a 1-byte code per row and look the fp64 value up in a 256-entry table held in LDS — the same bits, 7 bytes less per
    // row.  Decided per setup; any 257th value keeps the fp64 array (variable viscosity fields).  PS_NO_DIAG_CODES=1 disables.
    ps::DevBuf<uint8_t> uCode, mcCode;
    ps::DevBuf<double> uDict, mcDict;      // 256 entries each
    bool uCoded = false, mcCoded = false;
    int32_t diagFlagsHost = 0;
    int32_t fusedStepHost = 0;
    int32_t streamRunsHost[4] = {0, 0, 0, 0};
    int32_t rowPerLaneHost[2] = {0, 0};
    void buildDiagonalCodes();
    ps::DevBuf<ps::diag_t> dinvF;   // the Jacobi diagonal as the PCG kernels read it (16-bit storage: ps_common.hpp diag_t, see constructPreconditioner)
    ps::DevBuf<double> b, x, r, pvec, Ap, dinv, ts, vreg, wreg, recovered, tmp1, tmp2, tmp3, tmp4, tmp5;
    ps::DevBuf<double> chebPartials, chebPartials2;   // r.z partials of the Chebyshev polynomial's last term
    double chebLmax = 8.4;                            // upper end of the Chebyshev interval (estimateLambdaMax)
    bool chebInner32Req = false;                      // the caller asked for PS_PRE_CHEBYSHEV_F32 (P.preconditioner then holds PS_PRE_CHEBYSHEV)
    bool chebInner32 = false;                         // ... and this system runs it: the polynomial's z_j and face-row vector are stored as fp32 (ps_solve.hip: chebyshevApply)
    int32_t chebInner32Host = 0;
    ps::DevBuf<double> guess;   // [pressureGuess; stressGuess] of constructGuessVectors (Solver.cpp:512-531), internal numbering
    // dotPartials: p.Ap partials of the St kernel; dotPartials2: their first-stage sums (one-shot St kernel only);
    // dotPartialsR: r.r / r.z partials of k_cg_update_r; dotPartials3: x.x partials of k_cg_update_xp.  Separate buffers:
    // every block of a step kernel sums its predecessor's partials while other blocks already write this kernel's.
    ps::DevBuf<double> dotPartials, dotPartials2, dotPartials3, dotPartialsR;
    ps::DevBuf<double> fusedPart;   // fused step (ps_solve.hip): S-kernel, tile, uInv p^2 and r.r/r.z partials, in that order
    ps::DevBuf<ps::CGScalars> scal;

    // ---- multi-GPU (slab decomposition; ps_dist.hip) ----
    bool slabEnabled = false;                // a decomposition is set (slab or brick) and world > 1
    int gOff[3] = {0, 0, 0};    // global index of the local cell (0, 0, 0): face positions use global indices, so a tile's matrices do not depend on the decomposition
    ps_slab slab{};             // rank / world (and the z-range when ps_set_slab was used)
    ps_brick brick{};           // the owned box in local coordinates, neighbours, global position (ps_set_slab fills it too)
    int64_t ownLo = 0, ownHi = 0;            // owned DOF range in the internal numbering (contiguous: the owned lattice blocks come first)
    ps::DevBuf<int32_t> regionOwned;         // R flags
    ps::DevBuf<int32_t> blockMap;            // sequence position -> lattice block (owned blocks first); empty: lattice order
    int blockMapOwned = -1, blockMapFor = 0; // owned lattice blocks (-1: the map has to be rebuilt: ps_set_brick), blocks it was built for
    // exchange lists per LINK (internal system indices, canonical order over the cut's cross-section): the neighbour's samples my rows touch (halo)
    // and mine the neighbour's rows touch (own), below and above.  Links 0..2: the face neighbours along x, y, z.  Links 3..5 (one-round mode
    // only, r05): the DIAGONAL neighbours across two cuts — (x, y), (x, z), (y, z) — "below" = one brick down along both axes, "above" = one up along
    // both.  Only one kind of sample crosses a diagonal: the edge stress that lives on both cut planes (XY / XZ / YZ edges on the corner line), which
    // the skin rows of a tile that ends at both planes touch (a tile's rows belong to the tile's owner whatever plane they lie on) — so a diagonal
    // link has an upper halo list and a lower own list, the other two are empty.
    static constexpr int NLINK = 6;
    ps::DevBuf<int32_t> listLowHalo[NLINK], listLowOwn[NLINK], listUpHalo[NLINK], listUpOwn[NLINK];
    int64_t nLowHalo[NLINK] = {0, 0, 0, 0, 0, 0}, nLowOwn[NLINK] = {0, 0, 0, 0, 0, 0}, nUpHalo[NLINK] = {0, 0, 0, 0, 0, 0}, nUpOwn[NLINK] = {0, 0, 0, 0, 0, 0};
    ps::DevBuf<double> sendLo[NLINK], sendUp[NLINK], recvLo[NLINK], recvUp[NLINK], redbuf;
    static void linkAxes(int l, int& a, int& b) { a = l < 3 ? l : (l == 5 ? 1 : 0); b = l < 3 ? -1 : (l == 3 ? 1 : 2); }
    int axisStride(int a) const { return a == 0 ? 1 : (a == 1 ? brick.dims[0] : brick.dims[0] * brick.dims[1]); }
    bool linkLower(int l) const { int a, b; linkAxes(l, a, b); return brick.hasLower[a] && (b < 0 || brick.hasLower[b]); }
    bool linkUpper(int l) const { int a, b; linkAxes(l, a, b); return brick.hasUpper[a] && (b < 0 || brick.hasUpper[b]); }
    int nbrLo(int l) const { int a, b; linkAxes(l, a, b); return linkLower(l) ? brick.rank - axisStride(a) - (b >= 0 ? axisStride(b) : 0) : -1; }
    int nbrUp(int l) const { int a, b; linkAxes(l, a, b); return linkUpper(l) ? brick.rank + axisStride(a) + (b >= 0 ? axisStride(b) : 0) : -1; }
    int64_t exchangeEntries() const { int64_t n = 0; for (int a = 0; a < NLINK; ++a) n += nLowOwn[a] + nUpOwn[a] + nLowHalo[a] + nUpHalo[a]; return n; }
    // Overlap of the halo exchanges with the rows that do not need them (ps_dist.hpp: Dist::solve).  Chunk lists of the row-per-lane
    // kernels: [0] S chunks without a halo column, [1] S chunks with one; [2] St chunks holding halo rows with entries (their A p goes
    // to the neighbour), [3] St chunks of owned rows only.  St chunks of halo rows without entries are in neither: never launched.
    ps::DevBuf<int32_t> distList[5];          // [4]: the St chunks of [2] and [3] together, in chunk order — the ONE St launch of a rank whose exchanges are not overlapped
    int nDistList[5] = {0, 0, 0, 0, 0};
    bool distListsOk = false;
    hipStream_t commStream = nullptr;        // transports run here (= stream for in-process ranks: nothing to overlap on one stream)
    hipEvent_t distEv[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // what one distributed solve did (ps_dist_stats): bytes per iteration over the cuts, sampled transport / all-reduce times
    double distStats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ps::DevBuf<int32_t> labelFlags;          // Dist::exchangeLabels: labels changed, REDUCED cells without a component
    std::vector<int32_t> hostOwnList[2 * NLINK];   // host copies of listLowOwn / listUpOwn (index 2 l / 2 l + 1): Dist::buildFixup merges them per DOF
    ps::DevBuf<int32_t> fixDof, fixSrc;      // the merged fix-up of the fused step (k_dist_fixup_merged): DOFs that receive contributions, their sources
    ps::DevBuf<const double*> fixBufs;       // ... and the table of the twelve receive buffers
    int64_t nFix = 0;
    ps::DevBuf<unsigned char> scrMark;       // setup scratch of Dist::decideExchangeMode
    // in-process groups (all ranks of the group on one device and one stream): the rows of ALL ranks — what passes through the device's caches per
    // iteration — decide the cache policy (ntLevel), not this rank's share; and an exchange is ONE kernel for all ranks and links that gathers from
    // the senders' vectors and scatters into the receivers' (Dist::buildDirectExchange; table on rank 0: [0] values of p out, [1] contributions of A p back)
    int64_t deviceShareRows = 0;
    ps::DevBuf<unsigned char> xsegTab[2];
    int nXseg[2] = {0, 0}, xsegBlocks[2] = {0, 0};
    bool haloForward = false;                // the exchange lists carry the copies of an earlier axis's exchange (three forwarding rounds x -> y -> z); false: every
                                             // list holds the sender's OWN samples only and the three axes travel in ONE round (ps_dist.hpp: Dist::decideExchangeMode)
    int64_t haloLabelChanges = 0;            // halo cells whose label the owners' exchange changed in the last setup (both passes)
    void* rcclComm = nullptr;                // ncclComm_t when one process per GPU
    void* hostComm = nullptr;                // host-staged TCP transport (ps_comm_init_tcp): same algorithm without RCCL
    uint64_t hashLowHalo[NLINK] = {0, 0, 0, 0, 0, 0}, hashLowOwn[NLINK] = {0, 0, 0, 0, 0, 0}, hashUpHalo[NLINK] = {0, 0, 0, 0, 0, 0}, hashUpOwn[NLINK] = {0, 0, 0, 0, 0, 0};   // order-sensitive hashes of the lists' global keys
    ps::DevBuf<ps::CGScalars> benchScal;     // scratch of ps_bench_kernel
    ps::DevBuf<double> benchOnes, benchZeros;
    bool ownsStream = true;
    ps::DevBuf<float> ownedFace[3];          // 1 where this rank is responsible for the output face
    ps::Own own() const {
        ps::Own o;
        o.enabled = slabEnabled ? 1 : 0;
        for (int a = 0; a < 3; ++a) { o.lo[a] = brick.lo[a]; o.hi[a] = brick.hi[a]; o.hasUpper[a] = brick.hasUpper[a]; }
        return o;
    }
    void buildHaloLists();                   // ps_grid.hip

    ps_interrupt_fn interruptCb = nullptr;   // polled between CG batches (UT_Interrupt equivalent)
    void* interruptUser = nullptr;
    bool interrupted = false;

    // ---- results ----
    int solveIterations = -1;
    double solveError = -1;
    int usedBiCGStab = 0;
    ps_stats lastStats{};

    std::map<std::string, ps::ArrayInfo> arrays;
    std::map<std::string, std::vector<char>> hostArrays;   // materialised-on-demand exports

    // ---- stage methods: names follow exec/HDK_PolyStokesSolver.h:96-190 ----
    void upload(const ps_params* p, const ps_fields_in* in);
    void buildIntegrationWeightsAlt();                    // ps_grid.hip
    void classifyCells();
    void constructReducedRegions();
    void constructOnlyActiveRegions();
    void classifyFaces();
    void classifyEdges();
    void constructCenterReducedIndices(int part);         // 0: components + fixReducedRegionBoundaries, 1: fixSmallReducedRegions
    void constructFacesReducedIndices();
    void constructEdgesReducedIndices();
    void constructActiveIndices();
    void buildValidFaces();
    void computeRegionBoxes();                            // ps_tiles.hip
    void computeCenterOfMasses();
    void computeLeastSquaresFits();
    void computeReducedMassMatrices();
    void computeReducedViscosityMatricesInteriorOnly();
    void assembleReducedBlocks();                         // AssembleBlocks.cpp:147-244,356-367
    void constructMatrixBlocks();                         // ps_blocks.hip
    void buildCol16(ps::DevCSR& M, int counterSlot, const std::vector<int32_t>& cuts, const uint8_t* rowCode, int codeRows);   // ps_blocks.hip; cuts: row indices where a chunk should start
    void buildEll(ps::DevCSR& M);                         // the row-per-lane form of M's compressed stream (ps_blocks.hip)
    void buildStreams(bool share);                        // both compressed streams (ps_blocks.hip)
    bool shareRuns = true;
    ps::DevBuf<int2> scrChunkRows; ps::DevBuf<int32_t> scrSlice, scrStart4, scrVals, scrKeep, scrRemap, scrEllCol, scrEllCode, scrEllW; ps::DevBuf<unsigned long long> scrHash, scrKeys, scrUniq;   // buildCol16 scratch
    void ensureValues(ps::DevCSR& M);                    // decode the fp64 values of a coded block on demand (ps_blocks.hip)
    void buildVal4(ps::DevCSR& M);                        // fp64 values in the compressed stream's layout (fallback / A-B)
    std::vector<int32_t> regionRowPtrHost;                // R+1 offsets into the reduced rows (host copy)
    std::vector<int32_t> hostTab[20];                     // host-built tables of one setup, alive until the next: asynchronous uploads without a synchronisation
    void assembleSystemPressureStressFactored();          // ps_solve.hip
    void constructPreconditioner();
    int solve();
    void estimateLambdaMax();
    void applyPreconditionerDevice(const double* r, double* z, double* scratch);   // z = M^-1 r (parity hook, ps_apply_preconditioner)
    int chebyshevApply(const double* rvec, double* zA, double* zB, double* rzPartial, const ps::CGScalars* sc, bool firstDone = false, double** zOut = nullptr);
    double chebTheta() const;
    int ntLevel() const;
    int solveEigenCG();                                   // Solver.cpp:814-862 on the factored device operator
    void constructGuessVectors();                         // Solver.cpp:512-531
    // explicit A (AssembleSystem.cpp:351-430) in reference numbering, assembled on the host from the device blocks (export only)
    void buildExplicitA(std::vector<int64_t>& ptr, std::vector<int32_t>& col, std::vector<double>& val);
    void recoverVelocityFromPressureStress();
    void applySolutionToVelocity();
    void applyOperator(const double* x, double* y, double* dotPartialsOut);   // device pointers

    // orchestration (ps_context.hip)
    int setup(ps_stats* stats);
    void setupPhase(int phase);              // 0, 1, 2 in this order (setup() = all three); between them Dist exchanges the halo's cell labels
    std::shared_ptr<void> setupState;        // stage timer and clocks of the setup in flight
    int setupPhaseDone = -1;
    int solveStage(ps_stats* stats);
    void fillDimData(ps_stats* st) const;
    void registerArrays();

    // helpers
    int32_t orderedIndexAssign(int s, int mode, ps::DevBuf<int32_t>& out, int counterSlot = -1);   // ps_grid.hip
    int64_t interleavedIndexAssign(int ngroups, const int* samples, const int* weights, int32_t* const* outs);
    int64_t interleavedIndexAssignEx(int ngroups, const int* samples, const int* weights, int32_t* const* outs, bool ownFilter,
                                     int64_t* ownedRange);
    void buildInternalNumbering();                                            // ps_grid.hip
    int64_t exclusiveScanI32(int32_t* data, int64_t n, int counterSlot = -1);   // ps_grid.hip (counterSlot >= 0: total to counters[slot], no synchronisation)
    int32_t readCounter(int idx);
    void fetchCounters(int idx, int n, int32_t* out);   // counters[idx .. idx + n) in one round trip through the page-locked mirror (synchronises the stream: copies queued before it have landed too)
    int32_t* pinnedCounters = nullptr;       // page-locked mirror of `counters` (64 words): a count read lands there directly instead of through the runtime's staging copy
    void zeroCounters();
};

// kernel micro-benchmark dispatch (ps_solve.hip), used by ps_bench_kernel
void ps_bench_launch(ps_context* c, const std::string& kernel, const double* x, double* y);
int ps_dist_step_single(ps_context* c, ps_stats* stats);   // ps_solve.hip: distributed step of one rank (RCCL or TCP transport)
void ps_dist_release(ps_context* c);                       // ps_solve.hip: destroys the rank's communicator / sockets

namespace ps {
constexpr int FB_CHUNK = 4096;   // face-box positions per work item (per-region dense reductions)
constexpr int PS_SCAN_TILE = 2048;   // entries per workgroup of the scan kernels (ps_grid.hip: SCAN_TILE; asserted equal there): sizes their block-sum scratch
constexpr int RC_ROWS = 1280;    // reduced rows per chunk in the three-kernel tile apply (regions too large for one workgroup)
constexpr int TILE_FUSED_MAX_ROWS = 32768;   // regions up to this many skin rows: one workgroup gathers, solves and expands (k_tile_apply)
}  // namespace ps
