/*
 * polystokes.h — C ABI of the MI355X-native PolyStokes hot path.
 *
 * Drop-in boundary for the reference's per-step reduced-viscosity Stokes solve
 * (panuelosj/polystokes).  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree, file:line).
 *
 * The boundary is plain C: POD structs, raw pointers and sizes.  Host buffers
 * are owned by the caller (Houdini owns its SIM fields, exec/HDK_PolyStokes.C:235-246);
 * device memory is owned by the opaque ps_context and reused across steps.
 *
 * Array layout (all dense, x-fastest, i + dim0*(j + dim1*k)):
 *   cell   fields : nx   * ny   * nz
 *   faceX  fields : (nx+1)* ny   * nz        faceY: nx*(ny+1)*nz     faceZ: nx*ny*(nz+1)
 *   edgeXY fields : (nx+1)*(ny+1)* nz        edgeXZ: (nx+1)*ny*(nz+1) edgeYZ: nx*(ny+1)*(nz+1)
 * (exec/HDK_PolyStokesSolver.h:294-314 are the matching out-of-bounds predicates.)
 */
#ifndef POLYSTOKES_H
#define POLYSTOKES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Reduced model per tile: a compile-time choice, as in the reference (lib/include/units.h:9-18).  Default QUADRATIC_REGIONS,
 * 26 divergence-free quadratic DOFs; -DPS_AFFINE_REGIONS builds the AFFINE_REGIONS variant, 11 DOFs
 * (exec/HDK_PolyStokesSolver.cpp:2153-2184) -> libpolystokes_hip_affine.so, same ABI, ps_reduced_dof() tells which. */
#ifdef PS_AFFINE_REGIONS
#define PS_REDUCED_DOF 11
#else
#define PS_REDUCED_DOF 26
#endif

/* exec/HDK_PolyStokesSolver.h:61-70  enum class SolverResult */
enum ps_result {
    PS_UNSUPPORTED_SOLVER = -4,
    PS_INCOMPLETE = -3,
    PS_INVALID = -2,
    PS_FAILED = -1,
    PS_NOCONVERGE = 0,
    PS_SUCCESS = 1,
    PS_NOCHANGE = 2
};

/* exec/HDK_PolyStokesSolver.h:71-82  enum MaterialLabels */
enum ps_label {
    PS_UNASSIGNED = -1,
    PS_UNSOLVED = -2,
    PS_GENERICFLUID = -3,
    PS_ACTIVEFLUID = -4,
    PS_SOLID = -5,
    PS_REDUCED = -6,
    PS_UNVISITED = -7,
    PS_VISITED = -8,
    PS_BOUNDARY = -9
};

/* lib/include/units.h:76-94 */
enum ps_matrix_scheme { PS_PRESSURE_STRESS = 0 };
enum ps_solver_type { PS_PCG_MATRIX_VECTOR_PRODUCTS = 0, PS_EIGEN = 1 };
/* lib/include/units.h:47-53; DIAGONAL is the empty stub at
 * exec/HDK_PolyStokesSolver_Preconditioners.cpp:37-41 that BASELINE.json asks for (Jacobi-PCG). */
/* PS_PRE_CHEBYSHEV (extension): z = q(D^-1 A) D^-1 r with q the degree-(k-1) Chebyshev polynomial
 * of the interval [lmax/PS_CHEB_INTERVAL_RATIO, lmax], D = diag(A), k = ps_params.preconditionerDegree (default 4): k-1 operator applies per
 * CG iteration, the same smoother-as-preconditioner idea as the reference's abandoned GS designs
 * (lib/src/Preconditioner.cpp:30-158) on the live pressure-stress operator.  lmax = max(8.4, 1.25 x the estimate of 10 power iterations at setup).
 * PS_PRE_CHEBYSHEV_F32 (r06): the same polynomial with its INNER vectors — the iterates z_j and the face-row vector of the k-1 inner operator
 * applies — stored in single precision (half the bytes of those applies; every product, sum and recurrence, the residual r, the outer PCG
 * and its stop rule stay fp64).  The preconditioner only approximates an inverse, so its storage rounding (6e-8 relative per stored value)
 * perturbs the iteration count (<= +5 % accepted; equal on the scenes measured), not the solution: x converges to the same tolerance.  It runs
 * where the row-per-lane two-unit kernels run (coded stencil values, a value-set coded McInv, single domain, >= 8 chunks); elsewhere — fallback
 * formats, decompositions — the fp64 form above runs (array "chebInner32" says which). */
enum ps_preconditioner { PS_PRE_IDENTITY = 1, PS_PRE_DIAGONAL = 5, PS_PRE_CHEBYSHEV = 6, PS_PRE_CHEBYSHEV_F32 = 7 };
#define PS_CHEB_INTERVAL_RATIO 250.0   /* lmax / lmin of the Chebyshev interval: flat optimum 120..1000 on the 256^3 scenes (30: 5 % slower) */
/* order in which serialAssignFieldIndices walks a field (Classifier.cpp:1738-1770):
 * 0 = UT_VoxelArray order (16^3 voxel tiles, tile-linear, x-fastest inside), 1 = plain x-fastest. */
enum ps_index_order { PS_ORDER_VOXEL_TILES = 0, PS_ORDER_LINEAR = 1 };

/*
 * Node parameters: one member per entry of the reference's PRM template
 * (exec/HDK_PolyStokes.C:88-208, accessors exec/HDK_PolyStokes.h:23-43), same names.
 * Field-name string parms stay in the Houdini shim; they have no meaning below it.
 */
typedef struct ps_params {
    double mindensity;                  /* 1      (unused by the live path, kept for the surface) */
    double maxdensity;                  /* 100000 */
    int32_t matrixSetup;                /* ps_matrix_scheme, 0 */
    int32_t solverType;                 /* ps_solver_type,   0 */
    int32_t doSolve;                    /* 1 */
    int32_t keepNonConvergedResults;    /* 1 */
    int32_t exportMatrices;             /* 0 */
    int32_t exportComponentMatrices;    /* 0 */
    int32_t exportStats;                /* 0 */
    int32_t useWarmStart;               /* 1 (built, then discarded: Solver.cpp:768) */
    double tolerance;                   /* 1e-3 */
    int32_t maxSolverIterations;        /* 5000 */
    int32_t useInputSurfaceWeights;     /* 1 (read, ignored by buildIntegrationWeightsAlt) */
    int32_t useInputCollisionWeights;   /* 1 (idem) */
    int32_t activeLiquidBoundaryLayerSize; /* 2 */
    int32_t activeSolidBoundaryLayerSize;  /* 2 */
    int32_t doReducedRegions;           /* 1 */
    int32_t doTile;                     /* 1 */
    int32_t tileSize;                   /* 16 */
    int32_t tilePadding;                /* 2 */
    /* --- extensions (not in the reference's template) --- */
    int32_t preconditioner;             /* ps_preconditioner, default PS_PRE_IDENTITY */
    int32_t indexOrder;                 /* ps_index_order, default PS_ORDER_VOXEL_TILES */
    int32_t negateCollision;            /* 1: `collision` is Houdini-convention (negative inside the
                                           solid) and is negated before the reference's
                                           computeSDFWeightsSampled(invert=false) call is applied
                                           (Solver.cpp:308-326); 0: use as given. default 1 */
    int32_t preconditionerDegree;       /* PS_PRE_CHEBYSHEV: terms k of the polynomial (k-1 applies); 0 = default 4 */
    const char* exportDataPrefix;       /* may be NULL */
} ps_params;

/* Inputs of solveGasSubclass (exec/HDK_PolyStokes.C:235-246, :319-320; Solver.cpp:39-44). */
typedef struct ps_fields_in {
    int32_t nx, ny, nz;
    double dx;                          /* max voxel size, HDK_PolyStokes.C:320 */
    double dt;                          /* timestep,       HDK_PolyStokes.C:319 */
    double orig[3];                     /* grid origin (debug output only, Solver.cpp:1134) */
    float density;                      /* constant liquid density, HDK_PolyStokes.C:298-304 */
    int32_t reserved;
    const float* vel[3];                /* face sampled velocity (in) */
    const float* surface;               /* cell, liquid SDF (<0 inside liquid) */
    const float* collision;             /* cell, solid SDF */
    const float* viscosity;             /* cell */
    const float* collisionvel[3];       /* face sampled */
    /* optional precomputed volume fractions, order:
     * 0 centerLiquid 1 faceXLiquid 2 faceYLiquid 3 faceZLiquid 4 edgeYZLiquid 5 edgeXZLiquid 6 edgeXYLiquid
     * 7..13 the same for Fluid.  All NULL -> the library samples the SDFs itself (Solver.cpp:238-326). */
    const float* weights[14];
} ps_fields_in;

/* Outputs (Solver.cpp:937-1028 velocity, Classifier.cpp:4-54 valid). */
typedef struct ps_fields_out {
    float* vel[3];                      /* may alias ps_fields_in.vel */
    float* valid[3];
} ps_fields_out;

/* exportStats(): dimData (27) + solveData (6), Solver.cpp:574-606, same order. */
typedef struct ps_stats {
    double dimData[27];
    double solveData[6];                /* error, iterations, solve CPU ms, solve wall ms, setup CPU ms, setup wall ms */
    int32_t result;                     /* ps_result */
    int32_t usedBiCGStab;               /* CG hit maxit and the fallback ran (Solver.cpp:784-799) */
    double stage_ms[16];                /* device time per stage, see PS_STAGE_* */
} ps_stats;

enum ps_stage {
    PS_STAGE_WEIGHTS = 0, PS_STAGE_CLASSIFY = 1, PS_STAGE_REGIONS = 2, PS_STAGE_INDICES = 3,
    PS_STAGE_TILE_MATRICES = 4, PS_STAGE_BLOCKS = 5, PS_STAGE_ASSEMBLE = 6, PS_STAGE_PRECOND = 7,
    PS_STAGE_SOLVE = 8, PS_STAGE_RECOVER = 9, PS_STAGE_WRITEBACK = 10, PS_STAGE_COUNT = 11
};

typedef struct ps_context ps_context;

/* Library/ABI version and a loud availability check (0 devices -> error string). */
int32_t ps_abi_version(void);
int32_t ps_reduced_dof(void);                     /* REDUCED_DOF of this build: 26 (quadratic) or 11 (affine) */

/* Context = what `Solver mySolver(...)` owns for one call (HDK_PolyStokes.C:333-343), kept alive
 * across steps so device buffers are reused.  One context per GPU. */
ps_context* ps_context_create(int32_t device);
void ps_context_destroy(ps_context* ctx);
const char* ps_last_error(const ps_context* ctx); /* replaces addError strings, HDK_PolyStokes.C:251-314 */

void ps_params_default(ps_params* p);             /* defaults of HDK_PolyStokes.C:88-208 */

/* Host -> device copy of the SIM fields (no reference equivalent: Houdini fields are host memory). */
int32_t ps_upload_fields(ps_context* ctx, const ps_params* p, const ps_fields_in* in);

/* The whole hot path on device-resident inputs: buildIntegrationWeightsAlt ... solve ...
 * recoverVelocityFromPressureStress, applySolutionToVelocity (HDK_PolyStokes.C:344-583).
 * Returns ps_result. */
int32_t ps_step_device(ps_context* ctx, ps_stats* stats);

/* Setup only (everything before solve(), HDK_PolyStokes.C:344-476); used by tests and exports. */
int32_t ps_setup_device(ps_context* ctx, ps_stats* stats);
/* solve() + recover + write-back on an already set-up context (HDK_PolyStokes.C:518-583). */
int32_t ps_solve_device(ps_context* ctx, ps_stats* stats);

/* UT_Interrupt equivalent (the reference polls boss->opInterrupt() in its sweeps, e.g. Classifier.cpp:73,386): the
 * callback is polled between batches of CG iterations; a non-zero return stops the solve, which then reports
 * PS_INCOMPLETE and leaves the velocity untouched.  Pass NULL to clear. */
typedef int32_t (*ps_interrupt_fn)(void* user);
int32_t ps_set_interrupt(ps_context* ctx, ps_interrupt_fn cb, void* user);

/* Device -> host copy of vel / valid. */
int32_t ps_download_fields(ps_context* ctx, ps_fields_out* out);

/* solveGasSubclass equivalent on host buffers: upload + step + download (HDK_PolyStokes.C:222-609). */
int32_t polystokes_step(ps_context* ctx, const ps_params* p, const ps_fields_in* in,
                        ps_fields_out* out, ps_stats* stats);

/* y = A x for host vectors of length nPressures+nStresses:
 * ApplyPressureStressMatrix::apply (lib/include/ApplyPressureStressMatrix.h:102-184). */
int32_t ps_apply_operator(ps_context* ctx, const double* x, double* y);

/* z = M^-1 r with the preconditioner of the set-up context (identity, Jacobi, Chebyshev), host vectors in reference numbering:
 * the parity hook for the preconditioner (tests compare it with the oracle's). */
int32_t ps_apply_preconditioner(ps_context* ctx, const double* r, double* z);

/* Inspection of solver state by name — the data behind printAllData()'s 43 point clouds
 * (Solver.cpp:1030-1074) and exportComponentMatrices() (Solver.cpp:543-566).
 * ps_query_array returns the element count (or <0 if unknown) and the element size in bytes;
 * ps_read_array copies it to host.
 * Beside the reference's arrays there are a few int32 diagnostics of the device path: "valuesCoded" (1: stencil values are
 * int8 codes), "columns16" (bit 0 / 1: S / St have the compressed 16-bit-column stream), "diagonalsCoded" (bit 0 / 1: uInv /
 * McInv are 1-byte value-set codes), "fusedStep" (1: the last PCG solve ran the four-kernel step), "streamRuns" (4 values:
 * entries of the distinct runs / all entries of the compressed stream of S, then of St). */
int64_t ps_query_array(ps_context* ctx, const char* name, int32_t* elem_bytes);
int32_t ps_read_array(ps_context* ctx, const char* name, void* dst, int64_t dst_bytes);

/* MatrixMarket export with the reference's file names and text format
 * (Solver.cpp:533-606; extern/eigen/unsupported/Eigen/src/SparseExtra/MarketIO.h:310-380). */
int32_t ps_export_component_matrices(ps_context* ctx, const char* prefix);
/* exportMatrices + exportMatricesPostSolve (Solver.cpp:533-572): <prefix>Mat_A.mtx (n x n, empty: the live factored path
 * never assembles A), Vec_b.mtx, Vec_guess.mtx (zero) and, after a solve, solutionVector.mtx. */
int32_t ps_export_matrices(ps_context* ctx, const char* prefix);
int32_t ps_export_stats(ps_context* ctx, const ps_stats* stats, const char* prefix);

/* Solve a component set exported by exportComponentMatrices() (Solver.cpp:543-566: <prefix>Mat_G.mtx, Mat_Dt, Mat_JG,
 * Mat_JDt, Mat_McInv, Mat_uInv, Mat_Inv_Mr_plus_2JDtuDJ, Vec_b) with the same PCG (params: tolerance,
 * maxSolverIterations, preconditioner); the operator is applied literally as in ApplyPressureStressMatrix.h:102-179.
 * x_out receives [p; tau] (length nPressures + nStresses, reference numbering).  `dt` is dimData entry 27. */
int32_t ps_solve_exported_system(ps_context* ctx, const char* prefix, const ps_params* params, double dt, double* x_out,
                                 int64_t x_len, ps_stats* stats);

/* Micro-benchmark hooks used by bench.py for the roofline object: run `iters` launches of the
 * dominant kernel(s) on the solver stream bracketed by HIP events, return avg ms per launch. */
int32_t ps_bench_kernel(ps_context* ctx, const char* kernel, int32_t iters, double* avg_ms,
                        double* algorithmic_bytes);

/* Device memory held through this library, for a host application's bookkeeping and for the tests that check a context does not grow
 * across steps: out4 = { bytes allocated by all contexts of this process (buffers waiting for release included), their peak,
 * bytes of buffers THIS context dropped that still wait for release (0 after every successful or failed setup / solve / step: they are
 * released where the context's stream has just been synchronised), live contexts }.  ctx may be null (entry 2 is then 0).
 * Replaces nothing in the reference (its Solver owns host temporaries for the duration of the call, exec/HDK_PolyStokesSolver.h:272-375). */
int32_t ps_memory_stats(const ps_context* ctx, int64_t* out4);

/* ---- Multi-GPU (not in the reference, which is single-process) -------------------------
 * Slab decomposition along z, cut at multiples of lcm(16, tileSize).  Each rank is given (as an ordinary
 * ps_fields_in) its slab plus one halo tile per interior side, and is told which local cell layers it owns.
 * Per operator apply the ranks exchange the one-cell layer of x their rows touch across each cut and the y
 * contributions their rows make to the neighbour's layer; CG scalars are all-reduced. */
typedef struct ps_slab {
    int32_t rank, world;
    int32_t zLoOwned, zHiOwned;     /* owned cell layers [zLo, zHi) in LOCAL grid coordinates (multiples of 16) */
    int32_t hasLower, hasUpper;     /* a neighbouring rank exists below / above */
    int32_t zGlobalOwned;           /* GLOBAL index of the cell layer zLoOwned: tile offsets are formed with global z, so a tile's
                                     * matrices and fit do not depend on the decomposition */
} ps_slab;
int32_t ps_set_slab(ps_context* ctx, const ps_slab* slab);            /* after ps_upload_fields, before setup */
/* The same along all three axes (bricks whose faces lie on tile-size multiples; 8 GPUs as 2 x 2 x 2): the rank's local
 * grid is its owned box plus one 16-cell halo block on every side that has a neighbour.  rank = c0 + dims[0] * (c1 + dims[1] * c2)
 * for the brick at (c0, c1, c2); the neighbour below / above along axis a is rank -+ the stride of a.  A plane on a cut belongs to the
 * rank above it (faces and edges with offset 0 along that axis).  Every rank exchanges with its <= 6 face neighbours only: the
 * DOFs a row touches across two cuts at once belong to rows of another rank.  ps_set_slab is ps_set_brick with dims = {1, 1, world}. */
typedef struct ps_brick {
    int32_t rank, world;
    int32_t dims[3];                /* ranks per axis */
    int32_t lo[3], hi[3];           /* owned cells [lo, hi) per axis in LOCAL grid coordinates (multiples of 16 and of the tile size) */
    int32_t hasLower[3], hasUpper[3];
    int32_t globalLo[3];            /* GLOBAL index of the cell lo: tile offsets are formed with global indices */
} ps_brick;
int32_t ps_set_brick(ps_context* ctx, const ps_brick* brick);         /* after ps_upload_fields, before setup */

/* One process per GPU: RCCL communicator on the solver stream.  Rank 0 calls ps_comm_unique_id and hands the
 * 128 bytes to the other ranks (bench.py broadcasts them over gloo); every rank then calls ps_comm_init_rccl.
 * ps_step_device then runs the distributed step; ps_setup_device / ps_solve_device / polystokes_step refuse a
 * context with a slab (a rank's local system is only a fragment).  The communicator is destroyed with the context.
 * An interrupt callback (ps_set_interrupt) on ANY rank stops all ranks at the same CG batch (PS_INCOMPLETE); a rank
 * whose setup fails makes every rank return PS_FAILED instead of leaving its neighbours waiting. */
int32_t ps_comm_unique_id(void* id128);
int32_t ps_comm_init_rccl(ps_context* ctx, const void* id128, int32_t rank, int32_t world);
int32_t ps_comm_selftest(ps_context* ctx);   /* collective: all-reduce, grouped send/recv to self, and (slab set, world > 1) one ring
                                              * step with the real neighbours and a check of the all-reduced sums */
/* What the last distributed solve of this rank did (no reference counterpart: the reference is single-process, its only parallelism
 * is lib/include/ApplyPressureStressMatrix.h:122-164): out8 = { bytes sent per CG iteration over the rank's cuts, owned DOFs,
 * 1 if the halo exchanges overlapped with the interior rows, sum [ms] and count of sampled x-exchange transports, sum [ms] and
 * count of sampled scalar all-reduces (incl. synchronisation), cells of the rank's halo blocks whose label was replaced by the
 * owner's during setup (the classification reaches beyond a halo block: boundary layers, fixReducedRegionBoundaries) }. */
int32_t ps_dist_stats(ps_context* ctx, double* out8);
/* Host-staged transport instead of RCCL (pack -> D2H -> TCP -> H2D -> unpack; scalar all-reduce through rank 0):
 * one process per rank, several ranks may share one GPU (RCCL refuses duplicate devices) — the route by which the
 * real multi-process path runs on a single-GPU box, and the fallback where librccl is missing.  Rank r listens on
 * base_port + r of `host` (dotted IPv4).  Collective over the `world` ranks. */
int32_t ps_comm_init_tcp(ps_context* ctx, int32_t rank, int32_t world, const char* host, int32_t base_port);

/* Several ranks inside ONE process on one GPU (device-to-device copies instead of RCCL): used to test the
 * distributed algorithm on a single-GPU box.  Same kernels, same exchange lists, same reduction order. */
typedef struct ps_group ps_group;
ps_group* ps_group_create(int32_t device, int32_t world);
void ps_group_destroy(ps_group* g);
ps_context* ps_group_rank(ps_group* g, int32_t rank);                 /* upload fields / set slab per rank */
int32_t ps_group_step(ps_group* g, ps_stats* stats);                  /* setup on every rank + distributed solve */

#ifdef __cplusplus
}
#endif
#endif /* POLYSTOKES_H */
