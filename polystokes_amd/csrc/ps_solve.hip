// The per-iteration hot loop: y = A x for the factored pressure-stress operator and the PCG around it.
//
//   A = -dt [G Dt]^T McInv [G Dt] - [JG JDt]^T BInv [JG JDt] - 1/2 diag(0, uInv)
//       (lib/include/ApplyPressureStressMatrix.h:102-179; explicit form exec/..._AssembleSystem.cpp:381-389)
// evaluated as  s = S x;  t_f = dt McInv_f s_f (active rows),  t_f = C_f . (BInv_r sum_{g in r} C_g s_g)
// (reduced rows, J evaluated on the fly);  y = -S^T t - 1/2 uInv x_tau.
//
// One translation unit, split over included parts:
//   ps_kernels_spmv.hpp  : k_spmv_S / k_spmv_St (CSR-stream one-shot kernels, fp64 or int8-coded values) and
//                          k_spmv_S_pipe / k_spmv_St_pipe (persistent, software-pipelined, compressed 3 B/nnz stream) —
//                          a 256-thread block owns 256 consecutive rows, products go through LDS, each thread sums its
//                          own short row; epilogues fuse the diagonal scalings, the -1/2 uInv x term and the p.Ap partials.
//   ps_kernels_tiles.hpp : k_tile_gather / k_tile_solve / k_tile_expand — per-tile J^T, 26x26 BInv, J.
//   ps_kernels_cg.hpp    : k_cg_update_r / k_cg_update_xp (the PCG step of pcg_external_matrix_A, lib/include/pcg.h:268-340,
//                          with the scalar reductions and the stop rule folded in; scalars stay on the device), the legacy
//                          k_cg_update_xr / _p / scal* used by the exported-system path, BiCGStab helpers, Jacobi diagonal,
//                          velocity recovery / write-back.
//   this file            : the launch dispatch (Launch), ps_context::applyOperator / assemble / solve / recover.
//   ps_dist.hpp          : the z-slab distributed solve (RCCL or in-process ranks) and its C ABI.
//   ps_import.hpp        : MatrixMarket import + general CSR PCG (ps_solve_exported_system).
#include <chrono>
#include <cstring>
#include <cmath>
#include <limits>
#include <ctime>

#include "ps_context.hpp"

using namespace ps;

namespace {
#include "ps_kernels_spmv.hpp"
#include "ps_kernels_tiles.hpp"
#include "ps_kernels_cg.hpp"
}  // namespace

// ---------------------------------------------------------------------------------------------------
namespace {
struct Launch {
    ps_context* c;
    const int* done;
    int rowsS, rowsSt, nA, nP;
    void spmvS_(int mode, const double* x, double* out) const {
        const dim3 gr(gridFor(rowsS, BS)), bl(BS);
        const ps::DevCSR& M = c->S;
#define PS_LAUNCH_S(MODE_, PK_) hipLaunchKernelGGL((k_spmv_S<MODE_, 8, PK_>), gr, bl, 0, c->stream, M.ptr.p, M.col.p, M.val.p, M.code.p, \
                                                   c->valScale, x, rowsS, nA, c->dt, c->McInv.p, out, done)
        if (mode == 0) { if (M.packed) PS_LAUNCH_S(0, true); else PS_LAUNCH_S(0, false); }
        else { if (M.packed) PS_LAUNCH_S(1, true); else PS_LAUNCH_S(1, false); }
#undef PS_LAUNCH_S
    }
    // fused residual update (solve(): FusedR): where the S and tile kernels leave their shares of p.Ap (null: not asked for)
    double* sPart = nullptr;
    double* wvPart = nullptr;
    // chunk lists (row-per-lane kernels only; null: every chunk): ps_dist.hpp launches the chunks next to a cut and the others separately
    const int32_t* sList = nullptr; int nSList = 0;
    const int32_t* stList = nullptr; int nStList = 0;
    bool stOwnedOnly = false;        // the St chunk list holds owned rows only (the decomposition's launch under the exchange): FX bit 2
    bool listsOk() const { return pipeGrid > 0 && c->S.col16ok && c->S.packed && c->S.ellok && c->St.col16ok && c->St.packed && c->St.ellok; }
    // workgroups of a launch over n chunks of S / St (the partial sums it writes)
    int sBlocksFor(int n) const { int xcd = xcdAware; return n > 0 ? pipeBlocks(n, xcd, true, sCap()) : 0; }
    int stBlocksFor(int n, int mode) const { int xcd = xcdAware; return n > 0 ? pipeBlocks(n, xcd, true, stGridFor(mode)) : 0; }
    bool ntSpmv = true;   // cache policy of the pipelined kernels' streams (ps_context::ntLevel >= 1)
    // POL of the pipelined kernels (ps_kernels_spmv.hpp): 0 = default policy; 1 = non-temporal stores and epilogue streams, cached matrix
    // stream (most runs shared between chunks: read again and again); 3 = the matrix stream non-temporal too (every run read once)
    int policy(const ps::DevCSR& M) const { return !ntSpmv ? 0 : (2 * M.uniqueLen <= M.streamLen ? 1 : 3); }
    int pipeGrid;   // 0: one-shot kernels; >0: persistent software-pipelined kernels with this many blocks
    int stGrid = 0; // > 0: the St kernel's own cap
    int xcdAware;   // pipelined kernels: runs of this many chunks are dealt to the XCDs round robin (ChunkWalk); 0 = plain walk
    void spmvS(int mode, const double* x, double* out) const {
        if (rowsS == 0) return;
        const ps::DevCSR& M = c->S;
        if (pipeGrid > 0 && M.col16ok && M.packed && M.ellok) {   // row-per-lane kernels on the coded stream (ps_kernels_spmv.hpp: k_spmv_S_ell)
            const int nChunks = sList ? nSList : c->S.nChunks;     // (a chunk list: the slab decomposition's interior / boundary launches)
            if (nChunks == 0) return;
            int xcdAware = this->xcdAware;
            const dim3 gr(pipeBlocks(nChunks, xcdAware, true, sCap())), bl(BS);
            // two units in flight per wave (k_spmv_S_ell2; r04): 256^3, same box, two interleaved rounds: S 0.2809 / 0.2816 -> 0.2672 / 0.2609 ms in
            // sequence, step 1128.6 / 1130.2 -> 1117.3 / 1118.5 ms (profiles/r04_s_dual.txt).  PS_S_DUAL=0: the one-unit kernel.
            static const bool dual = !(PS_ENV("PS_S_DUAL") && atoi(PS_ENV("PS_S_DUAL")) == 0);
            if (dual && mode == 0 && c->mcCoded && (gr.x & 7) == 0) {
                const int pol = policy(M);
#define PS_LAUNCH_S2L(POL_, LIST_) hipLaunchKernelGGL((k_spmv_S_ell2<POL_, LIST_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                              M.echunk.p, c->valScale, x, (int)M.cols, rowsS, nA, c->dt, out, done, nChunks, (const uint8_t*)c->mcCode.p, c->mcDict.p, sPart, sList)
#define PS_LAUNCH_S2(POL_) do { if (sList) PS_LAUNCH_S2L(POL_, true); else PS_LAUNCH_S2L(POL_, false); } while (0)
                if (pol == 3) PS_LAUNCH_S2(3); else if (pol == 1) PS_LAUNCH_S2(1); else PS_LAUNCH_S2(0);
#undef PS_LAUNCH_S2
#undef PS_LAUNCH_S2L
                return;
            }
#define PS_LAUNCH_SE(MODE_, POL_) do { if (sList) PS_LAUNCH_SEL(MODE_, POL_, true); else PS_LAUNCH_SEL(MODE_, POL_, false); } while (0)
#define PS_LAUNCH_SEL(MODE_, POL_, LIST_) hipLaunchKernelGGL((k_spmv_S_ell<MODE_, POL_, LIST_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                                     M.echunk.p, c->valScale, x, (int)M.cols, rowsS, nA, c->dt, c->McInv.p, out, done, nChunks, xcdAware, c->mcCoded ? c->mcCode.p : (const uint8_t*)nullptr, c->mcDict.p, sPart, sList)
#define PS_LAUNCH_SE2(MODE_) do { const int pol = policy(M); if (pol == 3) PS_LAUNCH_SE(MODE_, 3); else if (pol == 1) PS_LAUNCH_SE(MODE_, 1); else PS_LAUNCH_SE(MODE_, 0); } while (0)
            if (mode == 0) PS_LAUNCH_SE2(0); else PS_LAUNCH_SE2(1);
#undef PS_LAUNCH_SE2
#undef PS_LAUNCH_SEL
#undef PS_LAUNCH_SE
            return;
        }
        if (pipeGrid > 0 && M.col16ok && (M.packed || M.val4.p)) {
            const int nChunks = c->S.nChunks;
            int xcdAware = this->xcdAware;
            const dim3 gr(pipeBlocks(nChunks, xcdAware, M.packed)), bl(BS);
#define PS_LAUNCH_SP(MODE_, NV_, F64_, POL_) hipLaunchKernelGGL((k_spmv_S_pipe<MODE_, NV_, F64_, POL_>), gr, bl, 0, c->stream, M.col16.p, M.code4.p, M.val4.p, (int)M.streamLen, M.winBase.p, \
                                                    M.chunkInfo.p, M.len8.p, c->valScale, x, (int)M.cols, rowsS, nA, c->dt, c->McInv.p, out, done, nChunks, xcdAware, c->mcCoded ? c->mcCode.p : (const uint8_t*)nullptr, c->mcDict.p, sPart)
#define PS_LAUNCH_SP2(MODE_, NV_) do { const int pol = policy(M); if (!M.packed) PS_LAUNCH_SP(MODE_, NV_, true, 3); else if (pol == 3) PS_LAUNCH_SP(MODE_, NV_, false, 3); \
                                       else if (pol == 1) PS_LAUNCH_SP(MODE_, NV_, false, 1); else PS_LAUNCH_SP(MODE_, NV_, false, 0); } while (0)
            if (M.nv == 1) { if (mode == 0) PS_LAUNCH_SP2(0, 1); else PS_LAUNCH_SP2(1, 1); }
            else { if (mode == 0) PS_LAUNCH_SP2(0, 2); else PS_LAUNCH_SP2(1, 2); }
#undef PS_LAUNCH_SP2
#undef PS_LAUNCH_SP
            return;
        }
        spmvS_(mode, x, out);
    }
    void tiles(int mode, double* ts) const {   // ts: face-row vector; reduced part rewritten in place
        if (c->regionCount == 0) return;
        double* sred = ts + nA;
        const dim3 gr((unsigned)c->regionCount);
        static const bool noFuse = PS_ENV("PS_TILE_SPLIT") && atoi(PS_ENV("PS_TILE_SPLIT")) != 0;   // A/B: force the three-kernel form
        if (c->maxRegionRows <= TILE_FUSED_MAX_ROWS && !noFuse) {   // one workgroup per region: gather, 26x26 block, expand
#define PS_TILE_APPLY(MODE_, TB_) hipLaunchKernelGGL((k_tile_apply<MODE_, TB_>), gr, dim3(TB_), 0, c->stream, c->regionRowPtr.p, c->rrowFace.p, c->COM.p, c->dx, make_int3(c->gOff[0], c->gOff[1], c->gOff[2]), c->Binv.p, \
                                                c->rhsR.p, c->invDt, sred, c->vreg.p, done, wvPart)
            // threads per region: enough threads in flight chip-wide (~256 K) without starving a region of work.  Measured
            // at 256^3 (4096 tiles of 3204 rows): 0.080 ms with 64 threads, 0.087 / 0.106 / 0.166 with 128 / 256 / 512; at 32^3
            // (8 tiles) one wavefront per tile serialises 50 rows per lane behind memory latency (60 us per CG iteration
            // against 43 with 256 threads per tile; 1024 threads: 50, the block reduction over 16 waves costs more than it hides).
            static const int tbEnv = PS_ENV("PS_TILE_TB") ? atoi(PS_ENV("PS_TILE_TB")) : 0;
            int tb = tbEnv;
            if (!tb) {
                tb = 64;
                while (tb < 256 && (int64_t)tb * c->regionCount < 262144 && (int64_t)tb * 2 < c->maxRegionRows) tb *= 2;
            }
#define PS_TILE_APPLY_TB(MODE_) do { if (tb >= 1024) PS_TILE_APPLY(MODE_, 1024); else if (tb >= 512) PS_TILE_APPLY(MODE_, 512); else if (tb >= 256) PS_TILE_APPLY(MODE_, 256); \
                                     else if (tb >= 128) PS_TILE_APPLY(MODE_, 128); else PS_TILE_APPLY(MODE_, 64); } while (0)
            if (mode == 0) PS_TILE_APPLY_TB(0); else if (mode == 1) PS_TILE_APPLY_TB(1); else PS_TILE_APPLY_TB(2);
#undef PS_TILE_APPLY_TB
#undef PS_TILE_APPLY
            return;
        }
        if (mode != 2 && c->nRChunks > 0)
            hipLaunchKernelGGL(k_tile_gather, dim3((unsigned)c->nRChunks), dim3(64), 0, c->stream, c->rchunkRegion.p, c->rchunkStart.p,
                               c->rchunkEnd.p, c->rrowFace.p, c->COM.p, c->dx, make_int3(c->gOff[0], c->gOff[1], c->gOff[2]), sred, c->wreg.p, done);
        const dim3 bl(64);
        if (mode == 0)
            hipLaunchKernelGGL(k_tile_solve<0>, gr, bl, 0, c->stream, c->regionChunkPtr.p, c->wreg.p, c->Binv.p, c->rhsR.p, c->invDt, c->vreg.p, done, wvPart);
        else if (mode == 1)
            hipLaunchKernelGGL(k_tile_solve<1>, gr, bl, 0, c->stream, c->regionChunkPtr.p, c->wreg.p, c->Binv.p, c->rhsR.p, c->invDt, c->vreg.p, done, (double*)nullptr);
        else
            hipLaunchKernelGGL(k_tile_solve<2>, gr, bl, 0, c->stream, c->regionChunkPtr.p, c->wreg.p, c->Binv.p, c->rhsR.p, c->invDt, c->vreg.p, done, (double*)nullptr);
        if (mode != 1 && c->nRChunks > 0)
            hipLaunchKernelGGL(k_tile_expand, dim3((unsigned)c->nRChunks), dim3(BS), 0, c->stream, c->rchunkRegion.p, c->rchunkStart.p,
                               c->rchunkEnd.p, c->rrowFace.p, c->COM.p, c->dx, make_int3(c->gOff[0], c->gOff[1], c->gOff[2]), c->vreg.p, sred, done);
    }
    // ---- the inner operator applies of the single-precision Chebyshev polynomial (PS_PRE_CHEBYSHEV_F32): z_j and the face-row vector are stored
    // as fp32 (ps_kernels_spmv.hpp: VecIO), on the two-units-per-wave kernels only — cheb32Ok() says whether this system runs them
    bool cheb32Ok() const {
        static const bool dualS = !(PS_ENV("PS_S_DUAL") && atoi(PS_ENV("PS_S_DUAL")) == 0), dualT = !(PS_ENV("PS_ST_DUAL") && atoi(PS_ENV("PS_ST_DUAL")) == 0);
        static const bool noFuse = PS_ENV("PS_TILE_SPLIT") && atoi(PS_ENV("PS_TILE_SPLIT")) != 0;
        return dualS && dualT && listsOk() && c->mcCoded && !c->slabEnabled && !sList && !stList && xcdAware > 0 && c->S.nChunks >= 8 && c->St.nChunks >= 8 &&
               rowsS > 0 && rowsSt > 0 && (c->regionCount == 0 || (c->maxRegionRows <= TILE_FUSED_MAX_ROWS && !noFuse));
    }
    void spmvS32(const float* x, float* out) const {
        const ps::DevCSR& M = c->S;
        int xcd = xcdAware;
        const dim3 gr(pipeBlocks(M.nChunks, xcd, true, sCap())), bl(BS);
        const int pol = policy(M);
#define PS_LAUNCH_S2F(POL_) hipLaunchKernelGGL((k_spmv_S_ell2<POL_, false, float>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                               M.echunk.p, c->valScale, x, (int)M.cols, rowsS, nA, c->dt, out, done, M.nChunks, (const uint8_t*)c->mcCode.p, c->mcDict.p, (double*)nullptr, (const int32_t*)nullptr)
        if (pol == 3) PS_LAUNCH_S2F(3); else if (pol == 1) PS_LAUNCH_S2F(1); else PS_LAUNCH_S2F(0);
#undef PS_LAUNCH_S2F
    }
    void tiles32(float* ts) const {
        if (c->regionCount == 0) return;
        float* sred = ts + nA;
        const dim3 gr((unsigned)c->regionCount);
        int tb = 64;
        while (tb < 256 && (int64_t)tb * c->regionCount < 262144 && (int64_t)tb * 2 < c->maxRegionRows) tb *= 2;
#define PS_TILE_APPLY_F(TB_) hipLaunchKernelGGL((k_tile_apply<0, TB_, float>), gr, dim3(TB_), 0, c->stream, c->regionRowPtr.p, c->rrowFace.p, c->COM.p, c->dx, make_int3(c->gOff[0], c->gOff[1], c->gOff[2]), c->Binv.p, \
                                                c->rhsR.p, c->invDt, sred, c->vreg.p, done, (double*)nullptr)
        if (tb >= 256) PS_TILE_APPLY_F(256); else if (tb >= 128) PS_TILE_APPLY_F(128); else PS_TILE_APPLY_F(64);
#undef PS_TILE_APPLY_F
    }
    int spmvSt2c32(const float* t, const float* xin, float* out, double* partial, const ChebArgs& ca) const {   // returns the number of partials written
        const ps::DevCSR& M = c->St;
        int xcd = xcdAware;
        const dim3 gr(pipeBlocks(M.nChunks, xcd, true, stGridFor(2))), bl(BS);
        const int pol = policy(M);
        const uint8_t* uArg = c->uCoded ? (const uint8_t*)c->uCode.p : (const uint8_t*)c->uInv.p;
#define PS_LAUNCH_T2CF(POL_, UC_) hipLaunchKernelGGL((k_spmv_St_ell2c<POL_, float, UC_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                                M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, xin, out, partial, done, M.nChunks, ca, uArg, c->uDict.p)
        if (c->uCoded) { if (pol == 3) PS_LAUNCH_T2CF(3, true); else if (pol == 1) PS_LAUNCH_T2CF(1, true); else PS_LAUNCH_T2CF(0, true); }
        else { if (pol == 3) PS_LAUNCH_T2CF(3, false); else if (pol == 1) PS_LAUNCH_T2CF(1, false); else PS_LAUNCH_T2CF(0, false); }
#undef PS_LAUNCH_T2CF
        return (int)gr.x;
    }
    bool cz32 = false;   // MODE 3 with the polynomial's first term: fr.cz points at floats (k_spmv_St_ell2<.., float>)
    void spmvSt_(int mode, const double* t, const double* xin, const double* add, double* out, double* partial) const {
        const dim3 gr(gridFor(rowsSt, BS)), bl(BS);
        const ps::DevCSR& M = c->St;
#define PS_LAUNCH_T(MODE_, PK_) hipLaunchKernelGGL((k_spmv_St<MODE_, 6, PK_>), gr, bl, 0, c->stream, M.ptr.p, M.col.p, M.val.p, M.code.p, \
                                                   c->valScale, t, rowsSt, nP, c->uInv.p, xin, add, out, partial, done)
        if (mode == 0) { if (M.packed) PS_LAUNCH_T(0, true); else PS_LAUNCH_T(0, false); }
        else { if (M.packed) PS_LAUNCH_T(1, true); else PS_LAUNCH_T(1, false); }
#undef PS_LAUNCH_T
    }
    bool stOnPipe() const { return pipeGrid > 0 && c->St.col16ok && (c->St.packed || c->St.val4.p); }
    // mode 2 (Chebyshev term fused into the epilogue, ChebArgs) exists on the pipelined kernels only: callers check stOnPipe()
    // mode 3 (residual update fused into the epilogue, FusedR): pipelined kernels on the coded stream only — callers check fusedOk()
    void spmvSt(int mode, const double* t, const double* xin, const double* add, double* out, double* partial, const ChebArgs* cheb = nullptr,
                const FusedR* fused = nullptr) const {
        if (rowsSt == 0) return;
        const ps::DevCSR& M = c->St;
        ChebArgs ca{nullptr, nullptr, nullptr, 0., 0.};
        if (cheb) ca = *cheb;
        FusedR fr{};
        if (fused) fr = *fused;
        if (pipeGrid > 0 && M.col16ok && M.packed && M.ellok) {   // row-per-lane kernels on the coded stream (k_spmv_St_ell)
            const int nChunks = stList ? nStList : c->St.nChunks;
            if (nChunks == 0) return;
            int xcdAware = this->xcdAware;
            const dim3 gr(pipeBlocks(nChunks, xcdAware, true, stGridFor(mode))), bl(BS);
#define PS_LAUNCH_TE(MODE_, POL_) PS_LAUNCH_TEX(MODE_, POL_, 0)
#define PS_LAUNCH_TEX(MODE_, POL_, FX_) do { if (stList) PS_LAUNCH_TEL(MODE_, POL_, FX_, true); else PS_LAUNCH_TEL(MODE_, POL_, FX_, false); } while (0)
#define PS_LAUNCH_TEL(MODE_, POL_, FX_, LIST_) hipLaunchKernelGGL((k_spmv_St_ell<MODE_, POL_, FX_, LIST_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                                     M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, c->uInv.p, xin, add, out, partial, done, nChunks, xcdAware, ca, c->uCoded ? c->uCode.p : (const uint8_t*)nullptr, c->uDict.p, fr, stList)
#define PS_LAUNCH_TE2(MODE_) do { const int pol = policy(M); if (pol == 3) PS_LAUNCH_TE(MODE_, 3); else if (pol == 1) PS_LAUNCH_TE(MODE_, 1); else PS_LAUNCH_TE(MODE_, 0); } while (0)
            // MODE 3 specialisations (k_spmv_St_ell: FX): coded uInv without the Chebyshev term, in a single domain (3) or on a slab rank (1)
            // (r06: the two-unit kernels also take the stress diagonal as the fp64 array — UC = false, a viscosity field with more than 256 values —; the one-unit
            // FX forms stay coded-only)
            const bool codedU = c->uCoded;
            const uint8_t* uArg = codedU ? (const uint8_t*)c->uCode.p : (const uint8_t*)c->uInv.p;
            const bool coded3 = mode == 3 && codedU && !fr.cz, single3 = mode == 3 && !fr.cz && plain3Hint && !fr.yOut && !fr.red && fr.rStride == 0;
            // two units in flight per wave (k_spmv_St_ell2; r04): 256^3, one box, interleaved rounds: St with the residual update 0.4251 / 0.4242 ->
            // 0.3923 / 0.3918 ms in sequence at 5 waves per SIMD on 1280 workgroups, step 1124.5 / 1122.6 -> 1098.3 / 1096.6 ms; compiled for 6 waves per
            // SIMD on 1536 workgroups another 0.6 % (profiles/r04_st_dual.txt).  PS_ST_DUAL=0: the one-unit kernel on 1792 workgroups.
            static const bool dualC = !(PS_ENV("PS_ST_DUAL") && atoi(PS_ENV("PS_ST_DUAL")) == 0);
            if (mode == 2 && dualC && !stList && (gr.x & 7) == 0 && !c->slabEnabled) {   // a Chebyshev term, two units in flight per wave
                const int pol = policy(M);
#define PS_LAUNCH_T2C(POL_, UC_) hipLaunchKernelGGL((k_spmv_St_ell2c<POL_, double, UC_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                               M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, xin, out, partial, done, nChunks, ca, uArg, c->uDict.p)
                if (codedU) { if (pol == 3) PS_LAUNCH_T2C(3, true); else if (pol == 1) PS_LAUNCH_T2C(1, true); else PS_LAUNCH_T2C(0, true); }
                else { if (pol == 3) PS_LAUNCH_T2C(3, false); else if (pol == 1) PS_LAUNCH_T2C(1, false); else PS_LAUNCH_T2C(0, false); }
#undef PS_LAUNCH_T2C
                return;
            }
            if (mode == 3 && dualC && plain3Hint2 && fr.cz && !fr.dinvF && !fr.yOut && !fr.red && fr.rStride == 0 && !stList && (gr.x & 7) == 0) {
                const int pol = policy(M);   // the Chebyshev step's St launch (first term of the polynomial in the epilogue), two units in flight per wave
#define PS_LAUNCH_T2Z(POL_, TZ_, UC_) hipLaunchKernelGGL((k_spmv_St_ell2<POL_, true, false, false, TZ_, false, UC_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                               M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, xin, done, nChunks, uArg, c->uDict.p, fr, (const int32_t*)nullptr)
#define PS_LAUNCH_T2Z2(POL_, TZ_) do { if (codedU) PS_LAUNCH_T2Z(POL_, TZ_, true); else PS_LAUNCH_T2Z(POL_, TZ_, false); } while (0)
                if (cz32) { if (pol == 3) PS_LAUNCH_T2Z2(3, float); else if (pol == 1) PS_LAUNCH_T2Z2(1, float); else PS_LAUNCH_T2Z2(0, float); }
                else { if (pol == 3) PS_LAUNCH_T2Z2(3, double); else if (pol == 1) PS_LAUNCH_T2Z2(1, double); else PS_LAUNCH_T2Z2(0, double); }
#undef PS_LAUNCH_T2Z2
#undef PS_LAUNCH_T2Z
                return;
            }
            if (cz32 && mode == 3 && fr.cz) throw Error("internal: single-precision Chebyshev vectors without the two-unit St kernel");
            if (single3 && stDual() && !stList && (gr.x & 7) == 0) {
                const int pol = policy(M);
#define PS_LAUNCH_T2(POL_, UC_) hipLaunchKernelGGL((k_spmv_St_ell2<POL_, false, false, false, double, false, UC_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                              M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, xin, done, nChunks, uArg, c->uDict.p, fr, (const int32_t*)nullptr)
                if (codedU) { if (pol == 3) PS_LAUNCH_T2(3, true); else if (pol == 1) PS_LAUNCH_T2(1, true); else PS_LAUNCH_T2(0, true); }
                else { if (pol == 3) PS_LAUNCH_T2(3, false); else if (pol == 1) PS_LAUNCH_T2(1, false); else PS_LAUNCH_T2(0, false); }
#undef PS_LAUNCH_T2
                return;
            }
            if (single3 && codedU) { const int pol = policy(M); if (pol == 3) PS_LAUNCH_TEX(3, 3, 3); else if (pol == 1) PS_LAUNCH_TEX(3, 1, 3); else PS_LAUNCH_TEX(3, 0, 3); }
            else if (mode == 3 && !fr.cz && stOwnedOnly && stList && dualC && fr.red && (gr.x & 7) == 0) {   // a rank's launch over owned rows only, two units in flight per wave
                const int pol = policy(M);
#define PS_LAUNCH_T2D(POL_, UC_) hipLaunchKernelGGL((k_spmv_St_ell2<POL_, false, true, true, double, false, UC_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                               M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, xin, done, nChunks, uArg, c->uDict.p, fr, stList)
                if (codedU) { if (pol == 3) PS_LAUNCH_T2D(3, true); else if (pol == 1) PS_LAUNCH_T2D(1, true); else PS_LAUNCH_T2D(0, true); }
                else { if (pol == 3) PS_LAUNCH_T2D(3, false); else if (pol == 1) PS_LAUNCH_T2D(1, false); else PS_LAUNCH_T2D(0, false); }
#undef PS_LAUNCH_T2D
            }
            else if (mode == 3 && !fr.cz && dualC && fr.red && fr.yOut && !stOwnedOnly && (gr.x & 7) == 0) {   // a rank's launch that holds halo rows (the chunks next to a cut, or the whole rank), two units in flight per wave
                const int pol = policy(M);
#define PS_LAUNCH_T2H(POL_, LIST_, UC_) hipLaunchKernelGGL((k_spmv_St_ell2<POL_, false, true, LIST_, double, true, UC_>), gr, bl, 0, c->stream, M.ecol.p, M.ecode.p, (unsigned)(M.ellCols * 2), (unsigned)M.ellCodes, M.winBase.p, \
                                               M.echunk.p, c->valScale, t, (int)M.cols, rowsSt, xin, done, nChunks, uArg, c->uDict.p, fr, stList)
#define PS_LAUNCH_T2H2(POL_, UC_) do { if (stList) PS_LAUNCH_T2H(POL_, true, UC_); else PS_LAUNCH_T2H(POL_, false, UC_); } while (0)
                if (codedU) { if (pol == 3) PS_LAUNCH_T2H2(3, true); else if (pol == 1) PS_LAUNCH_T2H2(1, true); else PS_LAUNCH_T2H2(0, true); }
                else { if (pol == 3) PS_LAUNCH_T2H2(3, false); else if (pol == 1) PS_LAUNCH_T2H2(1, false); else PS_LAUNCH_T2H2(0, false); }
#undef PS_LAUNCH_T2H2
#undef PS_LAUNCH_T2H
            }
            else if (coded3 && stOwnedOnly && stList) { const int pol = policy(M); if (pol == 3) PS_LAUNCH_TEL(3, 3, 5, true); else if (pol == 1) PS_LAUNCH_TEL(3, 1, 5, true); else PS_LAUNCH_TEL(3, 0, 5, true); }
            else if (coded3) { const int pol = policy(M); if (pol == 3) PS_LAUNCH_TEX(3, 3, 1); else if (pol == 1) PS_LAUNCH_TEX(3, 1, 1); else PS_LAUNCH_TEX(3, 0, 1); }
            else if (mode == 0) PS_LAUNCH_TE2(0); else if (mode == 1) PS_LAUNCH_TE2(1); else if (mode == 2) PS_LAUNCH_TE2(2); else PS_LAUNCH_TE2(3);
#undef PS_LAUNCH_TE2
#undef PS_LAUNCH_TEL
#undef PS_LAUNCH_TEX
#undef PS_LAUNCH_TE
            return;
        }
        if (pipeGrid > 0 && M.col16ok && (M.packed || M.val4.p)) {
            const int nChunks = c->St.nChunks;
            int xcdAware = this->xcdAware;
            const dim3 gr(pipeBlocks(nChunks, xcdAware, M.packed, stGridFor(mode))), bl(BS);
#define PS_LAUNCH_TP(MODE_, NV_, F64_, POL_) hipLaunchKernelGGL((k_spmv_St_pipe<MODE_, NV_, F64_, POL_>), gr, bl, 0, c->stream, M.col16.p, M.code4.p, M.val4.p, (int)M.streamLen, M.winBase.p, \
                                                    M.chunkInfo.p, M.len8.p, c->valScale, t, (int)M.cols, rowsSt, c->uInv.p, xin, add, out, partial, done, nChunks, xcdAware, ca, c->uCoded ? c->uCode.p : (const uint8_t*)nullptr, c->uDict.p, fr)
#define PS_LAUNCH_TP2(MODE_, NV_) do { const int pol = policy(M); if (!M.packed) PS_LAUNCH_TP(MODE_, NV_, true, 3); else if (pol == 3) PS_LAUNCH_TP(MODE_, NV_, false, 3); \
                                       else if (pol == 1) PS_LAUNCH_TP(MODE_, NV_, false, 1); else PS_LAUNCH_TP(MODE_, NV_, false, 0); } while (0)
            if (mode == 3) {
                if (!M.packed) throw Error("internal: fused residual update on the fp64 stream");
                if (M.nv == 1) PS_LAUNCH_TP2(3, 1); else PS_LAUNCH_TP2(3, 2);
                return;
            }
            if (M.nv == 1) { if (mode == 0) PS_LAUNCH_TP2(0, 1); else if (mode == 1) PS_LAUNCH_TP2(1, 1); else PS_LAUNCH_TP2(2, 1); }
            else { if (mode == 0) PS_LAUNCH_TP2(0, 2); else if (mode == 1) PS_LAUNCH_TP2(1, 2); else PS_LAUNCH_TP2(2, 2); }
#undef PS_LAUNCH_TP2
#undef PS_LAUNCH_TP
            return;
        }
        if (mode >= 2) throw Error("internal: fused St epilogue without the pipelined St kernel");
        spmvSt_(mode, t, xin, add, out, partial);
    }
    // grid of a persistent kernel; the XCD-grouped walk needs a multiple of 8 blocks (workgroup b runs on XCD b & 7)
    // The fp64-value stream (10 B per entry) runs one chunk per workgroup: measured at 256^3 St 0.54 ms against 0.64 ms
    // persistent (the persistent walk pays when the stream is short and the loop is issue-bound, not when it is 3x heavier).
    int pipeBlocks(int nChunks, int& xcd, bool packed, int gridCap = 0) const {
        if (!packed) { xcd = 0; return nChunks; }
        int g = std::min(nChunks, gridCap > 0 ? gridCap : pipeGrid);
        if (xcd > 0) { if (g >= 8) g &= ~7; else xcd = 0; }
        // Balance (r06).  The two-unit kernels walk PAIRS of chunks, runs of 32 pairs per XCD: workgroup l of an XCD takes steps l, l + per, ... below
        // qEnd.  With a launch a little larger than its grid cap (a rank's 13.8 k chunks of S on 6144 workgroups) an eighth of the workgroups took
        // two steps and everybody waited for them: 59 us where two launches of half the chunks took 2 x 20.6.  Shrink the grid to the size at which every
        // workgroup takes the same number of steps (never above the cap; the 256^3 single domain keeps its 6144 / 1536: 9 / 36 steps each).
        if (xcd > 0 && g >= 8 && c->S.ellok && c->St.ellok) {
            const int nPairs = (nChunks + 1) >> 1, qEnd = ((nPairs + 255) >> 8) << 5;
            const int steps = std::max(1, (qEnd * 8 + g - 1) / g), per = (qEnd + steps - 1) / steps;
            if (per * 8 <= g) g = per * 8;
        }
        return g;
    }
    // the fused step needs both products on the persistent coded-stream kernels (their per-workgroup partials)
    bool fusedOk() const {
        return stOnPipe() && c->St.packed && pipeGrid > 0 && c->S.col16ok && c->S.packed && rowsS > 0;
    }
    int sBlocks() const {
        const int nChunks = c->S.nChunks;
        int xcd = xcdAware;
        return pipeBlocks(nChunks, xcd, true, sCap());
    }
    // The row-per-lane S kernel has no per-workgroup prologue and balances better on more, shorter workgroups — 256^3, same box:
    // 4096 workgroups 0.265 ms, 5120 0.260, 6144 0.257, 8192 0.258, 12288 0.250 (each is one more partial sum for every St workgroup to read)
    int sCap() const { return (c->S.ellok && c->S.packed && c->S.col16ok && pipeGrid == 4096) ? 6144 : 0; }
    // MODE 3 of the row-per-lane St kernel in its plain form (FX = 1: no Chebyshev first term, no halo rows, coded uInv) fits 7 workgroups per CU
    bool plain3Hint = false;
    bool plain3Hint2 = false;   // the same for the Chebyshev step: single domain, coded uInv
    // Workgroups of the St kernel.  With the residual update in its epilogue (mode 3) it runs best on 6 per CU — measured at 256^3,
    // rocprof average in a solve: 1280 / 1536 workgroups 415 us, 1792 489, 2048 445, 2560 / 3072 425, 4096 430 (and every workgroup
    // less is 10 K partial sums less to read in the prologue); S and the other St modes keep 16 per CU (S: 300 us at 4096, 324 at
    // 1536, 339 at 1024).  PS_PIPE_GRID_ST overrides.
    // Row-per-lane kernel, plain MODE 3: 7 per CU — 1536 workgroups 0.418 ms, 1792 0.403, 2048 0.515 (the eighth does not fit: a second round),
    // 3584 / 5376 as 1792.
    bool stDual() const {
        static const bool on = !(PS_ENV("PS_ST_DUAL") && atoi(PS_ENV("PS_ST_DUAL")) == 0);
        return on && plain3Hint && c->St.ellok && c->St.packed && c->St.col16ok && pipeGrid >= 1536;
    }
    int stGridFor(int mode) const {
        if (stGrid > 0) return stGrid;
        static const int g2 = PS_ENV("PS_PIPE_GRID_ST2") ? atoi(PS_ENV("PS_PIPE_GRID_ST2")) : 0;   // A/B: the Chebyshev term's launch alone
        if (mode == 2 && g2 > 0) return g2;
        if (mode != 3 || pipeGrid < 1536) return 0;
        if (stDual()) return 1536;   // k_spmv_St_ell2: 80 VGPRs, six workgroups per CU
        return (plain3Hint && c->uCoded && c->St.ellok && c->St.packed && pipeGrid >= 1792) ? 1792 : 1536;
    }
    int stBlocks(int mode = 0) const {   // number of partials the St kernel writes: one per block
        int xcd = xcdAware;
        return stOnPipe() ? pipeBlocks(c->St.nChunks, xcd, c->St.packed, stGridFor(mode)) : gridFor(rowsSt, BS);
    }
};
Launch mk(ps_context* c, const int* done) {
    Launch L;
    L.c = c; L.done = done;
    L.rowsS = (int)c->nRows; L.rowsSt = (int)c->nSystem; L.nA = (int)c->nActiveVs; L.nP = (int)c->nPressures;
    static int pg = -1;
    if (pg < 0) {
        const char* g = PS_ENV("PS_PIPE_GRID");   // A/B switch: 0 = one-shot kernels
        pg = g ? atoi(g) : 4096;                   // persistent pipelined kernels, 16 blocks per CU, by default
    }
    L.pipeGrid = pg;
    static const int sg = PS_ENV("PS_PIPE_GRID_ST") ? atoi(PS_ENV("PS_PIPE_GRID_ST")) : 0;
    L.stGrid = pg > 0 ? sg : 0;
    static int xa = -1;
    if (xa < 0) { const char* e = PS_ENV("PS_XCD"); xa = e ? atoi(e) : 64; }   // chunks per XCD run (rounded down to a power of two); 0: plain walk.  64: same kernel times as 4 / 16 / 256 on the row-per-lane kernels, a fifth less HBM-side traffic than 16 (FETCH_SIZE of S 0.64 / 0.54 / 0.44 / 0.42 M KiB at 4 / 16 / 64 / 256)
    L.xcdAware = xa > 0 ? xa : 0;
    // log2 of the consecutive chunks a workgroup takes in a row (ChunkWalk): 2 chunks on the row-per-lane kernels (256^3, same box: S 0.264 ->
    // 0.257 ms, St with the residual update 0.431 -> 0.408; 4 / 8 / 16 chunks: S 0.282 / 0.274 / 0.273, St 0.412 / 0.412 / 0.428)
    static const int wr = PS_ENV("PS_WG_RUN") ? atoi(PS_ENV("PS_WG_RUN")) : -1;
    const int run = wr >= 0 ? wr : ((c->S.ellok && c->St.ellok) ? 1 : 0);
    if (L.xcdAware > 0) L.xcdAware |= (run & 7) << 16;
    L.ntSpmv = c->ntLevel() >= 1;
    L.plain3Hint = c->P.preconditioner != PS_PRE_CHEBYSHEV && !c->slabEnabled;     // (the stress diagonal coded or not: the two-unit kernels take both, r06)
    L.plain3Hint2 = c->P.preconditioner == PS_PRE_CHEBYSHEV && !c->slabEnabled;
    return L;
}
constexpr int64_t FUSED_STEP_MIN_ROWS = 1200000;   // see solve() (r05: 2 M -> 1.2 M: the coil 128^3 of BASELINE config 2, 1.49 M rows, solves 3 % faster in four kernels — 10.55 against 10.86 ms,
                                                    // two rounds on one box; the 64^3 cavity, 0.8 M rows, stays faster in five: 57.3 against 58.8 us per iteration)
constexpr int64_t NT_LEVEL1_MIN_ROWS = 4000000, NT_LEVEL2_MIN_ROWS = 10000000;   // see ps_context::ntLevel
int dotBlocks(int64_t n) { return (int)std::min<int64_t>(VGRID, std::max<int64_t>(1, (n + BS - 1) / BS)); }
}  // namespace

// y = A x on device vectors (ApplyPressureStressMatrix::apply).  dotPartialsOut receives the per-block
// partials of x.y (gridFor(nSystem,256) entries).
void ps_context::applyOperator(const double* xdev, double* ydev, double* dotPartialsOut) {
    Launch L = mk(this, nullptr);
    L.spmvS(0, xdev, ts.p);
    L.tiles(0, ts.p);
    L.spmvSt(0, ts.p, xdev, nullptr, ydev, dotPartialsOut);
}

// AssembleSystem.cpp:432-470 (+ the reduced blocks of AssembleBlocks.cpp)
void ps_context::assembleSystemPressureStressFactored() {
    assembleReducedBlocks();
    const int64_t n = nSystem;
    ts.alloc((size_t)nRows + 1);
    vreg.alloc((size_t)std::max<int64_t>(1, regionCount) * PS_RD);
    wreg.alloc((size_t)std::max<int64_t>(1, nRChunks) * PS_RD);
    b.alloc((size_t)n); x.alloc((size_t)n); r.alloc((size_t)n); pvec.alloc((size_t)n); Ap.alloc((size_t)n);
    dotPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor(std::max<int64_t>(n, 1), BS)) + 16);
    scal.alloc(1);
    dotPartials2.alloc(RED_BLOCKS);
    dotPartials3.alloc(VGRID);
    dotPartialsR.alloc(2 * VGRID);
    // t0 = McInv rhs_a on active rows, C (invDt BInv rhs_r) on reduced rows;  b = -S^T t0 + [rhs_p; rhs_tau]
    if (nActiveVs > 0)
        hipLaunchKernelGGL(k_scale_rows, dim3(dotBlocks(nActiveVs)), dim3(BS), 0, stream, ts.p, McInv.p, rhsA.p, nActiveVs);
    Launch L = mk(this, nullptr);
    L.tiles(2, ts.p);
    L.spmvSt(1, ts.p, nullptr, rhsPT.p, b.p, nullptr);
    HIP_CHECK(hipMemsetAsync(x.p, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), stream));
}

// Preconditioners.cpp:4-9 (identity) / Jacobi extension
void ps_context::constructPreconditioner() {
    if (P.preconditioner != PS_PRE_DIAGONAL && P.preconditioner != PS_PRE_CHEBYSHEV && P.solverType != PS_EIGEN) return;   // Eigen's CG always runs its DiagonalPreconditioner
    dinv.alloc((size_t)nSystem);
    if (nSystem == 0) return;
    hipLaunchKernelGGL(k_jacobi_diag, dim3(gridFor(nSystem, 256)), dim3(256), 0, stream, St.ptr.p, St.col.p, (const double*)St.val.p, (const int8_t*)St.code.p, valScale, (int)nSystem,
                       (int)nPressures, (int)nActiveVs, dt, McInv.p, uInv.p, rrowFace.p, rrowRegion.p, COM.p, dx, make_int3(gOff[0], gOff[1], gOff[2]), Binv.p, dinv.p,
                       slabEnabled ? 0 : 1);
    // The PCG kernels read the diagonal in 16 bits (ps_common.hpp: diag_t — 2 instead of 8 bytes per DOF in both step kernels; fp32 until
    // r05).  Any diagonal of the operator's sign is a valid preconditioner; the Jacobi option itself is an extension (the reference's is a stub,
    // Preconditioners.cpp:37-41).  The fp64 array stays for export / tests / Eigen's CG / the interval estimate of the Chebyshev polynomial
    // (estimateLambdaMax: the power iteration runs on D64^-1 A while the polynomial applies D16^-1 A, whose entries differ by <= 2^-8: the
    // spectrum of D16^-1 A lies within (1 +- 0.004) of the other's, an order of magnitude inside the 1.25 x margin and the 8.4 floor of the
    // estimate — the oracle's estimate is on the fp64 diagonal too, so the two intervals agree to rounding); with a slab the conversion
    // follows the cross-rank completion of the diagonal (Dist::finishSetup).
    dinvF.alloc((size_t)nSystem);
    if (!slabEnabled) hipLaunchKernelGGL(k_to_diag, dim3(dotBlocks(nSystem)), dim3(BS), 0, stream, dinv.p, dinvF.p, nSystem);
    if (P.preconditioner == PS_PRE_CHEBYSHEV && !slabEnabled) estimateLambdaMax();   // with a slab: Dist::finishSetup, across the ranks
}

// lambda_max(D^-1 A) for the Chebyshev polynomial: 10 power iterations from the all-ones vector, Rayleigh quotient of the
// last iterate, then max(8.4, 1.25 * estimate) — same procedure as the oracle (ps_oracle_solve.cpp:estimateLambdaMax).
// The stencil part of A is a sum of rank-one face terms with <= 8 entries, so its lambda_max(D^-1 A) <= 8 by Cauchy-Schwarz;
// the measurement covers the tile part.  10 applies at setup (~1 % of a 256^3 step).
// Cache policy by system size (PS_NT_LEVEL = 0 / 1 / 2 forces): 2 = non-temporal streams in the SpMV kernels and the vector
// kernels (a 45 M-row iteration moves 6.7 GB: nothing survives to the next kernel, and keeping the once-per-launch streams out of
// the way of the gathers is worth 7 % of a step), 1 = in the SpMV kernels only, 0 = default policy everywhere (the working set of
// an iteration, ~150 B per row, fits the 256 MB memory-side cache or nearly: let it serve the next kernel).  Measured us per
// iteration at level 0 / 1 / 2 (cavity): 64^3 (0.8 M rows) 58.6 / 60.6 / 61.2; 96^3 (2.6 M) 97.9 / 104.8 / 103.4; 128^3 (5.9 M) 194.0 /
// 187.4 / 191.8; 160^3 (11.4 M) 341.7 / 337.7 / 336.0; 192^3 (19.4 M) 558 / 535 / 530; 224^3 (30.6 M) 858 / 838 / 810.
int ps_context::ntLevel() const {
    static const int env = PS_ENV("PS_NT_LEVEL") ? atoi(PS_ENV("PS_NT_LEVEL")) : -1;
    if (env >= 0) return env;
    const int64_t rows = std::max(nSystem, deviceShareRows);   // (ranks of an in-process group share the device's caches: ps_context::deviceShareRows)
    return rows < NT_LEVEL1_MIN_ROWS ? 0 : (rows < NT_LEVEL2_MIN_ROWS ? 1 : 2);
}

static double chebRatio() { static const double r = PS_ENV("PS_CHEB_RATIO") ? atof(PS_ENV("PS_CHEB_RATIO")) : PS_CHEB_INTERVAL_RATIO; return r; }   // lmax / lmin (PS_CHEB_RATIO: experiments only — the oracle uses the constant)
double ps_context::chebTheta() const { return 0.5 * (chebLmax + chebLmax / chebRatio()); }   // centre of the interval [lmax/250, lmax]

void ps_context::estimateLambdaMax() {
    // 10 steps of the power iteration on D^-1 A from the ones vector, Rayleigh quotient of the last step (oracle:
    // estimateLambdaMax).  The iterate is not normalised between steps (the quotient does not depend on its length and the
    // spectrum lies in (0, ~8]: ten steps grow it by < 1e10), so nothing comes back to the host until the end.
    const int64_t n = nSystem;
    chebLmax = 8.4;
    if (n == 0) return;
    const int vb = dotBlocks(n);
    tmp1.alloc((size_t)n); tmp2.alloc((size_t)n); tmp3.alloc((size_t)n);
    double* v = tmp1.p; double* w = tmp2.p; double* Av = tmp3.p;
    chebPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor(n, BS)) + 16);
    hipLaunchKernelGGL(k_fill_f64, dim3(vb), dim3(BS), 0, stream, v, 1., n);
    for (int it = 0; it < 10; ++it) {
        applyOperator(v, Av, dotPartials.p);
        hipLaunchKernelGGL(k_power_step, dim3(vb), dim3(BS), 0, stream, (const double*)v, (const double*)Av, (const double*)dinv.p, w, n, chebPartials.p);
        std::swap(v, w);
    }
    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(BS), 0, stream, chebPartials.p, vb, chebPartials.p + 2 * vb);
    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(BS), 0, stream, chebPartials.p + vb, vb, chebPartials.p + 2 * vb + 1);
    double h[2];
    HIP_CHECK(hipMemcpyAsync(h, chebPartials.p + 2 * vb, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const double lam = (h[0] > 0. && std::isfinite(h[1] / h[0])) ? h[1] / h[0] : 0.;   // A v = 0 on the way: the floor below
    chebLmax = std::max(8.4, 1.25 * lam);
}

// z = q(D^-1 A) D^-1 r: k terms of the Chebyshev iteration on [lmax/250, lmax] (k-1 operator applies), see include/polystokes.h.
// Three-term form: z_1 = D^-1 r / theta, z_{j+1} = z_j + c1 (z_j - z_{j-1}) + c2 D^-1 (r - A z_j) — two buffers, zA and zB, taking turns
// (z_1 in zA, z_2 in zB, z_3 in zA, ...); *zOut is the one holding the final z.  Terms 2..k run as S, tiles and the St kernel with the
// update fused into its epilogue (MODE 2): per term it reads r, dinv, z_{j-1} besides its own operands and writes z_{j+1} over
// z_{j-1} — no separate vector pass.  rzPartial receives the partials of r.z of the final z (count returned); `sc` (may be null) lets
// the kernels of a converged solve exit early.
// firstDone: the caller already holds z_1 in zA (the St kernel of the four-kernel PCG step forms it on the rows it updates) — with
// a one-term polynomial nothing is launched and 0 is returned.
int ps_context::chebyshevApply(const double* rvec, double* zA, double* zB, double* rzPartial, const ps::CGScalars* sc, bool firstDone, double** zOut) {
    const int64_t n = nSystem;
    const int k = P.preconditionerDegree > 0 ? P.preconditionerDegree : 4;
    const double lmax = chebLmax, lmin = lmax / chebRatio();
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1. / sigma;
    const int vb = dotBlocks(n);
    const int* done = sc ? &sc->done : nullptr;
    Launch L = mk(this, done);
    if (chebInner32) {
        // PS_PRE_CHEBYSHEV_F32: the same recurrence with z_j (zA / zB) and the face-row vector of the inner applies (the first half of ts) STORED as
        // fp32; r, the diagonal, every product and sum fp64.  The two-units-per-wave kernels only (Launch::cheb32Ok decided chebInner32).
        float* cur = (float*)zA; float* other = (float*)zB; float* tsF = (float*)ts.p;
        if (!firstDone) hipLaunchKernelGGL(k_cheb_first<float>, dim3(vb), dim3(BS), 0, stream, sc, rvec, (const diag_t*)dinvF.p, 1. / theta, cur, n, rzPartial);
        int count = firstDone ? 0 : vb;
        for (int j = 1; j < k; ++j) {
            const double rhoN = 1. / (2. * sigma - rho);
            const double c1 = rhoN * rho, c2 = 2. * rhoN / delta;
            const ChebArgs ca{rvec, dinvF.p, j == 1 ? (const double*)nullptr : (const double*)other, c1, c2};   // (zprev points at floats: k_spmv_St_ell2c<.., float>)
            L.spmvS32(cur, tsF);
            L.tiles32(tsF);
            count = L.spmvSt2c32(tsF, cur, other, rzPartial, ca);
            std::swap(cur, other);
            rho = rhoN;
        }
        if (zOut) *zOut = (double*)cur;
        return count;
    }
    if (!firstDone) hipLaunchKernelGGL(k_cheb_first<double>, dim3(vb), dim3(BS), 0, stream, sc, rvec, (const diag_t*)dinvF.p, 1. / theta, zA, n, rzPartial);
    int count = firstDone ? 0 : vb;
    double* cur = zA; double* other = zB;    // z_j, and the buffer of z_{j-1} that receives z_{j+1}
    for (int j = 1; j < k; ++j) {
        const double rhoN = 1. / (2. * sigma - rho);
        const double c1 = rhoN * rho, c2 = 2. * rhoN / delta;
        const double* zprev = j == 1 ? nullptr : other;            // z_0 = 0
        L.spmvS(0, cur, ts.p);
        L.tiles(0, ts.p);
        if (L.stOnPipe()) {
            const ChebArgs ca{rvec, dinvF.p, zprev, c1, c2};
            L.spmvSt(2, ts.p, cur, nullptr, other, rzPartial, &ca);
            count = L.stBlocks();
        } else {
            tmp5.alloc((size_t)n);
            L.spmvSt(0, ts.p, cur, nullptr, tmp5.p, dotPartials2.p);
            hipLaunchKernelGGL(k_cheb_step, dim3(vb), dim3(BS), 0, stream, sc, rvec, (const diag_t*)dinvF.p, (const double*)tmp5.p, c1, c2, (const double*)cur, zprev, other, n, rzPartial);
            count = vb;
        }
        std::swap(cur, other);
        rho = rhoN;
    }
    if (zOut) *zOut = cur;
    return count;
}

void ps_context::applyPreconditionerDevice(const double* rvec, double* z, double* scratch) {
    const int64_t n = nSystem;
    if (n == 0) return;
    const int vb = dotBlocks(n);
    if (P.preconditioner == PS_PRE_CHEBYSHEV) {
        chebPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor(n, BS)) + 16);
        double* zfin = z;
        chebInner32 = chebInner32Req && mk(this, nullptr).cheb32Ok();
        chebInner32Host = chebInner32 ? 1 : 0;
        HIP_CHECK(hipMemcpyAsync(counters.p + 36, &chebInner32Host, sizeof(int32_t), hipMemcpyHostToDevice, stream));   // (array "chebInner32")
        chebyshevApply(rvec, z, scratch, chebPartials.p, nullptr, false, &zfin);
        if (chebInner32) {      // the result is an fp32 vector in one of the two buffers: widen it through a third
            tmp5.alloc((size_t)n);
            hipLaunchKernelGGL(k_widen_f32, dim3(vb), dim3(BS), 0, stream, tmp5.p, (const float*)zfin, n);
            HIP_CHECK(hipMemcpyAsync(z, tmp5.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        } else
        if (zfin != z) HIP_CHECK(hipMemcpyAsync(z, zfin, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    } else if (P.preconditioner == PS_PRE_DIAGONAL) {
        hipLaunchKernelGGL(k_mul_diag, dim3(vb), dim3(BS), 0, stream, z, (const diag_t*)dinvF.p, rvec, n);   // the diagonal as the PCG kernels read it
    } else {
        HIP_CHECK(hipMemcpyAsync(z, rvec, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
}

// Solver.cpp:734-812 solveSPDwithMatrixVectorPCG -> pcg_external_matrix_A (pcg.h:268-340), BiCGStab fallback (pcg.h:134-200)
int ps_context::solve() {
    const int64_t n = nSystem;
    const int maxit = P.maxSolverIterations;
    const double tol = P.tolerance;
    usedBiCGStab = 0;
    interrupted = false;
    if (P.solverType == PS_EIGEN) return solveEigenCG();
    if (P.solverType != PS_PCG_MATRIX_VECTOR_PRODUCTS) { err = "Unsupported Solver."; return PS_UNSUPPORTED_SOLVER; }
    if (n == 0) { solveIterations = 0; solveError = 0; return PS_SUCCESS; }
    const bool cheb = P.preconditioner == PS_PRE_CHEBYSHEV;
    const diag_t* dv = (P.preconditioner == PS_PRE_DIAGONAL) ? dinvF.p : nullptr;
    const int vb = dotBlocks(n);
    CGScalars* sc = scal.p;
    const int* done = &sc->done;
    Launch L = mk(this, done);
    const int stBlocks = L.stBlocks(0), stBF = L.stBlocks(3);   // workgroups of the St kernel: plain / with the residual update
    double* zvec = nullptr; double* dvec = nullptr; double* rzPart = nullptr;
    chebInner32 = cheb && chebInner32Req && L.cheb32Ok();     // PS_PRE_CHEBYSHEV_F32 where the two-unit kernels run; fp64 inner vectors otherwise
    L.cz32 = chebInner32;
    if (cheb) {
        tmp1.alloc((size_t)n); tmp2.alloc((size_t)n);
        zvec = tmp1.p; dvec = tmp2.p;
        chebPartials.alloc((size_t)std::max<int64_t>(3 * VGRID, gridFor(n, BS)) + 16);
        chebPartials2.alloc(RED_BLOCKS);
        rzPart = chebPartials.p;
    }
    // r.z partials of the polynomial's last term, reduced to <= RED_BLOCKS values when there is one per 256-row chunk
    auto rzReduce = [&](const double* src, int count, const double*& part, int& cnt) {
        part = src; cnt = count;
        if (count > 8192) {
            hipLaunchKernelGGL(k_reduce_partials, dim3(RED_BLOCKS), dim3(BS), 0, stream, sc, src, count, chebPartials2.p);
            part = chebPartials2.p; cnt = RED_BLOCKS;
        }
    };

    // Fused step (default on the coded stream; PS_FUSED_R=0 keeps the five-kernel step): p.Ap = -(sum_active s.t + sum_tiles w.v
    // + 1/2 sum uInv p^2) is complete before the St kernel starts, so that kernel forms alpha and updates r in its epilogue —
    // A p is neither written nor read back (16 B per row less) and the step is four launches (FusedR, ps_kernels_spmv.hpp).
    // Every St workgroup sums the partials of three producers in its prologue (up to 4096 + regions + 1024 + 1024 values, from
    // L2): a fixed cost per iteration, against 16 B per row saved.  Measured us per iteration, four / five kernels (St on 1536
    // workgroups, shared runs): 64^3 (0.8 M rows) 58.8 / 57.3, 96^3 (2.6 M) 93.2 / 95.6, 128^3 (5.9 M) 164.8 / 175.5, 160^3 (11.4 M)
    // 285.9 / 309.3, 192^3 (19.4 M) 466.7 / 506.5, 256^3 (45 M) 1147 / 1248 (before shared runs) -> on from 1.2 M rows (FUSED_STEP_MIN_ROWS).  (Folding the
    // partials 64 to 1 in the producers with a ticket per group costs more than it saves: one device-scope atomic per workgroup,
    // +30 us per iteration with write-through stores and no fence, +650 us with __threadfence(), which flushes the XCD's L2.)
    // PS_FUSED_R = 0 / 1 forces it off / on (on only where the kernels exist).
    static const int fusedEnv = PS_ENV("PS_FUSED_R") ? atoi(PS_ENV("PS_FUSED_R")) : -1;
    const bool fused = fusedEnv != 0 && (fusedEnv > 0 || n >= FUSED_STEP_MIN_ROWS) && L.fusedOk();
    fusedStepHost = fused ? 1 : 0;
    const int sBlocks = fused ? L.sBlocks() : 0;
    double *fS = nullptr, *fT = nullptr, *fU = nullptr, *fR = nullptr;
    const uint8_t* ucode = uCoded ? uCode.p : nullptr;
    if (fused) {
        fusedPart.alloc((size_t)sBlocks + (size_t)regionCount + VGRID + 2 * (size_t)stBF + 16);
        fS = fusedPart.p; fT = fS + sBlocks; fU = fT + regionCount; fR = fU + VGRID;
        L.sPart = fS; L.wvPart = fT;
    }

    HIP_CHECK(hipMemsetAsync(dotPartials3.p, 0, VGRID * sizeof(double), stream));
    hipLaunchKernelGGL(k_cg_init_f, dim3(vb), dim3(BS), 0, stream, b.p, dv, x.p, r.p, pvec.p, n, dotPartials.p);
    if (cheb) {   // z = M^-1 r, p = z, rsold = r.z
        HIP_CHECK(hipMemsetAsync(sc, 0, sizeof(CGScalars), stream));   // `done` must read 0 inside the polynomial's kernels
        double* z0 = zvec;
        const int cnt0 = chebyshevApply(r.p, zvec, dvec, rzPart, nullptr, false, &z0);
        if (chebInner32) hipLaunchKernelGGL(k_widen_f32, dim3(vb), dim3(BS), 0, stream, pvec.p, (const float*)z0, n);
        else
        HIP_CHECK(hipMemcpyAsync(pvec.p, z0, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(BS), 0, stream, rzPart, cnt0, dotPartials.p);
        hipLaunchKernelGGL(k_cg_scal0, dim3(1), dim3(BS), 0, stream, sc, dotPartials.p, 1, tol, maxit, ntLevel() >= 2 ? 1 : 0);
    } else
    hipLaunchKernelGGL(k_cg_scal0, dim3(1), dim3(BS), 0, stream, sc, dotPartials.p, vb, tol, maxit, ntLevel() >= 2 ? 1 : 0);
    // the first direction's share of p.Ap on the diagonal
    if (fused) hipLaunchKernelGGL(k_uinv_pp, dim3(vb), dim3(BS), 0, stream, (const double*)pvec.p, ucode, (const double*)uDict.p, (const double*)uInv.p, n, fU);
    CGScalars h{};
    const int batch = 25;
    int it = 0;
    bool finished = false;
    while (it < maxit && !finished) {
        const int upto = std::min(maxit, it + batch);
        for (; it < upto; ++it) {
            L.spmvS(0, pvec.p, ts.p);
            L.tiles(0, ts.p);
            if (fused && cheb) {   // St kernel: r -= alpha A p and the polynomial's first term on the new r; then terms 2..k; then x, p
                const FusedR fr{sc, fS, sBlocks, fT, (int)regionCount, fU, vb, dotPartials3.p, vb, it, r.p, nullptr, fR, dinvF.p, 1. / chebTheta(), zvec, nullptr, 0, (int)n, nullptr};
                L.spmvSt(3, ts.p, pvec.p, nullptr, nullptr, nullptr, nullptr, &fr);
                double* zfin = zvec;
                const int c2 = chebyshevApply(r.p, zvec, dvec, rzPart, sc, true, &zfin);
                const double* part; int cnt;
                if (c2 > 0) rzReduce(rzPart, c2, part, cnt); else { part = fR + stBF; cnt = stBF; }
                if (chebInner32)
                hipLaunchKernelGGL(k_cg_update_xp_z_u<float>, dim3(vb), dim3(BS), 0, stream, sc, (const double*)fR, stBF, part, cnt, it, (const float*)zfin,
                                   x.p, pvec.p, n, dotPartials3.p, ucode, (const double*)uDict.p, (const double*)uInv.p, fU);
                else
                hipLaunchKernelGGL(k_cg_update_xp_z_u<double>, dim3(vb), dim3(BS), 0, stream, sc, (const double*)fR, stBF, part, cnt, it, (const double*)zfin,
                                   x.p, pvec.p, n, dotPartials3.p, ucode, (const double*)uDict.p, (const double*)uInv.p, fU);
                continue;
            }
            if (fused) {
                const FusedR fr{sc, fS, sBlocks, fT, (int)regionCount, fU, vb, dotPartials3.p, vb, it, r.p, dv, fR, nullptr, 0., nullptr, nullptr, 0, (int)n, nullptr};
                L.spmvSt(3, ts.p, pvec.p, nullptr, nullptr, nullptr, nullptr, &fr);
                hipLaunchKernelGGL(k_cg_update_xp_u, dim3(vb), dim3(BS), 0, stream, sc, (const double*)nullptr, (const double*)fR, stBF, dv ? 1 : 0, it, (const double*)r.p, dv, x.p,
                                   pvec.p, n, dotPartials3.p, ucode, (const double*)uDict.p, (const double*)uInv.p, fU);
                continue;
            }
            L.spmvSt(0, ts.p, pvec.p, nullptr, Ap.p, dotPartials.p);
            const double* pApPart = dotPartials.p;
            int pApCount = stBlocks;
            if (stBlocks > 8192) {   // one-shot St kernel: one partial per 256 rows, reduced in two stages
                hipLaunchKernelGGL(k_reduce_partials, dim3(RED_BLOCKS), dim3(BS), 0, stream, sc, dotPartials.p, stBlocks, dotPartials2.p);
                pApPart = dotPartials2.p; pApCount = RED_BLOCKS;
            }
            hipLaunchKernelGGL(k_cg_update_r, dim3(vb), dim3(BS), 0, stream, sc, (const double*)nullptr, pApPart, pApCount, dotPartials3.p, vb, it, Ap.p, dv,
                               r.p, n, dotPartialsR.p);
            if (cheb) {
                const double* part; int cnt;
                double* zfin = zvec;
                rzReduce(rzPart, chebyshevApply(r.p, zvec, dvec, rzPart, sc, false, &zfin), part, cnt);
                if (chebInner32)
                hipLaunchKernelGGL(k_cg_update_xp_z<float>, dim3(vb), dim3(BS), 0, stream, sc, (const double*)dotPartialsR.p, vb, part, cnt, it, (const float*)zfin,
                                   x.p, pvec.p, n, dotPartials3.p);
                else
                hipLaunchKernelGGL(k_cg_update_xp_z<double>, dim3(vb), dim3(BS), 0, stream, sc, (const double*)dotPartialsR.p, vb, part, cnt, it, (const double*)zfin,
                                   x.p, pvec.p, n, dotPartials3.p);
            } else
            hipLaunchKernelGGL(k_cg_update_xp, dim3(vb), dim3(BS), 0, stream, sc, (const double*)nullptr, dotPartialsR.p, vb, dv ? 1 : 0, it, r.p, dv, x.p,
                               pvec.p, n, dotPartials3.p);
        }
        hipLaunchKernelGGL(k_cg_check, dim3(1), dim3(BS), 0, stream, sc, (const double*)nullptr, dotPartials3.p, vb, it - 1);
        HIP_CHECK(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (h.done) finished = true;
        if (!finished && interruptCb && interruptCb(interruptUser)) { interrupted = true; break; }
    }
    if (interrupted) { solveIterations = it; solveError = std::sqrt(h.rre); return PS_INCOMPLETE; }
    solveIterations = h.done ? h.iter : maxit;
    solveError = std::sqrt(h.rre);

    if (solveIterations == maxit) {
        // bicgstab_external_matrix_A (pcg.h:134-200), restarted from zero (Solver.cpp:784-799).  Rare path: host-driven.
        usedBiCGStab = 1;
        tmp1.alloc((size_t)n); tmp2.alloc((size_t)n); tmp3.alloc((size_t)n); tmp4.alloc((size_t)n); tmp5.alloc((size_t)n);
        double* rhat = tmp1.p; double* v = tmp2.p; double* s = tmp3.p; double* t = tmp4.p; double* e = tmp5.p;
        double* hvec = Ap.p;
        auto dotH = [&](const double* a, const double* bb) {
            hipLaunchKernelGGL(k_dot, dim3(vb), dim3(BS), 0, stream, a, bb, n, dotPartials.p);
            hipLaunchKernelGGL(k_sum1, dim3(1), dim3(BS), 0, stream, dotPartials.p, vb, dotPartials.p + 3 * VGRID);
            double out;
            HIP_CHECK(hipMemcpyAsync(&out, dotPartials.p + 3 * VGRID, sizeof(double), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            return out;
        };
        auto lin = [&](double* out, double ca, const double* a, double cb, const double* bb, double cc, const double* c3) {
            hipLaunchKernelGGL(k_lin, dim3(vb), dim3(BS), 0, stream, out, ca, a, cb, bb, cc, c3, n);
        };
        HIP_CHECK(hipMemsetAsync(x.p, 0, (size_t)n * sizeof(double), stream));
        lin(r.p, 1., b.p, 0., nullptr, 0., nullptr);           // r = b - A*0
        lin(rhat, 1., r.p, 0., nullptr, 0., nullptr);
        HIP_CHECK(hipMemsetAsync(pvec.p, 0, (size_t)n * sizeof(double), stream));
        HIP_CHECK(hipMemsetAsync(v, 0, (size_t)n * sizeof(double), stream));
        double rhoCurr = 1., rhoOld = 1., alpha = 1., beta = 0., omega = 1., rre = 0.;
        int i = 0;
        solveIterations = maxit;
        for (; i < maxit; ++i) {
            rhoOld = rhoCurr;
            rhoCurr = dotH(rhat, r.p);
            beta = (rhoCurr / rhoOld) * (alpha / omega);
            lin(pvec.p, 1., r.p, beta, pvec.p, -beta * omega, v);      // p = r + beta (p - omega v)
            applyOperator(pvec.p, v, dotPartials.p);
            alpha = rhoCurr / dotH(rhat, v);
            lin(hvec, 1., x.p, alpha, pvec.p, 0., nullptr);            // h = x + alpha p
            lin(s, 1., r.p, -alpha, v, 0., nullptr);                   // s = r - alpha v
            applyOperator(s, t, dotPartials.p);
            omega = dotH(t, s) / dotH(t, t);
            lin(x.p, 1., hvec, omega, s, 0., nullptr);                 // x = h + omega s
            const double xmag = std::sqrt(dotH(x.p, x.p));
            applyOperator(x.p, e, dotPartials.p);
            lin(e, 1., b.p, -1., e, 0., nullptr);                      // err = b - A x
            const double rsnew = dotH(e, e);
            rre = rsnew;
            if (std::sqrt(rsnew) / xmag < rre) rre = std::sqrt(rsnew) / xmag;
            if (rre < tol) { solveIterations = i; break; }
            lin(r.p, 1., s, -omega, t, 0., nullptr);                   // r = s - omega t
        }
        solveError = rre;
    }
    return solveIterations == maxit ? PS_NOCONVERGE : PS_SUCCESS;
}

// initializeGuessVectors + constructGuessVectors (Solver.cpp:512-531) and the guessVector of the assemble functions
// (AssembleSystem.cpp:461-467):  pressureGuess = -G^T oldVs - JG^T cfit,  stressGuess = -2 uInv (-Dt^T oldVs - JDt^T cfit).
// With t = [oldVs ; C_f . cfit_region(f)] on the face rows this is one transposed product: g = -S^T t, stress part scaled.
void ps_context::constructGuessVectors() {
    const int64_t n = nSystem;
    guess.alloc((size_t)std::max<int64_t>(n, 1));
    HIP_CHECK(hipMemsetAsync(guess.p, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), stream));
    if (!P.useWarmStart || n == 0 || slabEnabled) return;
    if (nActiveVs > 0) HIP_CHECK(hipMemcpyAsync(ts.p, oldVs.p, (size_t)nActiveVs * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (regionCount > 0 && nRChunks > 0)
        hipLaunchKernelGGL(k_tile_expand, dim3((unsigned)nRChunks), dim3(BS), 0, stream, rchunkRegion.p, rchunkStart.p, rchunkEnd.p, rrowFace.p,
                           COM.p, dx, make_int3(gOff[0], gOff[1], gOff[2]), cfit.p, ts.p + nActiveVs, (const int*)nullptr);
    Launch L = mk(this, nullptr);
    L.spmvSt(1, ts.p, nullptr, x.p, guess.p, nullptr);   // x is zero here (assemble): guess = -S^T t
    hipLaunchKernelGGL(k_guess_finish, dim3(dotBlocks(n)), dim3(BS), 0, stream, guess.p, uInv.p, permSys.p, nPressures, n);
}

// solveEigenCG (Solver.cpp:814-862): Eigen::ConjugateGradient<SparseMatrix, Lower|Upper> with its default diagonal
// preconditioner, solveWithGuess(b, guessVector) — extern/eigen/Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:30-93,
// BasicPreconditioners.h:69-77 — run on the factored device operator instead of an assembled A (same A x; BASELINE config 1,
// a plumbing path: host-driven loop, one scalar read-back per dot product).  Stop rule ||r||^2 < tol^2 ||b||^2, returned
// count = completed iterations, error = ||r|| / ||b||, SUCCESS iff error <= tol (IterativeSolverBase::info()).
int ps_context::solveEigenCG() {
    const int64_t n = nSystem;
    const int maxit = P.maxSolverIterations;
    const double tol = P.tolerance;
    usedBiCGStab = 0;
    interrupted = false;
    if (slabEnabled) { err = "solverType EIGEN is a single-domain path"; return PS_UNSUPPORTED_SOLVER; }
    if (n == 0) { solveIterations = 0; solveError = 0; return PS_SUCCESS; }
    const int vb = dotBlocks(n);
    tmp1.alloc((size_t)n);
    double* z = tmp1.p;
    auto dotH = [&](const double* a, const double* bb) {
        hipLaunchKernelGGL(k_dot, dim3(vb), dim3(BS), 0, stream, a, bb, n, dotPartials.p);
        hipLaunchKernelGGL(k_sum1, dim3(1), dim3(BS), 0, stream, dotPartials.p, vb, dotPartials.p + 3 * VGRID);
        double out;
        HIP_CHECK(hipMemcpyAsync(&out, dotPartials.p + 3 * VGRID, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return out;
    };
    auto lin = [&](double* out, double ca, const double* a, double cb, const double* bb) {
        hipLaunchKernelGGL(k_lin, dim3(vb), dim3(BS), 0, stream, out, ca, a, cb, bb, 0., (const double*)nullptr, n);
    };
    HIP_CHECK(hipMemcpyAsync(x.p, guess.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));   // solveWithGuess
    applyOperator(x.p, Ap.p, dotPartials.p);
    lin(r.p, 1., b.p, -1., Ap.p);                                   // residual = rhs - A x
    const double rhsNorm2 = dotH(b.p, b.p);
    if (rhsNorm2 == 0.) {
        HIP_CHECK(hipMemsetAsync(x.p, 0, (size_t)n * sizeof(double), stream));
        solveIterations = 0; solveError = 0.;
        return PS_SUCCESS;
    }
    const double threshold = std::max(tol * tol * rhsNorm2, std::numeric_limits<double>::min());
    double residualNorm2 = dotH(r.p, r.p);
    int i = 0;
    if (residualNorm2 >= threshold) {
        hipLaunchKernelGGL(k_mulv, dim3(vb), dim3(BS), 0, stream, pvec.p, dinv.p, r.p, n);   // p = precond.solve(residual)
        double absNew = dotH(r.p, pvec.p);
        while (i < maxit) {
            applyOperator(pvec.p, Ap.p, dotPartials.p);
            const double alpha = absNew / dotH(pvec.p, Ap.p);
            lin(x.p, 1., x.p, alpha, pvec.p);
            lin(r.p, 1., r.p, -alpha, Ap.p);
            residualNorm2 = dotH(r.p, r.p);
            if (residualNorm2 < threshold) break;
            hipLaunchKernelGGL(k_mulv, dim3(vb), dim3(BS), 0, stream, z, dinv.p, r.p, n);
            const double absOld = absNew;
            absNew = dotH(r.p, z);
            const double beta = absNew / absOld;
            lin(pvec.p, 1., z, beta, pvec.p);
            ++i;
            if ((i % 25) == 0 && interruptCb && interruptCb(interruptUser)) { interrupted = true; break; }
        }
    }
    solveIterations = i;
    solveError = std::sqrt(residualNorm2 / rhsNorm2);
    if (interrupted) return PS_INCOMPLETE;
    return solveError <= tol ? PS_SUCCESS : PS_NOCONVERGE;
}

// Solver.cpp:492-510
void ps_context::recoverVelocityFromPressureStress() {
    recovered.alloc((size_t)(nActiveVs + nReducedVs) + 1);
    Launch L = mk(this, nullptr);
    L.spmvS(1, x.p, ts.p);
    if (nActiveVs > 0)
        hipLaunchKernelGGL(k_recover_active, dim3(dotBlocks(nActiveVs)), dim3(BS), 0, stream, ts.p, McInv.p, rhsA.p, dt, invDt, nActiveVs, recovered.p);
    L.tiles(1, ts.p);
    if (regionCount > 0)
        HIP_CHECK(hipMemcpyAsync(recovered.p + nActiveVs, vreg.p, (size_t)nReducedVs * sizeof(double), hipMemcpyDeviceToDevice, stream));
}

// Solver.cpp:937-1028
void ps_context::applySolutionToVelocity() {
    for (int a = 0; a < 3; ++a) {
        const int64_t n = g.count(1 + a);
        hipLaunchKernelGGL(k_writeback, dim3(gridFor(n, BS)), dim3(BS), 0, stream, g, a, labels[1 + a].p, activeIdx[1 + a].p, reducedIdx[1 + a].p,
                           faceRow[a].p, recovered.p, recovered.p + nActiveVs, COM.p, dx, make_int3(gOff[0], gOff[1], gOff[2]), cvel[a].p, vel[a].p, velOut[a].p, 1);
    }
}

// micro-benchmark dispatch for ps_bench_kernel (bench.py roofline object)
void ps_bench_launch(ps_context* c, const std::string& k, const double* x, double* y) {
    Launch L = mk(c, nullptr);
    // "<name>_fp64": the pipelined kernels on the fp64-value form of the stream (16-bit windowed columns + fp64 values, 10 B per
    // entry: what runs when the stencil values are not code * scale);  "<name>_csr": the one-shot kernels on the plain CSR
    // (int32 columns + fp64 values, 12 B per entry: what runs when a chunk needs more than 16 column windows)
    auto endsWith = [&](const char* suf) { const size_t m = std::strlen(suf); return k.size() > m && k.compare(k.size() - m, m, suf) == 0; };
    const bool fp64 = endsWith("_fp64"), csr = endsWith("_csr");
    const std::string base = fp64 ? k.substr(0, k.size() - 5) : (csr ? k.substr(0, k.size() - 4) : k);
    const bool keepS = c->S.packed, keepT = c->St.packed, keepS16 = c->S.col16ok, keepT16 = c->St.col16ok;
    struct Restore { ps_context* c; bool a, b, d, e; ~Restore() { c->S.packed = a; c->St.packed = b; c->S.col16ok = d; c->St.col16ok = e; } } restore{c, keepS, keepT, keepS16, keepT16};
    if (fp64) {
        if (!c->S.col16ok || !c->St.col16ok) throw Error("no compressed stream on this system");
        c->buildVal4(c->S); c->buildVal4(c->St);
        c->S.packed = false; c->St.packed = false;
    }
    if (csr) { c->ensureValues(c->S); c->ensureValues(c->St); c->S.packed = c->St.packed = false; c->S.col16ok = c->St.col16ok = false; }
    if (base == "spmv_S") L.spmvS(0, x, c->ts.p);
    else if (base == "spmv_St") L.spmvSt(0, c->ts.p, x, nullptr, y, c->dotPartials.p);
    else if (base == "apply") c->applyOperator(x, y, c->dotPartials.p);
    else if (base == "tiles") L.tiles(0, c->ts.p);
    else if (base == "cg_update_xr" || base == "cg_update_p" || base == "cg_update_r" || base == "cg_update_xp" || base == "cg_update_xp_u" || base == "spmv_St_r") {
        // streaming vector kernels on scratch vectors (alpha = beta = 0 keeps them finite over many launches)
        ps::DevBuf<CGScalars>& scratch = c->benchScal;
        scratch.alloc(1);
        CGScalars h{};
        h.tol2 = -1.;                                  // the stop test never fires
        h.vecNT = c->ntLevel() >= 2 ? 1 : 0;
        if (base == "cg_update_xp" || base == "cg_update_xp_u") h.rsold2[0] = 1.;   // beta = 0 / 1 ; (cg_update_r, spmv_St_r: alpha = 0 / p.Ap with p.Ap = +-1024 below)
        HIP_CHECK(hipMemcpyAsync(scratch.p, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
        ps::DevBuf<double>& ones = c->benchOnes;       // input partials of the fused scalar prologues
        if (ones.n < (size_t)2 * VGRID) {
            ones.alloc((size_t)2 * VGRID);
            std::vector<double> hv((size_t)2 * VGRID, 1.);
            HIP_CHECK(hipMemcpyAsync(ones.p, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, c->stream));   // (the context's stream is non-blocking: nothing
            HIP_CHECK(hipStreamSynchronize(c->stream));                                                      //  orders it behind the legacy stream; hv dies here)
        }
        ps::DevBuf<double>& zeros = c->benchZeros;
        if (zeros.n < (size_t)2 * VGRID) { zeros.alloc((size_t)2 * VGRID); HIP_CHECK(hipMemsetAsync(zeros.p, 0, (size_t)2 * VGRID * 8, c->stream)); }
        const int64_t n = c->nSystem;
        const int vb = dotBlocks(n);
        const double* dv = c->P.preconditioner == PS_PRE_DIAGONAL ? c->dinv.p : nullptr;
        const diag_t* dvf = c->P.preconditioner == PS_PRE_DIAGONAL ? c->dinvF.p : nullptr;
        c->tmp4.alloc((size_t)n); c->tmp5.alloc((size_t)n);
        c->dotPartials.alloc((size_t)std::max(3 * std::max(vb, VGRID), 2 * L.stBlocks()) + 16);
        const uint8_t* ucode = c->uCoded ? c->uCode.p : nullptr;
        if (base == "spmv_St_r") {   // the St kernel of the four-kernel step: r (scratch) -= 0 * A x in the epilogue
            if (!L.fusedOk()) throw Error("no fused step on this system");
            const FusedR fr{scratch.p, ones.p, VGRID, zeros.p, 0, zeros.p, 0, ones.p, 0, 0, c->tmp5.p, dvf, c->dotPartials.p, nullptr, 0., nullptr, nullptr, 0, (int)n, nullptr};
            L.spmvSt(3, c->ts.p, x, nullptr, nullptr, nullptr, nullptr, &fr);
        }
        else if (base == "cg_update_xp_u")
            hipLaunchKernelGGL(k_cg_update_xp_u, dim3(vb), dim3(BS), 0, c->stream, scratch.p, (const double*)nullptr, (const double*)zeros.p, VGRID, dvf ? 1 : 0, 0, x, dvf,
                               c->tmp4.p, c->tmp5.p, n, c->dotPartials.p, ucode, (const double*)c->uDict.p, (const double*)c->uInv.p, c->dotPartials.p + VGRID);
        else if (base == "cg_update_xr")
            hipLaunchKernelGGL(k_cg_update_xr, dim3(vb), dim3(BS), 0, c->stream, scratch.p, x, y, dv, c->tmp4.p, c->tmp5.p, n, c->dotPartials.p);
        else if (base == "cg_update_p")
            hipLaunchKernelGGL(k_cg_update_p, dim3(vb), dim3(BS), 0, c->stream, scratch.p, x, dv, c->tmp4.p, n);
        else if (base == "cg_update_r")
            hipLaunchKernelGGL(k_cg_update_r, dim3(vb), dim3(BS), 0, c->stream, scratch.p, (const double*)nullptr, ones.p, VGRID, ones.p, 0, 0, y, dvf,
                               c->tmp5.p, n, c->dotPartials.p);
        else
            hipLaunchKernelGGL(k_cg_update_xp, dim3(vb), dim3(BS), 0, c->stream, scratch.p, (const double*)nullptr, zeros.p, VGRID, dvf ? 1 : 0, 0, x, dvf,
                               c->tmp4.p, c->tmp5.p, n, c->dotPartials.p);
    }
    else throw Error("unknown kernel name: " + k);
}

#include "ps_dist.hpp"
#include "ps_import.hpp"
