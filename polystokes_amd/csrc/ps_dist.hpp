// Multi-GPU: distributed PCG over z-slabs (RCCL or in-process ranks) and its C ABI.
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once

// =====================================================================================================
// Multi-GPU: distributed PCG over z-slabs.  Not in the reference (single process).
// The same kernels as above run on every rank over its local rows / owned DOF range; what is added is
//   * pack / unpack of the one-layer exchange lists,
//   * a transport (RCCL send/recv + all-reduce on the solver stream, or device copies between ranks that
//     live in one process), and
//   * two-phase scalar kernels so that the all-reduce sits between "local sum" and "use".
// =====================================================================================================
#include <dlfcn.h>
#include <cstring>
#include <cstdio>

struct PsNcclUid { char internal[128]; };   // layout of ncclUniqueId

namespace {

// both cut planes of a rank in one launch: entries [0, nA) use list A / buffer A, entries [nA, nA + nB) list B / buffer B.
// (A DOF lies next to at most one cut — slabs are at least one 16-layer block thick — so the two lists are disjoint.)
__global__ void k_pack2(const int32_t* __restrict__ listA, int64_t nA, double* __restrict__ bufA, const int32_t* __restrict__ listB, int64_t nB,
                        double* __restrict__ bufB, const double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nA) bufA[i] = v[listA[i]];
    else if (i < nA + nB) bufB[i - nA] = v[listB[i - nA]];
}
template <bool ADD>
__global__ void k_unpack2(const int32_t* __restrict__ listA, int64_t nA, const double* __restrict__ bufA, const int32_t* __restrict__ listB, int64_t nB,
                          const double* __restrict__ bufB, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nA) { if (ADD) v[listA[i]] += bufA[i]; else v[listA[i]] = bufA[i]; }
    else if (i < nA + nB) { if (ADD) v[listB[i - nA]] += bufB[i - nA]; else v[listB[i - nA]] = bufB[i - nA]; }
}
// out[q] = sum of partial[q*stride .. q*stride+count)   (q < nq), one block
__global__ void __launch_bounds__(BS) k_sumq(const CGScalars* __restrict__ sc, const double* __restrict__ partial, int count, int stride, int nq,
                                             double* __restrict__ out) {
    if (sc && sc->done) return;
    for (int q = 0; q < nq; ++q) {
        const double s = sumPartials(partial + (int64_t)q * stride, count);
        if (threadIdx.x == 0) out[q] = s;
        __syncthreads();
    }
}
__global__ void k_dscal0(CGScalars* sc, const double* __restrict__ red, double tol, int maxit) {
    const double s = red[0];
    sc->rsold = s; sc->rsold2[0] = s; sc->rsold2[1] = 0.; sc->rre = 0.; sc->iter = maxit; sc->maxit = maxit; sc->tol2 = tol * tol;
    sc->done = (s == 0.) ? 1 : 0;
    if (s == 0.) sc->iter = 0;
    sc->alpha = sc->beta = sc->pAp = sc->rr = sc->xx = sc->rz = 0.;
    sc->pend = 0; sc->pendIter = 0;
}
__global__ void k_invert_diag(double* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = d[i];
        d[i] = v != 0. ? 1. / v : 1.;
    }
}
struct SumPtrs { double* p[16]; int n; };
__global__ void k_sum_across(SumPtrs P, int count) {
    const int i = threadIdx.x;
    if (i >= count) return;
    double s = 0.;
    for (int q = 0; q < P.n; ++q) s += P.p[q][i];
    for (int q = 0; q < P.n; ++q) P.p[q][i] = s;
}
// faces this rank is responsible for in the output fields
__global__ void k_owned_faces(Grid g, int axis, Own own, const int32_t* __restrict__ faceRow, const int32_t* __restrict__ reg,
                              const int32_t* __restrict__ regionOwned, float* __restrict__ out) {
    const int3 d = g.dims(1 + axis);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int k = (int)(c / ((int64_t)d.x * d.y));
    bool mine;
    const int r = reg[c];
    if (faceRow[c] >= 0) mine = true;
    else if (r >= 0 && regionOwned) mine = regionOwned[r] != 0;
    else mine = own.sample(1 + axis, k);
    out[c] = mine ? 1.f : 0.f;
}

// ---- RCCL through dlopen (no link-time dependency; torch.distributed only bootstraps the unique id) ----
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /*ncclUniqueId by value*/ PsNcclUid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void*) = nullptr;
};
Rccl& rccl() {
    static Rccl R;
    if (!R.h) {
        R.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!R.h) R.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!R.h) throw Error(std::string("cannot load librccl: ") + dlerror());
        auto sym = [&](const char* n) { void* p = dlsym(R.h, n); if (!p) throw Error(std::string("librccl lacks ") + n); return p; };
        R.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        R.CommInitRank = (int (*)(void**, int, PsNcclUid, int))sym("ncclCommInitRank");
        R.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        R.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        R.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        R.GroupStart = (int (*)())sym("ncclGroupStart");
        R.GroupEnd = (int (*)())sym("ncclGroupEnd");
        R.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    }
    return R;
}
constexpr int NCCL_DOUBLE = 8;   // ncclFloat64
constexpr int NCCL_SUM = 0;
void ncclCheck(int rc, const char* what) { if (rc != 0) throw Error(std::string("RCCL failure in ") + what + " (code " + std::to_string(rc) + ")"); }

struct Dist {
    std::vector<ps_context*> R;   // the ranks living in this process (1 with RCCL, `world` for an in-process group)
    bool useRccl = false;

    // sizes: kind 0 = x exchange (send own layers, receive halo), kind 1 = y exchange (send halo contributions, receive for own)
    void transport(int kind) {
        if (useRccl) {
            ps_context* c = R[0];
            Rccl& L = rccl();
            const int64_t sLo = kind == 0 ? c->nLowOwn : c->nLowHalo, sUp = kind == 0 ? c->nUpOwn : c->nUpHalo;
            const int64_t rLo = kind == 0 ? c->nLowHalo : c->nLowOwn, rUp = kind == 0 ? c->nUpHalo : c->nUpOwn;
            ncclCheck(L.GroupStart(), "ncclGroupStart");
            if (c->slab.hasLower) {
                if (sLo) ncclCheck(L.Send(c->sendLo.p, (size_t)sLo, NCCL_DOUBLE, c->slab.rank - 1, c->rcclComm, c->stream), "ncclSend");
                if (rLo) ncclCheck(L.Recv(c->recvLo.p, (size_t)rLo, NCCL_DOUBLE, c->slab.rank - 1, c->rcclComm, c->stream), "ncclRecv");
            }
            if (c->slab.hasUpper) {
                if (sUp) ncclCheck(L.Send(c->sendUp.p, (size_t)sUp, NCCL_DOUBLE, c->slab.rank + 1, c->rcclComm, c->stream), "ncclSend");
                if (rUp) ncclCheck(L.Recv(c->recvUp.p, (size_t)rUp, NCCL_DOUBLE, c->slab.rank + 1, c->rcclComm, c->stream), "ncclRecv");
            }
            ncclCheck(L.GroupEnd(), "ncclGroupEnd");
            return;
        }
        for (size_t q = 0; q < R.size(); ++q) {   // in-process ranks share one stream: plain device copies
            ps_context* c = R[q];
            const int64_t sLo = kind == 0 ? c->nLowOwn : c->nLowHalo, sUp = kind == 0 ? c->nUpOwn : c->nUpHalo;
            if (c->slab.hasLower && sLo)
                HIP_CHECK(hipMemcpyAsync(R[q - 1]->recvUp.p, c->sendLo.p, (size_t)sLo * 8, hipMemcpyDeviceToDevice, c->stream));
            if (c->slab.hasUpper && sUp)
                HIP_CHECK(hipMemcpyAsync(R[q + 1]->recvLo.p, c->sendUp.p, (size_t)sUp * 8, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    void exchangeX(DevBuf<double> ps_context::*vec) {
        for (ps_context* c : R)
            if (c->nLowOwn + c->nUpOwn > 0)
                hipLaunchKernelGGL(k_pack2, dim3(gridFor(c->nLowOwn + c->nUpOwn, BS)), dim3(BS), 0, c->stream, c->listLowOwn.p, c->nLowOwn, c->sendLo.p,
                                   c->listUpOwn.p, c->nUpOwn, c->sendUp.p, (c->*vec).p);
        transport(0);
        for (ps_context* c : R)
            if (c->nLowHalo + c->nUpHalo > 0)
                hipLaunchKernelGGL(k_unpack2<false>, dim3(gridFor(c->nLowHalo + c->nUpHalo, BS)), dim3(BS), 0, c->stream, c->listLowHalo.p, c->nLowHalo,
                                   c->recvLo.p, c->listUpHalo.p, c->nUpHalo, c->recvUp.p, (c->*vec).p);
    }
    void exchangeAddY(DevBuf<double> ps_context::*vec) {
        for (ps_context* c : R)
            if (c->nLowHalo + c->nUpHalo > 0)
                hipLaunchKernelGGL(k_pack2, dim3(gridFor(c->nLowHalo + c->nUpHalo, BS)), dim3(BS), 0, c->stream, c->listLowHalo.p, c->nLowHalo, c->sendLo.p,
                                   c->listUpHalo.p, c->nUpHalo, c->sendUp.p, (c->*vec).p);
        transport(1);
        for (ps_context* c : R)   // contributions from below and from above land on disjoint DOFs
            if (c->nLowOwn + c->nUpOwn > 0)
                hipLaunchKernelGGL(k_unpack2<true>, dim3(gridFor(c->nLowOwn + c->nUpOwn, BS)), dim3(BS), 0, c->stream, c->listLowOwn.p, c->nLowOwn,
                                   c->recvLo.p, c->listUpOwn.p, c->nUpOwn, c->recvUp.p, (c->*vec).p);
    }
    void allreduce(int count) {
        if (useRccl) {
            ps_context* c = R[0];
            ncclCheck(rccl().AllReduce(c->redbuf.p, c->redbuf.p, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->rcclComm, c->stream), "ncclAllReduce");
            return;
        }
        if (R.size() == 1) return;
        SumPtrs P;
        P.n = (int)R.size();
        for (size_t q = 0; q < R.size(); ++q) P.p[q] = R[q]->redbuf.p;
        hipLaunchKernelGGL(k_sum_across, dim3(1), dim3(64), 0, R[0]->stream, P, count);
    }
    void syncAll() { for (ps_context* c : R) HIP_CHECK(hipStreamSynchronize(c->stream)); }

    // neighbours must agree on the exchange list lengths (same labels on both sides of a cut)
    void checkLists() {
        if (!useRccl) {
            for (size_t q = 0; q + 1 < R.size(); ++q)
                if (R[q]->nUpHalo != R[q + 1]->nLowOwn || R[q]->nUpOwn != R[q + 1]->nLowHalo)
                    throw Error("slab exchange lists disagree across the cut between ranks " + std::to_string(q) + " and " + std::to_string(q + 1));
            return;
        }
        ps_context* c = R[0];
        const double mine[4] = {(double)c->nLowOwn, (double)c->nLowHalo, (double)c->nUpOwn, (double)c->nUpHalo};
        HIP_CHECK(hipMemcpyAsync(c->sendLo.p, mine, 16, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->sendUp.p, mine + 2, 16, hipMemcpyHostToDevice, c->stream));
        Rccl& L = rccl();
        ncclCheck(L.GroupStart(), "ncclGroupStart");
        if (c->slab.hasLower) { ncclCheck(L.Send(c->sendLo.p, 2, NCCL_DOUBLE, c->slab.rank - 1, c->rcclComm, c->stream), "send"); ncclCheck(L.Recv(c->recvLo.p, 2, NCCL_DOUBLE, c->slab.rank - 1, c->rcclComm, c->stream), "recv"); }
        if (c->slab.hasUpper) { ncclCheck(L.Send(c->sendUp.p, 2, NCCL_DOUBLE, c->slab.rank + 1, c->rcclComm, c->stream), "send"); ncclCheck(L.Recv(c->recvUp.p, 2, NCCL_DOUBLE, c->slab.rank + 1, c->rcclComm, c->stream), "recv"); }
        ncclCheck(L.GroupEnd(), "ncclGroupEnd");
        double lo[2] = {0, 0}, up[2] = {0, 0};
        if (c->slab.hasLower) HIP_CHECK(hipMemcpyAsync(lo, c->recvLo.p, 16, hipMemcpyDeviceToHost, c->stream));
        if (c->slab.hasUpper) HIP_CHECK(hipMemcpyAsync(up, c->recvUp.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        // the lower rank sent me its (nUpOwn, nUpHalo); the upper rank its (nLowOwn, nLowHalo)
        if (c->slab.hasLower && ((int64_t)lo[0] != c->nLowHalo || (int64_t)lo[1] != c->nLowOwn)) throw Error("slab exchange lists disagree with the lower neighbour");
        if (c->slab.hasUpper && ((int64_t)up[0] != c->nUpHalo || (int64_t)up[1] != c->nUpOwn)) throw Error("slab exchange lists disagree with the upper neighbour");
    }

    // everything after the per-rank local setup: finish b and the Jacobi diagonal across the cuts
    void finishSetup() {
        for (ps_context* c : R) c->redbuf.alloc(8);
        checkLists();
        exchangeAddY(&ps_context::b);
        const bool jac = R[0]->P.preconditioner == PS_PRE_DIAGONAL;
        if (jac) {
            exchangeAddY(&ps_context::dinv);
            for (ps_context* c : R) {
                const int64_t n = c->ownHi - c->ownLo;
                if (n > 0) {
                    hipLaunchKernelGGL(k_invert_diag, dim3(dotBlocks(n)), dim3(BS), 0, c->stream, c->dinv.p + c->ownLo, n);
                    hipLaunchKernelGGL(k_to_float, dim3(dotBlocks(n)), dim3(BS), 0, c->stream, c->dinv.p + c->ownLo, c->dinvF.p + c->ownLo, n);
                }
            }
        }
    }

    int solve() {
        ps_context* c0 = R[0];
        const int maxit = c0->P.maxSolverIterations;
        const double tol = c0->P.tolerance;
        const bool jac = c0->P.preconditioner == PS_PRE_DIAGONAL;
        if (c0->P.solverType != PS_PCG_MATRIX_VECTOR_PRODUCTS) { c0->err = "Unsupported Solver."; return PS_UNSUPPORTED_SOLVER; }
        struct Loc { int64_t n, lo; int vb, stBlocks; const float* dv; CGScalars* sc; Launch L; };
        std::vector<Loc> loc(R.size());
        for (size_t q = 0; q < R.size(); ++q) {
            ps_context* c = R[q];
            Loc& l = loc[q];
            l.lo = c->ownLo; l.n = c->ownHi - c->ownLo;
            l.vb = dotBlocks(std::max<int64_t>(l.n, 1));
            l.dv = jac ? c->dinvF.p + l.lo : nullptr;
            l.sc = c->scal.p;
            l.L = mk(c, &l.sc->done);
            l.stBlocks = l.L.stBlocks();
            c->usedBiCGStab = 0;
        }
        // r = b, x = 0, p = z on the owned range; rsold = sum over ranks of r.z
        for (size_t q = 0; q < R.size(); ++q) {
            ps_context* c = R[q];
            Loc& l = loc[q];
            HIP_CHECK(hipMemsetAsync(c->pvec.p, 0, (size_t)std::max<int64_t>(c->nSystem, 1) * 8, c->stream));
            HIP_CHECK(hipMemsetAsync(c->dotPartials3.p, 0, VGRID * sizeof(double), c->stream));
            hipLaunchKernelGGL(k_cg_init_f, dim3(l.vb), dim3(BS), 0, c->stream, c->b.p + l.lo, l.dv, c->x.p + l.lo, c->r.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials.p);
            hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)nullptr, c->dotPartials.p, l.vb, 0, 1, c->redbuf.p);
        }
        allreduce(1);
        for (size_t q = 0; q < R.size(); ++q)
            hipLaunchKernelGGL(k_dscal0, dim3(1), dim3(1), 0, R[q]->stream, loc[q].sc, R[q]->redbuf.p, tol, maxit);
        CGScalars h{};
        const int batch = 25;
        int it = 0;
        bool finished = false;
        while (it < maxit && !finished) {
            const int upto = std::min(maxit, it + batch);
            for (; it < upto; ++it) {
                exchangeX(&ps_context::pvec);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    l.L.spmvS(0, c->pvec.p, c->ts.p);
                    l.L.tiles(0, c->ts.p);
                    l.L.spmvSt(0, c->ts.p, c->pvec.p, nullptr, c->Ap.p, c->dotPartials.p);
                    if (l.stBlocks <= 8192) {
                        hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartials.p, l.stBlocks, 0, 1, c->redbuf.p);
                    } else {
                        hipLaunchKernelGGL(k_reduce_partials, dim3(RED_BLOCKS), dim3(BS), 0, c->stream, l.sc, c->dotPartials.p, l.stBlocks, c->dotPartials2.p);
                        hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartials2.p, RED_BLOCKS, 0, 1, c->redbuf.p);
                    }
                    // ||x||^2 of the x updated last iteration rides along (stop test of the previous iteration, see k_cg_update_r)
                    hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartials3.p, l.vb, 0, 1, c->redbuf.p + 1);
                }
                exchangeAddY(&ps_context::Ap);
                allreduce(2);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    hipLaunchKernelGGL(k_cg_update_r, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, (const double*)nullptr, 0,
                                       (const double*)nullptr, 0, it, c->Ap.p + l.lo, l.dv, c->r.p + l.lo, l.n, c->dotPartialsR.p);
                    hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, c->stream, (const CGScalars*)l.sc, c->dotPartialsR.p, l.vb, l.vb, 2, c->redbuf.p);
                }
                allreduce(2);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    hipLaunchKernelGGL(k_cg_update_xp, dim3(l.vb), dim3(BS), 0, c->stream, l.sc, (const double*)c->redbuf.p, (const double*)nullptr, 0,
                                       jac ? 1 : 0, it, c->r.p + l.lo, l.dv, c->x.p + l.lo, c->pvec.p + l.lo, l.n, c->dotPartials3.p);
                }
            }
            for (size_t q = 0; q < R.size(); ++q)   // the stop test of the batch's last iteration
                hipLaunchKernelGGL(k_sumq, dim3(1), dim3(BS), 0, R[q]->stream, (const CGScalars*)loc[q].sc, R[q]->dotPartials3.p, loc[q].vb, 0, 1, R[q]->redbuf.p);
            allreduce(1);
            for (size_t q = 0; q < R.size(); ++q)
                hipLaunchKernelGGL(k_cg_check, dim3(1), dim3(BS), 0, R[q]->stream, loc[q].sc, (const double*)R[q]->redbuf.p, (const double*)nullptr, 0, it - 1);
            HIP_CHECK(hipMemcpyAsync(&h, loc[0].sc, sizeof(h), hipMemcpyDeviceToHost, c0->stream));
            syncAll();
            if (h.done) finished = true;
        }
        int iters = h.done ? h.iter : maxit;
        double err = std::sqrt(h.rre);
        if (iters == maxit) {
            // bicgstab_external_matrix_A (pcg.h:134-200), restarted from zero (Solver.cpp:784-799): the host-driven loop
            // of ps_context::solve() with distributed applies and dots.  Rare path.
            for (ps_context* c : R) {
                c->usedBiCGStab = 1;
                const size_t nl = (size_t)std::max<int64_t>(c->nSystem, 1);
                c->tmp1.alloc(nl); c->tmp2.alloc(nl); c->tmp3.alloc(nl); c->tmp4.alloc(nl); c->tmp5.alloc(nl);
            }
            using Vec = DevBuf<double> ps_context::*;
            const Vec X = &ps_context::x, Rv = &ps_context::r, Pv = &ps_context::pvec, B = &ps_context::b, H = &ps_context::Ap,
                      Rhat = &ps_context::tmp1, V = &ps_context::tmp2, S = &ps_context::tmp3, T = &ps_context::tmp4, E = &ps_context::tmp5;
            auto apply = [&](Vec in, Vec out) {
                exchangeX(in);
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Launch L = mk(c, nullptr);
                    L.spmvS(0, (c->*in).p, c->ts.p);
                    L.tiles(0, c->ts.p);
                    L.spmvSt(0, c->ts.p, (c->*in).p, nullptr, (c->*out).p, c->dotPartials.p);
                }
                exchangeAddY(out);
            };
            auto dot = [&](Vec a, Vec bvec) {
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    hipLaunchKernelGGL(k_dot, dim3(l.vb), dim3(BS), 0, c->stream, (c->*a).p + l.lo, (c->*bvec).p + l.lo, l.n, c->dotPartials.p);
                    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(BS), 0, c->stream, c->dotPartials.p, l.vb, c->redbuf.p);
                }
                allreduce(1);
                double out = 0.;
                HIP_CHECK(hipMemcpyAsync(&out, c0->redbuf.p, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
                syncAll();
                return out;
            };
            auto lin = [&](Vec out, double ca, Vec a, double cb, Vec bvec, double cc, Vec c3) {   // out = ca a + cb b + cc c on the owned range
                for (size_t q = 0; q < R.size(); ++q) {
                    ps_context* c = R[q];
                    Loc& l = loc[q];
                    if (l.n <= 0) continue;
                    hipLaunchKernelGGL(k_lin, dim3(l.vb), dim3(BS), 0, c->stream, (c->*out).p + l.lo, ca, (const double*)(c->*a).p + l.lo, cb,
                                       bvec ? (const double*)(c->*bvec).p + l.lo : (const double*)nullptr, cc,
                                       c3 ? (const double*)(c->*c3).p + l.lo : (const double*)nullptr, l.n);
                }
            };
            auto zero = [&](Vec v) { for (ps_context* c : R) HIP_CHECK(hipMemsetAsync((c->*v).p, 0, (size_t)std::max<int64_t>(c->nSystem, 1) * 8, c->stream)); };
            zero(X);
            lin(Rv, 1., B, 0., nullptr, 0., nullptr);            // r = b - A*0
            lin(Rhat, 1., Rv, 0., nullptr, 0., nullptr);
            zero(Pv); zero(V);
            double rhoCurr = 1., rhoOld = 1., alpha = 1., beta = 0., omega = 1., rre = 0.;
            iters = maxit;
            for (int i = 0; i < maxit; ++i) {
                rhoOld = rhoCurr;
                rhoCurr = dot(Rhat, Rv);
                beta = (rhoCurr / rhoOld) * (alpha / omega);
                lin(Pv, 1., Rv, beta, Pv, -beta * omega, V);      // p = r + beta (p - omega v)
                apply(Pv, V);
                alpha = rhoCurr / dot(Rhat, V);
                lin(H, 1., X, alpha, Pv, 0., nullptr);             // h = x + alpha p
                lin(S, 1., Rv, -alpha, V, 0., nullptr);            // s = r - alpha v
                apply(S, T);
                omega = dot(T, S) / dot(T, T);
                lin(X, 1., H, omega, S, 0., nullptr);              // x = h + omega s
                const double xmag = std::sqrt(dot(X, X));
                apply(X, E);
                lin(E, 1., B, -1., E, 0., nullptr);                // err = b - A x
                const double rsnew = dot(E, E);
                rre = rsnew;
                if (std::sqrt(rsnew) / xmag < rre) rre = std::sqrt(rsnew) / xmag;
                if (rre < tol) { iters = i; break; }
                lin(Rv, 1., S, -omega, T, 0., nullptr);            // r = s - omega t
            }
            err = rre;
        }
        for (ps_context* c : R) { c->solveIterations = iters; c->solveError = err; }
        return iters == maxit ? PS_NOCONVERGE : PS_SUCCESS;
    }

    void recoverAndWriteBack(bool apply) {
        if (apply) exchangeX(&ps_context::x);
        for (ps_context* c : R) {
            c->buildValidFaces();
            if (apply) { c->recoverVelocityFromPressureStress(); c->applySolutionToVelocity(); }
            else for (int a = 0; a < 3; ++a)
                HIP_CHECK(hipMemcpyAsync(c->velOut[a].p, c->vel[a].p, (size_t)c->g.count(1 + a) * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            for (int a = 0; a < 3; ++a) {
                c->ownedFace[a].alloc((size_t)c->g.count(1 + a));
                hipLaunchKernelGGL(k_owned_faces, dim3(gridFor(c->g.count(1 + a), BS)), dim3(BS), 0, c->stream, c->g, a, c->own(), c->faceRow[a].p,
                                   c->reducedIdx[1 + a].p, (c->slabEnabled && c->regionCount > 0) ? c->regionOwned.p : (const int32_t*)nullptr,
                                   c->ownedFace[a].p);
            }
        }
        syncAll();
    }
};

int distStep(Dist& D, ps_stats* stats) {
    const auto w0 = std::chrono::high_resolution_clock::now();
    for (ps_context* c : D.R) c->setup(nullptr);
    D.finishSetup();
    D.syncAll();
    const auto w1 = std::chrono::high_resolution_clock::now();
    int result = PS_INCOMPLETE;
    ps_context* c0 = D.R[0];
    if (c0->P.doSolve) result = D.solve();
    D.syncAll();
    const auto w2 = std::chrono::high_resolution_clock::now();
    const bool apply = c0->P.doSolve && result != PS_UNSUPPORTED_SOLVER && (result == PS_SUCCESS || c0->P.keepNonConvergedResults);
    D.recoverAndWriteBack(apply);
    for (ps_context* c : D.R) {
        c->lastStats.solveData[0] = c->solveError;
        c->lastStats.solveData[1] = c->solveIterations;
        c->lastStats.solveData[3] = std::chrono::duration<double, std::milli>(w2 - w1).count();
        c->lastStats.solveData[5] = std::chrono::duration<double, std::milli>(w1 - w0).count();
        c->lastStats.stage_ms[PS_STAGE_SOLVE] = c->lastStats.solveData[3];
        c->lastStats.result = result;
        c->lastStats.usedBiCGStab = c->usedBiCGStab;
        c->isSolved = true;
        c->registerArrays();
    }
    if (stats) *stats = c0->lastStats;
    return result;
}

}  // namespace

struct ps_group {
    std::vector<ps_context*> ranks;
    hipStream_t stream = nullptr;
};

int ps_dist_step_single(ps_context* c, ps_stats* stats) {   // one process per GPU, RCCL
    Dist D;
    D.R.push_back(c);
    D.useRccl = true;
    return distStep(D, stats);
}

extern "C" {

int32_t ps_set_slab(ps_context* c, const ps_slab* slab) {
    if (!c || !slab) return PS_FAILED;
    try {
        if (!c->uploaded) throw Error("ps_upload_fields first");
        const int L = 16;
        if (slab->zLoOwned % L || slab->zHiOwned % L) {
            if (!(slab->zHiOwned == c->g.nz && !slab->hasUpper && slab->zLoOwned % L == 0)) throw Error("slab cuts must be multiples of 16");
        }
        if (c->P.doReducedRegions && c->P.doTile && (slab->zLoOwned % c->P.tileSize || (slab->hasUpper && slab->zHiOwned % c->P.tileSize)))
            throw Error("slab cuts must be multiples of the tile size");
        if (c->P.doReducedRegions && !c->P.doTile && slab->world > 1) throw Error("the slab decomposition needs doTile (tile-local regions)");
        if (slab->zLoOwned < 0 || slab->zHiOwned > c->g.nz || slab->zLoOwned >= slab->zHiOwned) throw Error("bad slab range");
        if ((slab->hasLower && slab->zLoOwned < 16) || (slab->hasUpper && c->g.nz - slab->zHiOwned < 16)) throw Error("a halo of at least 16 layers is required next to a cut");
        if (!slab->hasLower && slab->zLoOwned != 0) throw Error("without a lower neighbour the slab must start at layer 0");
        if (!slab->hasUpper && slab->zHiOwned != c->g.nz) throw Error("without an upper neighbour the slab must end at the top layer");
        c->slab = *slab;
        c->slabEnabled = slab->world > 1;
        c->isSetup = false;
        return PS_SUCCESS;
    } catch (const ps::Error& e) { c->err = e.msg; return PS_FAILED; }
}

int32_t ps_comm_unique_id(void* id128) {
    try { ncclCheck(rccl().GetUniqueId(id128), "ncclGetUniqueId"); return PS_SUCCESS; } catch (const ps::Error& e) { std::fprintf(stderr, "%s\n", e.msg.c_str()); return PS_FAILED; }
}
int32_t ps_comm_init_rccl(ps_context* c, const void* id128, int32_t rank, int32_t world) {
    if (!c || !id128) return PS_FAILED;
    try {
        HIP_CHECK(hipSetDevice(c->device));
        PsNcclUid id;
        std::memcpy(&id, id128, sizeof(id));
        void* comm = nullptr;
        ncclCheck(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
        c->rcclComm = comm;
        return PS_SUCCESS;
    } catch (const ps::Error& e) { c->err = e.msg; return PS_FAILED; }
}

// exercises the RCCL entry points used by the distributed solve (all-reduce, grouped send/recv to self) on this
// rank's communicator; returns PS_SUCCESS when the values come back right.
int32_t ps_comm_selftest(ps_context* c) {
    if (!c) return PS_FAILED;
    try {
        if (!c->rcclComm) throw Error("no communicator");
        HIP_CHECK(hipSetDevice(c->device));
        c->redbuf.alloc(8); c->sendLo.alloc(8); c->recvLo.alloc(8);
        const double v[4] = {1.5, -2.0, 3.25, 4.0};
        HIP_CHECK(hipMemcpyAsync(c->redbuf.p, v, 32, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->sendLo.p, v, 32, hipMemcpyHostToDevice, c->stream));
        Rccl& L = rccl();
        ncclCheck(L.AllReduce(c->redbuf.p, c->redbuf.p, 3, NCCL_DOUBLE, NCCL_SUM, c->rcclComm, c->stream), "ncclAllReduce");
        ncclCheck(L.GroupStart(), "ncclGroupStart");
        ncclCheck(L.Send(c->sendLo.p, 4, NCCL_DOUBLE, c->slab.rank, c->rcclComm, c->stream), "ncclSend");
        ncclCheck(L.Recv(c->recvLo.p, 4, NCCL_DOUBLE, c->slab.rank, c->rcclComm, c->stream), "ncclRecv");
        ncclCheck(L.GroupEnd(), "ncclGroupEnd");
        double a[4], b[4];
        HIP_CHECK(hipMemcpyAsync(a, c->redbuf.p, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpyAsync(b, c->recvLo.p, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 4; ++i) if (b[i] != v[i]) throw Error("send/recv self-test mismatch");
        if (a[3] != v[3]) throw Error("all-reduce touched elements beyond count");
        return PS_SUCCESS;   // a[0..2] = world * v (checked by the caller, who knows the world size)
    } catch (const ps::Error& e) { c->err = e.msg; return PS_FAILED; }
}

ps_group* ps_group_create(int32_t device, int32_t world) {
    if (world < 1 || world > 16) return nullptr;
    ps_group* g = new ps_group();
    for (int q = 0; q < world; ++q) {
        ps_context* c = ps_context_create(device);
        if (!c) { for (ps_context* d : g->ranks) ps_context_destroy(d); delete g; return nullptr; }
        if (q == 0) g->stream = c->stream;
        else { (void)hipStreamDestroy(c->stream); c->stream = g->stream; c->ownsStream = false; }   // one shared stream: launches are ordered
        g->ranks.push_back(c);
    }
    return g;
}
void ps_group_destroy(ps_group* g) {
    if (!g) return;
    for (size_t q = g->ranks.size(); q-- > 0;) ps_context_destroy(g->ranks[q]);
    delete g;
}
ps_context* ps_group_rank(ps_group* g, int32_t rank) { return (g && rank >= 0 && rank < (int)g->ranks.size()) ? g->ranks[(size_t)rank] : nullptr; }
int32_t ps_group_step(ps_group* g, ps_stats* stats) {
    if (!g || g->ranks.empty()) return PS_FAILED;
    try {
        Dist D;
        D.R = g->ranks;
        D.useRccl = false;
        return distStep(D, stats);
    } catch (const ps::Error& e) { g->ranks[0]->err = e.msg; return PS_FAILED; }
}

}  // extern "C"

