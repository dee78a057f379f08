// PCG vector / scalar kernels, BiCGStab helpers, Jacobi diagonal, velocity recovery and write-back.
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once

// ---- CG vector kernels ---------------------------------------------------------------------------
// Cache policy of the streams these kernels touch once per iteration (compile-time, scripts/build_variant.sh builds A/B copies).
// Measured at 256^3: x, Ap and r non-temporal (read / written once per iteration, next use a whole iteration away) take
// k_cg_update_xp from 0.345 to 0.30 ms and the step from 1378 to 1319 ms; p stays default-policy (the next S gathers it).
#ifndef PS_VEC_NT_X
#define PS_VEC_NT_X 1      // x: read and written only by k_cg_update_xp
#endif
#ifndef PS_VEC_NT_AP
#define PS_VEC_NT_AP 1     // Ap: written by St, read once by k_cg_update_r
#endif
#ifndef PS_VEC_NT_R
#define PS_VEC_NT_R 1      // r in k_cg_update_r (read + written)
#endif
#ifndef PS_VEC_NT_RX
#define PS_VEC_NT_RX 1     // r as read by k_cg_update_xp
#endif
#ifndef PS_VEC_NT_P
#define PS_VEC_NT_P 0      // p as written by k_cg_update_xp (gathered by the next S)
#endif
#ifndef PS_VEC_NT_PL
#define PS_VEC_NT_PL 0     // p as read by k_cg_update_xp
#endif
#ifndef PS_VEC_NT_U
#define PS_VEC_NT_U 1      // the uInv codes as read by k_cg_update_xp_u
#endif
#ifndef PS_VEC_NT_D
#define PS_VEC_NT_D 1      // the stored Jacobi diagonal (read by both step kernels, half an iteration apart)
#endif
// two consecutive entries of the stored Jacobi diagonal (ps_common.hpp: diag_t) as doubles; i2 = index of the pair
__device__ inline double2 ldDiag2(const diag_t* __restrict__ d, int64_t i2, bool nt) {
#ifdef PS_DIAG_FP32
    typedef float psf2 __attribute__((ext_vector_type(2)));
    const psf2* q = reinterpret_cast<const psf2*>(d) + i2;
    const psf2 v = nt ? __builtin_nontemporal_load(q) : *q;
    return make_double2((double)v.x, (double)v.y);
#else
    const uint32_t* q = reinterpret_cast<const uint32_t*>(d) + i2;
    const uint32_t v = nt ? __builtin_nontemporal_load(q) : *q;
    return make_double2((double)__builtin_bit_cast(float, v << 16), (double)__builtin_bit_cast(float, v & 0xFFFF0000u));
#endif
}
constexpr uintptr_t DIAG_PAIR_MASK = 2 * sizeof(diag_t) - 1;
typedef double psd2 __attribute__((ext_vector_type(2)));
__device__ inline double2 ldD2(const double2* p, bool nt) {
    if (!nt) return *p;
    const psd2 v = __builtin_nontemporal_load(reinterpret_cast<const psd2*>(p));
    return make_double2(v.x, v.y);
}
__device__ inline void stD2(double2* p, double2 v, bool nt) {
    if (!nt) { *p = v; return; }
    psd2 q; q.x = v.x; q.y = v.y;
    __builtin_nontemporal_store(q, reinterpret_cast<psd2*>(p));
}
__global__ void k_scale_rows(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}
// r = b; x = 0; z = pre(r); p = z; partial rsold = r.z
__global__ void __launch_bounds__(BS) k_cg_init(const double* __restrict__ b, const double* __restrict__ dinv, double* __restrict__ x,
                                                double* __restrict__ r, double* __restrict__ p, int64_t n, double* __restrict__ partial) {
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double rv = b[i];
        const double z = dinv ? dinv[i] * rv : rv;
        x[i] = 0.; r[i] = rv; p[i] = z;
        acc += rv * z;
    }
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// r = b; x = 0; z = pre(r); p = z; partial rsold = r.z — with the stored Jacobi diagonal of the main PCG path (diag_t)
__global__ void __launch_bounds__(BS) k_cg_init_f(const double* __restrict__ b, const diag_t* __restrict__ dinv, double* __restrict__ x,
                                                  double* __restrict__ r, double* __restrict__ p, int64_t n, double* __restrict__ partial) {
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double rv = b[i];
        const double z = dinv ? diagValue(dinv[i]) * rv : rv;
        x[i] = 0.; r[i] = rv; p[i] = z;
        acc += rv * z;
    }
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void k_to_diag(const double* __restrict__ a, diag_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = diagStore(a[i]);
}
__device__ inline double sumLocal(const double* __restrict__ partial, int count) {   // this thread's share (fixed stride order)
    double acc = 0.;
    for (int i = threadIdx.x; i < count; i += BS) acc += partial[i];
    return acc;
}
__device__ inline double sumPartials(const double* __restrict__ partial, int count) {
    double acc = 0.;
    for (int i = threadIdx.x; i < count; i += BS) acc += partial[i];
    return blockReduceSum(acc);
}
__global__ void __launch_bounds__(BS) k_cg_scal0(CGScalars* sc, const double* __restrict__ partial, int count, double tol, int maxit, int vecNT) {
    const double s = sumPartials(partial, count);
    if (threadIdx.x == 0) {
        sc->rsold = s; sc->rsold2[0] = s; sc->rsold2[1] = 0.; sc->rre = 0.; sc->iter = maxit; sc->maxit = maxit; sc->tol2 = tol * tol;
        sc->done = (s == 0.) ? 1 : 0;      // deviation: b == 0 -> return at once (reference divides 0/0, pcg.h:314)
        if (s == 0.) sc->iter = 0;
        sc->alpha = sc->beta = sc->pAp = sc->rr = sc->xx = sc->rz = 0.;
        sc->pend = 0; sc->pendIter = 0; sc->vecNT = vecNT;
    }
}
// stage A of the p.Ap reduction: RED_BLOCKS blocks each sum a contiguous slice of the SpMV block partials
constexpr int RED_BLOCKS = 256;
__global__ void __launch_bounds__(BS) k_reduce_partials(const CGScalars* __restrict__ sc, const double* __restrict__ partial, int count,
                                                        double* __restrict__ out) {
    if (sc->done) return;
    const int per = (count + RED_BLOCKS - 1) / RED_BLOCKS;
    const int lo = blockIdx.x * per, hi = min(lo + per, count);
    double acc = 0.;
    for (int i = lo + threadIdx.x; i < hi; i += BS) acc += partial[i];
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void __launch_bounds__(BS) k_cg_scal1(CGScalars* sc, const double* __restrict__ partial, int count) {
    if (sc->done) return;
    const double s = sumPartials(partial, count);
    if (threadIdx.x == 0) { sc->pAp = s; sc->alpha = sc->rsold / s; }   // pcg.h:314
}
// x += alpha p ; r -= alpha Ap ; partials of r.r, x.x, r.z   (pcg.h:315-319,331).  16-byte (double2) accesses.
__global__ void __launch_bounds__(BS) k_cg_update_xr(const CGScalars* __restrict__ sc, const double* __restrict__ p, const double* __restrict__ Ap,
                                                     const double* __restrict__ dinv, double* __restrict__ x, double* __restrict__ r, int64_t n,
                                                     double* __restrict__ partial) {
    if (sc->done) return;
    const double alpha = sc->alpha;
    double arr = 0., axx = 0., arz = 0.;
    const bool vec = ((((uintptr_t)p | (uintptr_t)Ap | (uintptr_t)x | (uintptr_t)r | (uintptr_t)dinv) & 15) == 0);
    const int64_t n2 = vec ? n / 2 : 0;
    const double2* p2 = (const double2*)p; const double2* A2 = (const double2*)Ap; const double2* d2 = (const double2*)dinv;
    double2* x2 = (double2*)x; double2* r2 = (double2*)r;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BS) {
        const double2 pv = p2[i], av = A2[i];
        double2 xv = x2[i], rv = r2[i];
        xv.x = xv.x + alpha * pv.x; xv.y = xv.y + alpha * pv.y;
        rv.x = rv.x - alpha * av.x; rv.y = rv.y - alpha * av.y;
        x2[i] = xv; r2[i] = rv;
        arr += rv.x * rv.x; arr += rv.y * rv.y;
        axx += xv.x * xv.x; axx += xv.y * xv.y;
        if (dinv) { const double2 dv = d2[i]; arz += rv.x * (dv.x * rv.x); arz += rv.y * (dv.y * rv.y); }
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double xv = x[i] + alpha * p[i];
        const double rv = r[i] - alpha * Ap[i];
        x[i] = xv; r[i] = rv;
        arr += rv * rv; axx += xv * xv;
        if (dinv) arz += rv * (dinv[i] * rv);
    }
    const double s0 = blockReduceSum(arr), s1 = blockReduceSum(axx), s2 = dinv ? blockReduceSum(arz) : 0.;
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s0;
        partial[gridDim.x + blockIdx.x] = s1;
        partial[2 * gridDim.x + blockIdx.x] = s2;
    }
}
__global__ void __launch_bounds__(BS) k_cg_scal2(CGScalars* sc, const double* __restrict__ partial, int count, int jacobi, int iterIndex) {
    if (sc->done) return;
    const double rr = sumPartials(partial, count);
    const double xx = sumPartials(partial + count, count);
    const double rz = jacobi ? sumPartials(partial + 2 * count, count) : rr;
    if (threadIdx.x == 0) {
        sc->rr = rr; sc->xx = xx; sc->rz = rz;
        double rre = rr;                              // pcg.h:319-325
        if (rr / xx < rre) rre = rr / xx;
        sc->rre = rre;
        if (rre < sc->tol2) { sc->done = 1; sc->iter = iterIndex; }
        else { sc->beta = rz / sc->rsold; sc->rsold = rz; }   // pcg.h:331-335
    }
}
__global__ void __launch_bounds__(BS) k_cg_update_p(const CGScalars* __restrict__ sc, const double* __restrict__ r, const double* __restrict__ dinv,
                                                    double* __restrict__ p, int64_t n) {
    if (sc->done) return;
    const double beta = sc->beta;
    const bool vec = ((((uintptr_t)p | (uintptr_t)r | (uintptr_t)dinv) & 15) == 0);
    const int64_t n2 = vec ? n / 2 : 0;
    const double2* r2 = (const double2*)r; const double2* d2 = (const double2*)dinv;
    double2* p2 = (double2*)p;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BS) {
        double2 z = r2[i];
        if (dinv) { const double2 dv = d2[i]; z.x = dv.x * z.x; z.y = dv.y * z.y; }
        double2 pv = p2[i];
        pv.x = z.x + beta * pv.x; pv.y = z.y + beta * pv.y;
        p2[i] = pv;
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double z = dinv ? dinv[i] * r[i] : r[i];
        p[i] = z + beta * p[i];
    }
}

// ---- PCG step: x update deferred into the p update, scalar reductions folded into the vector kernels ---------------
// pcg.h:311-335 updates x and r together, tests min(rr, rr/xx) < tol^2, then forms beta and the new p: 11 vector passes
// and (here) two one-block scalar kernels.  This step is 10 passes and 2 launches:
//   k_cg_update_r :  [stop test of the previous iteration]  alpha = rsold / p.Ap ;  r -= alpha Ap ;  partials r.r, r.z
//   k_cg_update_xp:  beta = r.z / rsold ;  x += alpha p ;  p = z + beta p (p read once for both) ;  partials x.x
// Every block sums the (<= 4096 + 1024) partials of the preceding kernel itself — same order in every block, so all
// blocks hold bit-identical scalars — and block 0 records them for the host and the next kernel; rsold is double-buffered
// by iteration parity so no block reads a scalar another block of the same launch writes.
// The stop test of iteration k — same rr, xx of the updated x, same iteration index as the reference — is evaluated at
// the start of iteration k+1 (or by k_cg_check before the host polls); when it fires every later kernel is a no-op and
// x already holds the iterate the reference returns.  Cost: one unused p update and one unused operator apply.
// With `red` (distributed solve) the sums come all-reduced from the ranks: red = {p.Ap, x.x} resp. {r.r, r.z}.
__device__ inline bool stopTest(CGScalars* sc, double xx, int iterIndex, bool writer) {
    const double rr = sc->rr;
    double rre = rr;                                   // pcg.h:319-325
    if (rr / xx < rre) rre = rr / xx;
    const bool fire = rre < sc->tol2;
    if (writer) { sc->xx = xx; sc->rre = rre; if (fire) { sc->done = 1; sc->iter = iterIndex; } }
    return fire;
}
__global__ void __launch_bounds__(BS) k_cg_check(CGScalars* sc, const double* __restrict__ red, const double* __restrict__ xxPartial, int vb, int lastIter) {
    if (sc->done) return;
    const double xx = red ? red[0] : blockSumAll(sumLocal(xxPartial, vb));
    stopTest(sc, xx, lastIter, threadIdx.x == 0);
}
// [stop test of iteration it-1] ; alpha ; r -= alpha Ap ; partials of r.r and r.z
__global__ void __launch_bounds__(BS) k_cg_update_r(CGScalars* sc, const double* __restrict__ red, const double* __restrict__ pApPartial, int pApCount,
                                                    const double* __restrict__ xxPartial, int xxCount, int it, const double* __restrict__ Ap,
                                                    const diag_t* __restrict__ dinv, double* __restrict__ r, int64_t n, double* __restrict__ partial) {
    if (sc->done) return;
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    double pAp, xx = 0.;
    if (red) { pAp = red[0]; xx = red[1]; }
    else {
        if (it > 0) xx = blockSumAll(sumLocal(xxPartial, xxCount));
        pAp = blockSumAll(sumLocal(pApPartial, pApCount));
    }
    if (it > 0 && stopTest(sc, xx, it - 1, writer)) return;           // same verdict in every block
    const double alpha = sc->rsold2[it & 1] / pAp;                      // pcg.h:314
    if (writer) { sc->pAp = pAp; sc->alpha = alpha; }
    double arr = 0., arz = 0.;
    const bool nt = sc->vecNT != 0;
    const bool vec = ((((uintptr_t)Ap | (uintptr_t)r) & 15) == 0) && (((uintptr_t)dinv & DIAG_PAIR_MASK) == 0);
    const int64_t n2 = vec ? n / 2 : 0;
    const double2* A2 = (const double2*)Ap;
    double2* r2 = (double2*)r;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BS) {
        const double2 av = ldD2(A2 + i, nt && PS_VEC_NT_AP);
        double2 rv = ldD2(r2 + i, nt && PS_VEC_NT_R);
        rv.x = rv.x - alpha * av.x; rv.y = rv.y - alpha * av.y;
        stD2(r2 + i, rv, nt && PS_VEC_NT_R);
        arr += rv.x * rv.x; arr += rv.y * rv.y;
        if (dinv) { const double2 dv = ldDiag2(dinv, i, nt && PS_VEC_NT_D); arz += rv.x * (dv.x * rv.x); arz += rv.y * (dv.y * rv.y); }
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double rv = r[i] - alpha * Ap[i];
        r[i] = rv;
        arr += rv * rv;
        if (dinv) arz += rv * (diagValue(dinv[i]) * rv);
    }
    const double s0 = blockReduceSum(arr), s2 = dinv ? blockReduceSum(arz) : 0.;
    if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[gridDim.x + blockIdx.x] = s2; }
}
// beta ; x += alpha p ; p = z + beta p (z = D^-1 r) ; partials of x.x
// UPP (fused residual update, FusedR in ps_kernels_spmv.hpp): also the partials of sum_j uInv_j p_j^2 of the NEW p — the diagonal
// share of the next p . A p — from the coded diagonal (1 B per entry + 256-entry table in LDS) or the fp64 one.
template <bool UPP>
__device__ inline void cgUpdateXp(CGScalars* sc, const double* __restrict__ red, const double* __restrict__ rPartial, int rCount, int jacobi,
                                  int it, const double* __restrict__ r, const diag_t* __restrict__ dinv, double* __restrict__ x,
                                  double* __restrict__ p, int64_t n, double* __restrict__ partial,
                                  const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, const double* __restrict__ uInv, double* __restrict__ uPart) {
    if (sc->done) return;
    __shared__ double dict[UPP ? 256 : 1];
    if (UPP && uCode) dict[threadIdx.x] = uDict[threadIdx.x];   // visible after the barriers of blockSumAll below
    double rr, rz;
    if (red) { rr = red[0]; rz = jacobi ? red[1] : red[0]; }
    else {
        rr = blockSumAll(sumLocal(rPartial, rCount));
        rz = jacobi ? blockSumAll(sumLocal(rPartial + rCount, rCount)) : rr;
    }
    if (UPP && red) __syncthreads();
    const double alpha = sc->alpha, beta = rz / sc->rsold2[it & 1];      // pcg.h:331-335
    const bool nt = sc->vecNT != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->rr = rr; sc->rz = rz; sc->beta = beta; sc->rsold2[(it + 1) & 1] = rz; sc->rsold = rz; }
    double axx = 0., aup = 0.;
    const bool vec = ((((uintptr_t)p | (uintptr_t)r | (uintptr_t)x) & 15) == 0) && (((uintptr_t)dinv & DIAG_PAIR_MASK) == 0) &&
                     (!UPP || ((((uintptr_t)uCode & 1) == 0) && (((uintptr_t)uInv & 15) == 0)));
    const int64_t n2 = vec ? n / 2 : 0;
    const double2* r2 = (const double2*)r;
    double2* p2 = (double2*)p; double2* x2 = (double2*)x;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BS) {
        double2 z = ldD2(r2 + i, nt && PS_VEC_NT_RX);
        if (dinv) { const double2 dv = ldDiag2(dinv, i, nt && PS_VEC_NT_D); z.x = dv.x * z.x; z.y = dv.y * z.y; }
        double2 pv = ldD2(p2 + i, nt && PS_VEC_NT_PL), xv = ldD2(x2 + i, nt && PS_VEC_NT_X);
        xv.x = xv.x + alpha * pv.x; xv.y = xv.y + alpha * pv.y;
        pv.x = z.x + beta * pv.x; pv.y = z.y + beta * pv.y;
        stD2(x2 + i, xv, nt && PS_VEC_NT_X); stD2(p2 + i, pv, nt && PS_VEC_NT_P);
        axx += xv.x * xv.x; axx += xv.y * xv.y;
        if (UPP) {
            double u0, u1;
            if (uCode) { const uint16_t cc = (nt && PS_VEC_NT_U) ? __builtin_nontemporal_load((const uint16_t*)uCode + i) : ((const uint16_t*)uCode)[i]; u0 = dict[cc & 255]; u1 = dict[cc >> 8]; }
            else { const double2 uv = ldD2((const double2*)uInv + i, nt); u0 = uv.x; u1 = uv.y; }
            aup += u0 * (pv.x * pv.x); aup += u1 * (pv.y * pv.y);
        }
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double z = dinv ? diagValue(dinv[i]) * r[i] : r[i];
        const double pv = p[i];
        const double xv = x[i] + alpha * pv;
        const double pn = z + beta * pv;
        x[i] = xv; p[i] = pn;
        axx += xv * xv;
        if (UPP) aup += (uCode ? dict[uCode[i]] : uInv[i]) * (pn * pn);
    }
    const double s1 = blockReduceSum(axx);
    if (threadIdx.x == 0) partial[blockIdx.x] = s1;
    if (UPP) {
        const double s2 = blockReduceSum(aup);
        if (threadIdx.x == 0) uPart[blockIdx.x] = s2;
    }
}
__global__ void __launch_bounds__(BS) k_cg_update_xp(CGScalars* sc, const double* __restrict__ red, const double* __restrict__ rPartial, int rCount, int jacobi,
                                                     int it, const double* __restrict__ r, const diag_t* __restrict__ dinv, double* __restrict__ x,
                                                     double* __restrict__ p, int64_t n, double* __restrict__ partial) {
    cgUpdateXp<false>(sc, red, rPartial, rCount, jacobi, it, r, dinv, x, p, n, partial, nullptr, nullptr, nullptr, nullptr);
}
__global__ void __launch_bounds__(BS) k_cg_update_xp_u(CGScalars* sc, const double* __restrict__ red, const double* __restrict__ rPartial, int rCount, int jacobi,
                                                       int it, const double* __restrict__ r, const diag_t* __restrict__ dinv, double* __restrict__ x,
                                                       double* __restrict__ p, int64_t n, double* __restrict__ partial,
                                                       const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, const double* __restrict__ uInv, double* __restrict__ uPart) {
    cgUpdateXp<true>(sc, red, rPartial, rCount, jacobi, it, r, dinv, x, p, n, partial, uCode, uDict, uInv, uPart);
}
// partials of sum_j uInv_j p_j^2 (the first search direction of a fused-step solve)
__global__ void __launch_bounds__(BS) k_uinv_pp(const double* __restrict__ p, const uint8_t* __restrict__ uCode, const double* __restrict__ uDict,
                                                const double* __restrict__ uInv, int64_t n, double* __restrict__ uPart) {
    __shared__ double dict[256];
    if (uCode) dict[threadIdx.x] = uDict[threadIdx.x];
    __syncthreads();
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double pv = p[i];
        acc += (uCode ? dict[uCode[i]] : uInv[i]) * (pv * pv);
    }
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) uPart[blockIdx.x] = s;
}

// ---- four-kernel step across slabs (ps_dist.hpp) ---------------------------------------------------------------------------
// this rank's share of p.Ap in its factored form and of ||x||^2: out = {sum S + sum T + 1/2 sum U, sum xx}   (one block, fixed order)
// (1024 threads, all loads of a thread issued before the sums: 21 us -> a few with 256 threads walking the four arrays one after the other)
__global__ void __launch_bounds__(1024) k_fused_local_sum(const CGScalars* __restrict__ sc, const double* __restrict__ sPart, int sCount, const double* __restrict__ tPart, int tCount,
                                                         const double* __restrict__ uPart, int uCount, const double* __restrict__ xxPart, int xxCount, double* __restrict__ out) {
    if (sc && sc->done) return;
    __shared__ double red[4][16];
    double acc[4] = {0., 0., 0., 0.};
    const double* arr[4] = {sPart, tPart, uPart, xxPart};
    const int cnt[4] = {sCount, tCount, uCount, xxCount};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int i = threadIdx.x; i < cnt[q]; i += 1024) acc[q] += arr[q][i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double v = waveReduceSum(acc[q]); if (lane == 0) red[q][w] = v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4];
        for (int q = 0; q < 4; ++q) { double a = 0.; for (int i = 0; i < 16; ++i) a += red[q][i]; t[q] = a; }
        out[0] = t[0] + t[1] + 0.5 * t[2]; out[1] = t[3];
    }
}
// The St kernel updated r on the owned DOFs with this rank's rows only.  The DOFs next to a cut also receive the neighbour's share
// c of (A p)_j (its halo rows): r_j -= alpha c, and the partial sums of r.r / r.z are corrected by the change of r_j^2.
__global__ void __launch_bounds__(BS) k_dist_fixup(const CGScalars* __restrict__ sc, const int32_t* __restrict__ listA, int64_t nA, const double* __restrict__ bufA,
                                                   const int32_t* __restrict__ listB, int64_t nB, const double* __restrict__ bufB, double* __restrict__ r,
                                                   const diag_t* __restrict__ dinv, double* __restrict__ partial, int ownHi) {
    if (sc->done) return;
    const double alpha = sc->alpha;
    double a0 = 0., a1 = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < nA + nB; i += (int64_t)gridDim.x * BS) {
        const int j = i < nA ? listA[i] : listB[i - nA];
        if (j >= ownHi) continue;                        // a copy on its way to another rank (Dist::contributionsBack), not a DOF of this one
        const double c = i < nA ? bufA[i] : bufB[i - nA];
        const double ro = r[j], rn = ro - alpha * c;
        r[j] = rn;
        const double d = rn * rn - ro * ro;
        a0 += d;
        if (dinv) a1 += diagValue(dinv[j]) * d;
    }
    const double s0 = blockReduceSum(a0), s1 = blockReduceSum(a1);
    if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[gridDim.x + blockIdx.x] = s1; }
}
// out = {r.r, r.z} of this rank: the St kernel's partials plus the corrections of the fix-up ([2][fixCount] per set; r05: ONE set — the merged
// fix-up k_dist_fixup_merged — and the two corrections folded thread by thread before ONE reduction each: this one-block kernel sits on the
// critical path of every rank-iteration and took 17 us with 2 + 2 x 6 block reductions)   (one block)
__global__ void __launch_bounds__(1024) k_sum_rr(const CGScalars* __restrict__ sc, const double* __restrict__ rPart, int rCount, const double* __restrict__ fixPart, int fixCount,
                                               int fixSets, double* __restrict__ out) {
    // (r06: 1024 threads, every load of a thread issued before the sums, ONE round of wave reductions for the four values — as k_fused_local_sum:
    // 12 us -> a few with 256 threads and four block reductions in a row)
    if (sc->done) return;
    __shared__ double red[4][16];
    double acc[4] = {0., 0., 0., 0.};
    for (int i = threadIdx.x; i < rCount; i += 1024) { acc[0] += rPart[i]; acc[1] += rPart[rCount + i]; }
    for (int q = 0; q < fixSets; ++q)
        for (int i = threadIdx.x; i < fixCount; i += 1024) { acc[2] += fixPart[(size_t)q * 2 * fixCount + i]; acc[3] += fixPart[(size_t)q * 2 * fixCount + fixCount + i]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double v = waveReduceSum(acc[q]); if (lane == 0) red[q][w] = v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4];
        for (int q = 0; q < 4; ++q) { double a = 0.; for (int i = 0; i < 16; ++i) a += red[q][i]; t[q] = a; }
        out[0] = t[0] + t[2]; out[1] = t[1] + t[3];
    }
}
// The fix-up of the fused step in ONE launch (r05): a thread owns a DOF that receives contributions — from up to MAXSRC links (a DOF next to two or
// three cuts) — and applies them in link order, each to the r the previous one left: the arithmetic of the per-link launches of k_dist_fixup, DOF by
// DOF, without the launches (3 - 6 per rank-iteration at ~6 us each).  src[k][i] = (list index << 4) | buffer (0 .. 11: recvLo / recvUp of the six
// links, a pointer table in device memory), -1 = none; built once per setup (Dist::buildFixup).
constexpr int FIX_MAXSRC = 4;
__global__ void __launch_bounds__(BS) k_dist_fixup_merged(const CGScalars* __restrict__ sc, const int32_t* __restrict__ dof, const int32_t* __restrict__ src, int64_t n, const double* const* __restrict__ bufs,
                                                          double* __restrict__ r, const diag_t* __restrict__ dinv, double* __restrict__ partial) {
    if (sc->done) return;
    __shared__ const double* sb[16];                      // the twelve receive buffers: one level less in the dependent chain src -> buffer -> value
    if (threadIdx.x < 12) sb[threadIdx.x] = bufs[threadIdx.x];
    __syncthreads();
    const double alpha = sc->alpha;
    double a0 = 0., a1 = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const int j = dof[i];
        int e[FIX_MAXSRC];
#pragma unroll
        for (int k = 0; k < FIX_MAXSRC; ++k) e[k] = src[(size_t)k * (size_t)n + (size_t)i];
        double cv[FIX_MAXSRC];
#pragma unroll
        for (int k = 0; k < FIX_MAXSRC; ++k) cv[k] = e[k] >= 0 ? sb[e[k] & 15][e[k] >> 4] : 0.;   // all loads in flight before the sums
        const double ro = r[j];
        double rn = ro;
#pragma unroll
        for (int k = 0; k < FIX_MAXSRC; ++k)
            if (e[k] >= 0) rn -= alpha * cv[k];
        r[j] = rn;
        const double d = rn * rn - ro * ro;
        a0 += d;
        if (dinv) a1 += diagValue(dinv[j]) * d;
    }
    const double s0 = blockReduceSum(a0), s1 = blockReduceSum(a1);
    if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[gridDim.x + blockIdx.x] = s1; }
}

// ---- generic vector helpers (BiCGStab fallback, rare) -----------------------------------------------
__global__ void __launch_bounds__(BS) k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ partial) {
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) acc += a[i] * b[i];
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void __launch_bounds__(BS) k_sum1(const double* __restrict__ partial, int count, double* __restrict__ out) {
    const double s = sumPartials(partial, count);
    if (threadIdx.x == 0) *out = s;
}
// ---- Chebyshev-Jacobi polynomial preconditioner (PS_PRE_CHEBYSHEV) -------------------------------------------------
// first term: z_1 = dinv r / theta ; partial of r.z (used when the polynomial has this one term only)
// TZ: element type z is stored in (float: PS_PRE_CHEBYSHEV_F32; r.z is formed with the value as stored)
template <class TZ>
__global__ void __launch_bounds__(BS) k_cheb_first(const CGScalars* __restrict__ sc, const double* __restrict__ r, const diag_t* __restrict__ dinv,
                                                   double invTheta, TZ* __restrict__ z, int64_t n, double* __restrict__ partial) {
    if (sc && sc->done) return;
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double rv = r[i];
        const TZ vs = (TZ)(diagValue(dinv[i]) * rv * invTheta);
        z[i] = vs;
        const double v = (double)vs;
        acc += rv * v;
    }
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// a later term, unfused form (the St kernel's MODE 2 epilogue does the same per row): Az = A z_j given; z_{j+1} -> znext, which may be
// the z_{j-1} buffer (zprev null: z_{j-1} = 0)
__global__ void __launch_bounds__(BS) k_cheb_step(const CGScalars* __restrict__ sc, const double* __restrict__ r, const diag_t* __restrict__ dinv,
                                                  const double* __restrict__ Az, double c1, double c2, const double* __restrict__ z, const double* zprev,
                                                  double* znext, int64_t n, double* __restrict__ partial) {
    if (sc && sc->done) return;
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double rv = r[i], zj = z[i];
        const double zn = zj + (c1 * (zj - (zprev ? zprev[i] : 0.)) + c2 * (diagValue(dinv[i]) * (rv - Az[i])));
        znext[i] = zn;
        acc += rv * zn;
    }
    const double s = blockReduceSum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// beta = r.z / rsold ; x += alpha p ; p = z + beta p with z a VECTOR (polynomial preconditioner) ; partials of x.x
// UPP: also the partials of sum uInv p^2 of the new p (four-kernel step, see cgUpdateXp)
// TZ: element type of z (float: the single-precision Chebyshev polynomial)
__device__ inline double2 ldZ2(const double* z, int64_t i2, bool nt) { return ldD2((const double2*)z + i2, nt); }
__device__ inline double2 ldZ2(const float* z, int64_t i2, bool nt) {
    typedef float psf2z __attribute__((ext_vector_type(2)));
    const psf2z* q = reinterpret_cast<const psf2z*>(z) + i2;
    const psf2z v = nt ? __builtin_nontemporal_load(q) : *q;
    return make_double2((double)v.x, (double)v.y);
}
template <bool UPP, class TZ>
__device__ inline void cgUpdateXpZ(CGScalars* sc, const double* __restrict__ rrPartial, int rrCount, const double* __restrict__ rzPartial,
                                   int rzCount, int it, const TZ* __restrict__ z, double* __restrict__ x, double* __restrict__ p,
                                   int64_t n, double* __restrict__ partial,
                                   const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, const double* __restrict__ uInv, double* __restrict__ uPart) {
    if (sc->done) return;
    __shared__ double dict[UPP ? 256 : 1];
    if (UPP && uCode) dict[threadIdx.x] = uDict[threadIdx.x];   // visible after the barriers of blockSumAll below
    const double rr = blockSumAll(sumLocal(rrPartial, rrCount));
    const double rz = blockSumAll(sumLocal(rzPartial, rzCount));
    const double alpha = sc->alpha, beta = rz / sc->rsold2[it & 1];      // pcg.h:331-335
    const bool nt = sc->vecNT != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->rr = rr; sc->rz = rz; sc->beta = beta; sc->rsold2[(it + 1) & 1] = rz; sc->rsold = rz; }
    double axx = 0., aup = 0.;
    const bool vec = ((((uintptr_t)p | (uintptr_t)z | (uintptr_t)x) & 15) == 0) && (!UPP || ((((uintptr_t)uCode & 1) == 0) && (((uintptr_t)uInv & 15) == 0)));
    const int64_t n2 = vec ? n / 2 : 0;
    double2* p2 = (double2*)p; double2* x2 = (double2*)x;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BS) {
        const double2 zv = ldZ2(z, i, nt);
        double2 pv = ldD2(p2 + i, nt && PS_VEC_NT_PL), xv = ldD2(x2 + i, nt && PS_VEC_NT_X);
        xv.x = xv.x + alpha * pv.x; xv.y = xv.y + alpha * pv.y;
        pv.x = zv.x + beta * pv.x; pv.y = zv.y + beta * pv.y;
        stD2(x2 + i, xv, nt && PS_VEC_NT_X); stD2(p2 + i, pv, nt && PS_VEC_NT_P);
        axx += xv.x * xv.x; axx += xv.y * xv.y;
        if (UPP) {
            double u0, u1;
            if (uCode) { const uint16_t cc = (nt && PS_VEC_NT_U) ? __builtin_nontemporal_load((const uint16_t*)uCode + i) : ((const uint16_t*)uCode)[i]; u0 = dict[cc & 255]; u1 = dict[cc >> 8]; }
            else { const double2 uv = ldD2((const double2*)uInv + i, nt); u0 = uv.x; u1 = uv.y; }
            aup += u0 * (pv.x * pv.x); aup += u1 * (pv.y * pv.y);
        }
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double pv = p[i];
        const double xv = x[i] + alpha * pv;
        const double pn = (double)z[i] + beta * pv;
        x[i] = xv; p[i] = pn;
        axx += xv * xv;
        if (UPP) aup += (uCode ? dict[uCode[i]] : uInv[i]) * (pn * pn);
    }
    const double s1 = blockReduceSum(axx);
    if (threadIdx.x == 0) partial[blockIdx.x] = s1;
    if (UPP) {
        const double s2 = blockReduceSum(aup);
        if (threadIdx.x == 0) uPart[blockIdx.x] = s2;
    }
}
template <class TZ>
__global__ void __launch_bounds__(BS) k_cg_update_xp_z(CGScalars* sc, const double* __restrict__ rrPartial, int rrCount, const double* __restrict__ rzPartial,
                                                       int rzCount, int it, const TZ* __restrict__ z, double* __restrict__ x, double* __restrict__ p,
                                                       int64_t n, double* __restrict__ partial) {
    cgUpdateXpZ<false, TZ>(sc, rrPartial, rrCount, rzPartial, rzCount, it, z, x, p, n, partial, nullptr, nullptr, nullptr, nullptr);
}
template <class TZ>
__global__ void __launch_bounds__(BS) k_cg_update_xp_z_u(CGScalars* sc, const double* __restrict__ rrPartial, int rrCount, const double* __restrict__ rzPartial,
                                                         int rzCount, int it, const TZ* __restrict__ z, double* __restrict__ x, double* __restrict__ p,
                                                         int64_t n, double* __restrict__ partial,
                                                         const uint8_t* __restrict__ uCode, const double* __restrict__ uDict, const double* __restrict__ uInv, double* __restrict__ uPart) {
    cgUpdateXpZ<true, TZ>(sc, rrPartial, rrCount, rzPartial, rzCount, it, z, x, p, n, partial, uCode, uDict, uInv, uPart);
}
// one step of the power iteration on D^-1 A (ps_context::estimateLambdaMax): w = dinv .* Av ; partials of v.v and v.w
__global__ void __launch_bounds__(BS) k_power_step(const double* __restrict__ v, const double* __restrict__ Av, const double* __restrict__ dinv,
                                                   double* __restrict__ w, int64_t n, double* __restrict__ partial) {
    double avv = 0., avw = 0.;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        const double vi = v[i], wi = dinv[i] * Av[i];
        w[i] = wi;
        avv += vi * vi; avw += vi * wi;
    }
    const double s0 = blockReduceSum(avv), s1 = blockReduceSum(avw);
    if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[gridDim.x + blockIdx.x] = s1; }
}
__global__ void k_widen_f32(double* __restrict__ out, const float* __restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (double)a[i];
}
__global__ void k_fill_f64(double* __restrict__ a, double v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
}
// out[0] = sum of partial[0..count)   (one block, fixed order)
__global__ void __launch_bounds__(BS) k_sum_to(const double* __restrict__ partial, int count, double* __restrict__ out) {
    const double s = sumPartials(partial, count);
    if (threadIdx.x == 0) out[0] = s;
}
// out = a .* b
__global__ void k_mul_diag(double* __restrict__ out, const diag_t* __restrict__ d, const double* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = diagValue(d[i]) * b[i];
}
__global__ void k_mulv(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}
// constructGuessVectors (Solver.cpp:528-529): g holds -(S^T t) in the internal numbering; pressure entries stay, stress entries
// become -2 uInv g.  Indexed by REFERENCE index (pressures first) through permSys: every internal index is visited once.
__global__ void k_guess_finish(double* __restrict__ g, const double* __restrict__ uInv, const int32_t* __restrict__ permSys, int64_t nP, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nP) continue;
        const int32_t j = permSys[i];
        g[j] = -2. * uInv[j] * g[j];
    }
}
// out = ca*a + cb*b + cc*c  (null pointers skipped)
__global__ void k_lin(double* __restrict__ out, double ca, const double* __restrict__ a, double cb, const double* __restrict__ b, double cc,
                      const double* __restrict__ c, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = ca * a[i];
        if (b) v += cb * b[i];
        if (c) v += cc * c[i];
        out[i] = v;
    }
}

// ---- Jacobi diagonal (extension; reference stub Preconditioners.cpp:37-41) ------------------------
// diag_j = -dt sum_f McInv_f S_fj^2 - sum_r q^T BInv_r q - 1/2 uInv_j,  q = sum_{f in r} C_f S_fj
// A lane owns DOF j.  The entries of its St row are sorted by column and a tile's skin rows are contiguous, so the entries on ONE
// tile form a run: every lane folds its next run into q (26 moments), then the wave does the quadratic forms tile by tile — the
// tile's BInv is wave-uniform, read with scalar loads into SGPRs, its upper triangle only (BInv is symmetric up to rounding; the test
// bound is 1e-9): 351 fused multiply-adds per form instead of 676 multiplies, 676 adds and 676 vector loads of one address each
// (the vector-memory path takes 16 cycles per wave for such a load: the r02 kernel took 6.3 ms of the 256^3 setup, this one 3.8;
// fetching a row's entries up front and folding them through the 3 x 10 moments was tried: 220 VGPRs, 6.2 ms).
__global__ void __launch_bounds__(256) k_jacobi_diag(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                              const int8_t* __restrict__ code, double valScale, int n, int nP,
                              int nA, double dt, const double* __restrict__ McInv, const double* __restrict__ uInv,
                              const uint32_t* __restrict__ rrowFace, const int32_t* __restrict__ rrowRegion, const double* __restrict__ COM,
                              double dx, int3 off, const double* __restrict__ Binv, double* __restrict__ dinv, int invert) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < n;
    double diag = 0.;
    int p = live ? ptr[j] : 0;
    const int pe = live ? ptr[j + 1] : 0;
    while (true) {
        double q[PS_RD];
#pragma unroll
        for (int m = 0; m < PS_RD; ++m) q[m] = 0.;
        int cur = -1;
        while (p < pe) {
            const int f = col[p];
            const double v = val ? val[p] : (double)code[p] * valScale;   // (coded blocks keep no fp64 values: the same bits)
            if (f < nA) { diag += -dt * McInv[f] * v * v; ++p; continue; }
            const int rr = f - nA;
            const int r = rrowRegion[rr];
            if (cur >= 0 && r != cur) break;                            // the next tile's run: next round
            cur = r;
            double o[3];
            int axis;
            rowOffset(rrowFace[rr], COM, r, dx, off, o, &axis);
            double c[PS_RD];
            basisRow(o[0], o[1], o[2], axis, c);
#pragma unroll
            for (int m = 0; m < PS_RD; ++m) q[m] += c[m] * v;
            ++p;
        }
        bool pend = cur >= 0;
        unsigned long long todo = __ballot(pend);
        if (todo == 0ull) break;                                        // wave-uniform: no lane has a run left
        while (todo != 0ull) {
            const int rl = __builtin_amdgcn_readlane(cur, __ffsll((long long)todo) - 1);
            const double* __restrict__ B = Binv + (int64_t)rl * PS_RD * PS_RD;   // wave-uniform: scalar loads
            double s = 0.;
#pragma unroll
            for (int m = 0; m < PS_RD; ++m) {
                double t = 0.;
#pragma unroll
                for (int k = m + 1; k < PS_RD; ++k) t = __builtin_fma(B[m * PS_RD + k], q[k], t);
                s = __builtin_fma(q[m], __builtin_fma(B[m * PS_RD + m], q[m], 2. * t), s);
            }
            if (pend && cur == rl) { diag -= s; pend = false; }          // (lanes of other tiles computed a value they drop)
            todo = __ballot(pend);
        }
    }
    if (!live) return;
    diag += -0.5 * uInv[j];
    dinv[j] = invert ? (diag != 0. ? 1. / diag : 1.) : diag;   // raw diagonal when halo contributions are still to be added
}

// ---- recovery and write-back ---------------------------------------------------------------------
// u_a = dt McInv (invDt rhs_a - (G p + Dt tau))      Solver.cpp:507
__global__ void k_recover_active(const double* __restrict__ s, const double* __restrict__ McInv, const double* __restrict__ rhsA, double dt,
                                 double invDt, int64_t nA, double* __restrict__ ua) {
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nA; f += (int64_t)gridDim.x * blockDim.x)
        ua[f] = dt * McInv[f] * (invDt * rhsA[f] - s[f]);
}
// applySolutionToVelocity, Solver.cpp:937-1028
__global__ void k_writeback(Grid g, int axis, const int32_t* __restrict__ lab, const int32_t* __restrict__ act, const int32_t* __restrict__ reg,
                            const int32_t* __restrict__ faceRow, const double* __restrict__ ua, const double* __restrict__ creg, const double* __restrict__ COM,
                            double dx, int3 off, const float* __restrict__ cvel, const float* __restrict__ velIn, float* __restrict__ velOut, int apply) {
    const int3 d = g.dims(1 + axis);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int64_t)d.x * d.y * d.z) return;
    const int l = lab[c];
    float out = velIn[c];
    if (apply && !(l == PS_UNSOLVED || l == PS_UNASSIGNED)) {
        const int r = reg[c];
        const int a = act[c];
        double v = 0.;
        if (r >= 0) {
            const int3 q = unlin3(d, c);
            double p[3] = {(double)(q.x + off.x), (double)(q.y + off.y), (double)(q.z + off.z)};
            p[axis] -= 0.5;
            const double ox = p[0] * dx - COM[(int64_t)r * 3 + 0], oy = p[1] * dx - COM[(int64_t)r * 3 + 1], oz = p[2] * dx - COM[(int64_t)r * 3 + 2];
            double C[PS_RD];
            basisRow(ox, oy, oz, axis, C);
            double s = 0.;
            for (int n = 0; n < PS_RD; ++n) s += creg[(int64_t)r * PS_RD + n] * C[n];
            v = s;
        } else if (a >= 0) {
            const int row = faceRow[c];
            v = row >= 0 ? ua[row] : (double)velIn[c];   // active face of another rank (halo): left untouched
        } else if (l == PS_SOLID) {
            v = (double)cvel[c];
        }
        out = (float)v;
    }
    velOut[c] = out;
}

