// TEST INFRASTRUCTURE — see ps_oracle.hpp.  Operator, Krylov solvers, recovery, write-back, C API.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <limits>

#include "ps_oracle.hpp"

namespace psoracle {

static inline bool isActive(int32_t l) { return l == PS_ACTIVEFLUID || l == PS_BOUNDARY; }

static double dot(const std::vector<double>& a, const std::vector<double>& b) {
    double s = 0;
    for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
    return s;
}

// ApplyPressureStressMatrix::applyMatrixVectorProducts, lib/include/ApplyPressureStressMatrix.h:102-179,
// with manualMatrixTransposeVectorDistribute2 (lib/include/util.h:203-230).  Same pass structure,
// including the per-call McInv*G / McInv*Dt products (:126,:156) and by-value temporaries.
// The three `omp section` bodies of applyMatrixVectorProducts (:122-164), one function each so that the single-thread
// parity path below and the timing harness (ps_oracle_mt.cpp: the same three bodies under `#pragma omp parallel sections`,
// BASELINE.md section 2 "baseline A") run the same code.
void Oracle::applySection1(const double* x, std::vector<double>& A11_1, std::vector<double>& A21_1) const {   // :124-134
    const int64_t nP = nPressures, nT = nStresses, nA = nActiveVs;
    const double* x_ps = x;
    std::vector<double> McInv_G_val(G.val.size());   // SparseMatrix McInv_G = McInv_Matrix * G_Matrix, formed on EVERY call (:126)
    for (int64_t f = 0; f < nA; ++f)
        for (int64_t p = G.ptr[(size_t)f]; p < G.ptr[(size_t)f + 1]; ++p) McInv_G_val[(size_t)p] = McInv[(size_t)f] * G.val[(size_t)p];
    std::vector<double> McInv_G_xps((size_t)nA);
    for (int64_t f = 0; f < nA; ++f) {
        double s = 0;
        for (int64_t p = G.ptr[(size_t)f]; p < G.ptr[(size_t)f + 1]; ++p) s += McInv_G_val[(size_t)p] * x_ps[G.col[(size_t)p]];
        McInv_G_xps[(size_t)f] = s;
    }
    A11_1.assign((size_t)nP, 0.); A21_1.assign((size_t)nT, 0.);
    Gt.mul(McInv_G_xps.data(), A11_1.data());
    for (auto& v : A11_1) v = -dt * v;
    D.mul(McInv_G_xps.data(), A21_1.data());
    for (auto& v : A21_1) v = -dt * v;
}
void Oracle::applySection2(const double* x, std::vector<double>& tp, std::vector<double>& tt) const {   // :136-152
    const int64_t nP = nPressures, nR = nReducedVs;
    const double* x_ps = x;
    const double* x_ts = x + nP;
    std::vector<double> BInv_JDt_xts((size_t)nR), BInv_JG_xps((size_t)nR), tmp((size_t)nR);
    auto binvMul = [&](const std::vector<double>& in, std::vector<double>& out) {
        for (int64_t r = 0; r < regionCount; ++r)
            for (int m = 0; m < RD; ++m) {
                double s = 0;
                for (int n = 0; n < RD; ++n) s += Binv[(size_t)r * RD * RD + m * RD + n] * in[(size_t)r * RD + n];
                out[(size_t)r * RD + m] = s;
            }
    };
    JDt.mul(x_ts, tmp.data());
    binvMul(tmp, BInv_JDt_xts);
    JG.mul(x_ps, tmp.data());
    binvMul(tmp, BInv_JG_xps);
    auto distribute2 = [&](const CSR& mat, std::vector<double>& out) {   // util.h:203-230
        const int64_t nDofs = mat.cols;
        out.assign((size_t)nDofs * 2, 0.);
        for (int64_t i = 0; i != regionCount; ++i)
            for (int j = 0; j < RD; ++j) {
                const int64_t colnum = i * RD + j;
                for (int64_t p = mat.ptr[(size_t)colnum]; p < mat.ptr[(size_t)colnum + 1]; ++p) {
                    out[(size_t)mat.col[(size_t)p]] += mat.val[(size_t)p] * BInv_JG_xps[(size_t)colnum];
                    out[(size_t)(mat.col[(size_t)p] + nDofs)] += mat.val[(size_t)p] * BInv_JDt_xts[(size_t)colnum];
                }
            }
        for (auto& v : out) v = -v;
    };
    distribute2(JG, tp);    // A11_2 = head, A12_2 = tail
    distribute2(JDt, tt);   // A21_2 = head, A22_2 = tail
}
void Oracle::applySection3(const double* x, std::vector<double>& A12_1, std::vector<double>& A22_1) const {   // :154-162
    const int64_t nP = nPressures, nT = nStresses, nA = nActiveVs;
    const double* x_ts = x + nP;
    std::vector<double> McInv_Dt_val(Dt.val.size());   // SparseMatrix McInv_Dt = McInv_Matrix * Dt_Matrix, per call (:156)
    for (int64_t f = 0; f < nA; ++f)
        for (int64_t p = Dt.ptr[(size_t)f]; p < Dt.ptr[(size_t)f + 1]; ++p) McInv_Dt_val[(size_t)p] = McInv[(size_t)f] * Dt.val[(size_t)p];
    std::vector<double> McInv_Dt_xts((size_t)nA);
    for (int64_t f = 0; f < nA; ++f) {
        double s = 0;
        for (int64_t p = Dt.ptr[(size_t)f]; p < Dt.ptr[(size_t)f + 1]; ++p) s += McInv_Dt_val[(size_t)p] * x_ts[Dt.col[(size_t)p]];
        McInv_Dt_xts[(size_t)f] = s;
    }
    A12_1.assign((size_t)nP, 0.); A22_1.assign((size_t)nT, 0.);
    Gt.mul(McInv_Dt_xts.data(), A12_1.data());
    for (auto& v : A12_1) v = -dt * v;
    D.mul(McInv_Dt_xts.data(), A22_1.data());
    for (auto& v : A22_1) v = -dt * v;
}
void Oracle::applyCombine(const double* x, const std::vector<double>& A11_1, const std::vector<double>& A21_1, const std::vector<double>& tp,
                          const std::vector<double>& tt, const std::vector<double>& A12_1, const std::vector<double>& A22_1, double* y) const {   // :166-176
    const int64_t nP = nPressures, nT = nStresses;
    const double* x_ts = x + nP;
    for (int64_t i = 0; i < nP; ++i) {
        const double A11 = A11_1[(size_t)i] + tp[(size_t)i];
        const double A12 = A12_1[(size_t)i] + tp[(size_t)(i + nP)];
        y[i] = A11 + A12;
    }
    for (int64_t i = 0; i < nT; ++i) {
        const double A21 = A21_1[(size_t)i] + tt[(size_t)i];
        const double A22_3 = -0.5 * uInv[(size_t)i] * x_ts[i];
        const double A22 = A22_1[(size_t)i] + tt[(size_t)(i + nT)] + A22_3;
        y[nP + i] = A21 + A22;
    }
}
void Oracle::applyOperator(const double* x, double* y) const {
    std::vector<double> A11_1, A21_1, tp, tt, A12_1, A22_1;
    applySection1(x, A11_1, A21_1);
    applySection2(x, tp, tt);
    applySection3(x, A12_1, A22_1);
    applyCombine(x, A11_1, A21_1, tp, tt, A12_1, A22_1, y);
}

// "Fair CPU" variant of the same operator (BASELINE.md §2 baseline B): t = McInv [G Dt] x once,
// y = -dt [G Dt]^T t - [JG JDt]^T BInv [JG JDt] x - 1/2 uInv x_t, no per-call matrix products.
void Oracle::applyOperatorFair(const double* x, double* y) const {
    const int64_t nP = nPressures, nT = nStresses, nA = nActiveVs, nR = nReducedVs;
    std::vector<double> t((size_t)nA), w((size_t)nR), v((size_t)nR), w2((size_t)nR);
    G.mul(x, t.data());
    std::vector<double> t2((size_t)nA);
    Dt.mul(x + nP, t2.data());
    for (int64_t f = 0; f < nA; ++f) t[(size_t)f] = -dt * McInv[(size_t)f] * (t[(size_t)f] + t2[(size_t)f]);
    Gt.mul(t.data(), y);
    D.mul(t.data(), y + nP);
    JG.mul(x, w.data());
    JDt.mul(x + nP, w2.data());
    for (int64_t r = 0; r < regionCount; ++r)
        for (int m = 0; m < RD; ++m) {
            double s = 0;
            for (int n = 0; n < RD; ++n) s += Binv[(size_t)r * RD * RD + m * RD + n] * (w[(size_t)r * RD + n] + w2[(size_t)r * RD + n]);
            v[(size_t)r * RD + m] = -s;
        }
    JG.mulT_add(v.data(), y);
    JDt.mulT_add(v.data(), y + nP);
    for (int64_t i = 0; i < nT; ++i) y[nP + i] += -0.5 * uInv[(size_t)i] * x[nP + i];
}

// The diagonal of the two preconditioner EXTENSIONS (Jacobi, Chebyshev-Jacobi; the reference has neither: its Jacobi is a stub,
// Preconditioners.cpp:37-41) is DEFINED as 1 / A_jj rounded to the upper 16 bits of its fp32 value, to nearest even — the form the
// product stores it in (polystokes_amd/csrc/ps_common.hpp: diag_t).  Any fixed positive diagonal preconditions; restating the
// rounding here keeps z = M^-1 r and the iteration counts comparable to the last digits instead of to 0.4 %.  The interval
// estimate below and Eigen's own diagonal preconditioner (eigenCG, reference behaviour) use the unrounded diagonal.
// exactDiagonal (po_set_exact_diagonal; default off): the textbook Jacobi diagonal 1 / A_jj in fp64 instead — what "Jacobi-PCG" means without
// the product's storage format.  tests/test_gpu_parity.py::test_stored_diagonal_jacobi_is_equivalent_to_exact_jacobi compares the product
// (16-bit diagonal) with THAT solve: iterations within 2 %, x within 10 tol.
static inline double storedDinv(double diag, bool exact) {
    if (exact) return diag != 0. ? 1. / diag : 1.;
    const float f = (float)(diag != 0. ? 1. / diag : 1.);
    uint32_t b;
    std::memcpy(&b, &f, 4);
    b = ((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16) << 16;
    float g;
    std::memcpy(&g, &b, 4);
    return (double)g;
}
// z = M^-1 r: identity (Preconditioner.cpp:271-274), the Jacobi extension, or the Chebyshev extension
void Oracle::precondition(const std::vector<double>& in, std::vector<double>& out) const {
    const size_t n = in.size();
    if (P.preconditioner == PS_PRE_CHEBYSHEV) { chebyshev(in, out); return; }
    if (P.preconditioner == PS_PRE_CHEBYSHEV_F32) { chebyshev32(in, out); return; }
    if (P.preconditioner != PS_PRE_DIAGONAL) { out = in; return; }
    out.resize(n);
    for (size_t i = 0; i < n; ++i) out[i] = storedDinv(diagA[i], exactDiagonal) * in[i];
}
// Largest eigenvalue of D^-1 A by 10 power iterations from the all-ones vector (Rayleigh quotient of the last iterate), with
// the safety margin the polynomial needs: an UNDER-estimate would make it negative beyond the interval.  A = sum over faces of
// rank-one terms with <= 8 entries, so lambda_max(D^-1 A) <= 8 for the stencil part (Cauchy-Schwarz); the tile part is not
// covered by that bound, hence the measurement:  lmax = max(8.4, 1.25 * estimate).
void Oracle::estimateLambdaMax() {
    const size_t n = (size_t)(nPressures + nStresses);
    std::vector<double> v(n, 1.), w(n), Av(n);
    double lam = 0.;
    for (int it = 0; it < 10 && n > 0; ++it) {
        applyOperator(v.data(), Av.data());
        for (size_t i = 0; i < n; ++i) w[i] = (diagA[i] != 0. ? 1. / diagA[i] : 1.) * Av[i];
        const double vv = dot(v, v);
        lam = dot(v, w) / vv;
        const double nw = std::sqrt(dot(w, w));
        if (nw == 0.) break;
        for (size_t i = 0; i < n; ++i) v[i] = w[i] / nw;
    }
    chebLmax = std::max(8.4, 1.25 * lam);
}
// k terms of the Chebyshev iteration for D^-1 A z = D^-1 r from z = 0 on [lmax/PS_CHEB_INTERVAL_RATIO, lmax] (k-1 operator applies)
void Oracle::chebyshev(const std::vector<double>& r, std::vector<double>& z) const {
    const size_t n = r.size();
    const int k = P.preconditionerDegree > 0 ? P.preconditionerDegree : 4;
    const double lmax = chebLmax, lmin = lmax / PS_CHEB_INTERVAL_RATIO;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1. / sigma;
    std::vector<double> d(n), Az(n);
    z.resize(n);
    auto dinv = [&](size_t i) { return storedDinv(diagA[i], exactDiagonal); };
    for (size_t i = 0; i < n; ++i) { d[i] = dinv(i) * r[i] / theta; z[i] = d[i]; }
    for (int j = 1; j < k; ++j) {
        const double rhoN = 1. / (2. * sigma - rho);
        const double c1 = rhoN * rho, c2 = 2. * rhoN / delta;
        applyOperator(z.data(), Az.data());
        for (size_t i = 0; i < n; ++i) {
            const double res = dinv(i) * (r[i] - Az[i]);
            d[i] = c1 * d[i] + c2 * res;
            z[i] = z[i] + d[i];
        }
        rho = rhoN;
    }
}

// PS_PRE_CHEBYSHEV_F32 (include/polystokes.h; an extension like the polynomial itself — there is no reference algorithm to depart from): the
// polynomial above with its INNER vectors stored in single precision, restated BEFORE the kernels were written to bound
// what the storage rounding does to the iteration count.  Rounded to fp32 here, at the points where the product stores a value: every iterate
// z_j (three-term form: z_{j+1} = z_j + c1 (z_j - z_{j-1}) + c2 dinv (r - A z_j), the form the product's St epilogue evaluates) and the
// ACTIVE rows of the face-row vector t = dt McInv [G Dt] z of each inner apply.  NOT restated: the product also stores the tile rows' share of
// that vector (s = [Ghat Dhat] z before the 26x26 block, t = J v after it) as fp32 — this oracle holds JG = J^T Ghat, not the per-row products,
// so its tile part stays fp64.  The two therefore agree to the rounding LEVEL (z within ~1e-6 of its norm), not to the bit: the tests say so.
// r, the stored diagonal, every product and sum, and the outer PCG with its stop rule are fp64 in both.
static inline double f32r(double v) { return (double)(float)v; }
void Oracle::applyOperatorInner32(const double* x, double* y) const {
    const int64_t nP = nPressures, nT = nStresses, nA = nActiveVs, nR = nReducedVs;
    std::vector<double> t((size_t)nA), t2((size_t)nA), w((size_t)nR), w2((size_t)nR), v((size_t)nR);
    G.mul(x, t.data());
    Dt.mul(x + nP, t2.data());
    for (int64_t f = 0; f < nA; ++f) t[(size_t)f] = -f32r((t[(size_t)f] + t2[(size_t)f]) * (dt * McInv[(size_t)f]));   // the stored fp32 value, sign folded in
    Gt.mul(t.data(), y);
    D.mul(t.data(), y + nP);
    JG.mul(x, w.data());
    JDt.mul(x + nP, w2.data());
    for (int64_t r = 0; r < regionCount; ++r)
        for (int m = 0; m < RD; ++m) {
            double s = 0;
            for (int n = 0; n < RD; ++n) s += Binv[(size_t)r * RD * RD + m * RD + n] * (w[(size_t)r * RD + n] + w2[(size_t)r * RD + n]);
            v[(size_t)r * RD + m] = -s;
        }
    JG.mulT_add(v.data(), y);
    JDt.mulT_add(v.data(), y + nP);
    for (int64_t i = 0; i < nT; ++i) y[nP + i] += -0.5 * uInv[(size_t)i] * x[nP + i];
}
void Oracle::chebyshev32(const std::vector<double>& r, std::vector<double>& z) const {
    const size_t n = r.size();
    const int k = P.preconditionerDegree > 0 ? P.preconditionerDegree : 4;
    const double lmax = chebLmax, lmin = lmax / PS_CHEB_INTERVAL_RATIO;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1. / sigma;
    std::vector<double> zprev(n, 0.), Az(n), znext(n);
    z.resize(n);
    auto dinv = [&](size_t i) { return storedDinv(diagA[i], exactDiagonal); };
    for (size_t i = 0; i < n; ++i) z[i] = f32r(dinv(i) * r[i] * (1. / theta));
    for (int j = 1; j < k; ++j) {
        const double rhoN = 1. / (2. * sigma - rho);
        const double c1 = rhoN * rho, c2 = 2. * rhoN / delta;
        applyOperatorInner32(z.data(), Az.data());
        for (size_t i = 0; i < n; ++i) znext[i] = f32r(z[i] + (c1 * (z[i] - zprev[i]) + c2 * (dinv(i) * (r[i] - Az[i]))));
        zprev.swap(z);
        z.swap(znext);
        rho = rhoN;
    }
}

// pcg_external_matrix_A, lib/include/pcg.h:268-340.  Preconditioner: identity (Preconditioner.cpp:18-28,
// 271-274) or the Jacobi extension.  Deviation: b == 0 returns immediately (reference divides 0/0, pcg.h:314).
int Oracle::pcg(std::vector<double>& x, const std::vector<double>& rhs, double tol, int maxit, double& rre) const {
    const size_t n = rhs.size();
    std::vector<double> r(n), z(n), p(n), Ap(n);
    auto pre = [&](const std::vector<double>& in, std::vector<double>& out) { precondition(in, out); };
    applyOperator(x.data(), Ap.data());
    for (size_t i = 0; i < n; ++i) r[i] = rhs[i] - Ap[i];
    pre(r, z);
    p = z;
    double rsold = dot(r, z), rsnew = 0., alpha = 0., beta = 0., xmag = 0.;
    rre = 0.;
    if (rsold == 0.) return 0;
    for (int i = 0; i < maxit; ++i) {
        applyOperator(p.data(), Ap.data());
        alpha = rsold / dot(p, Ap);
        for (size_t q = 0; q < n; ++q) x[q] = x[q] + alpha * p[q];
        for (size_t q = 0; q < n; ++q) r[q] = r[q] - alpha * Ap[q];
        rsnew = dot(r, r);
        xmag = dot(x, x);
        rre = rsnew;
        if (rsnew / xmag < rre) rre = rsnew / xmag;
        if (rre < tol * tol) { rre = std::sqrt(rre); return i; }
        pre(r, z);
        rsnew = dot(r, z);
        beta = rsnew / rsold;
        for (size_t q = 0; q < n; ++q) p[q] = z[q] + beta * p[q];
        rsold = rsnew;
    }
    rre = std::sqrt(rre);
    return maxit;
}

// bicgstab_external_matrix_A, lib/include/pcg.h:134-200
int Oracle::bicgstab(std::vector<double>& x, const std::vector<double>& rhs, double tol, int maxit, double& rre) const {
    const size_t n = rhs.size();
    std::vector<double> r(n), Ax(n);
    applyOperator(x.data(), Ax.data());
    for (size_t i = 0; i < n; ++i) r[i] = rhs[i] - Ax[i];
    std::vector<double> rhat = r, p(n, 0.), v(n, 0.), h(n, 0.), s(n, 0.), t(n, 0.), err(n, 0.);
    double rhoCurr = 1., rhoOld = 1., alpha = 1., beta = 0., omega = 1., xmag = 0., rsnew = 0.;
    for (int i = 0; i < maxit; ++i) {
        rhoOld = rhoCurr;
        rhoCurr = dot(rhat, r);
        beta = (rhoCurr / rhoOld) * (alpha / omega);
        for (size_t q = 0; q < n; ++q) p[q] = r[q] + beta * (p[q] - omega * v[q]);
        applyOperator(p.data(), v.data());
        alpha = rhoCurr / dot(rhat, v);
        for (size_t q = 0; q < n; ++q) h[q] = x[q] + alpha * p[q];
        for (size_t q = 0; q < n; ++q) s[q] = r[q] - alpha * v[q];
        applyOperator(s.data(), t.data());
        omega = dot(t, s) / dot(t, t);
        for (size_t q = 0; q < n; ++q) x[q] = h[q] + omega * s[q];
        xmag = std::sqrt(dot(x, x));
        applyOperator(x.data(), Ax.data());
        for (size_t q = 0; q < n; ++q) err[q] = rhs[q] - Ax[q];
        rsnew = dot(err, err);
        rre = rsnew;
        if (std::sqrt(rsnew) / xmag < rre) rre = std::sqrt(rsnew) / xmag;
        if (rre < tol) return i;
        for (size_t q = 0; q < n; ++q) r[q] = s[q] - omega * t[q];
    }
    return maxit;
}

// Eigen::ConjugateGradient<SparseMatrix, Lower|Upper> on the explicit A with the default diagonal
// preconditioner (Solver.cpp:814-862; extern/eigen/Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:30-93,
// BasicPreconditioners.h:69-77).  BASELINE config 1 only.
int Oracle::eigenCG(std::vector<double>& x, const std::vector<double>& rhs, double tol, int maxit, double& tolError) const {
    const size_t n = rhs.size();
    std::vector<double> invdiag(n, 1.);
    for (int64_t i = 0; i < A.rows; ++i)
        for (int64_t p = A.ptr[(size_t)i]; p < A.ptr[(size_t)i + 1]; ++p)
            if (A.col[(size_t)p] == i) invdiag[(size_t)i] = A.val[(size_t)p] != 0. ? 1. / A.val[(size_t)p] : 1.;
    std::vector<double> residual(n), p(n), z(n), tmp(n);
    A.mul(x.data(), tmp.data());
    for (size_t i = 0; i < n; ++i) residual[i] = rhs[i] - tmp[i];
    const double rhsNorm2 = dot(rhs, rhs);
    if (rhsNorm2 == 0) { std::fill(x.begin(), x.end(), 0.); tolError = 0; return 0; }
    const double threshold = std::max(tol * tol * rhsNorm2, std::numeric_limits<double>::min());
    double residualNorm2 = dot(residual, residual);
    if (residualNorm2 < threshold) { tolError = std::sqrt(residualNorm2 / rhsNorm2); return 0; }
    for (size_t i = 0; i < n; ++i) p[i] = invdiag[i] * residual[i];
    double absNew = dot(residual, p);
    int i = 0;
    while (i < maxit) {
        A.mul(p.data(), tmp.data());
        const double alpha = absNew / dot(p, tmp);
        for (size_t q = 0; q < n; ++q) x[q] += alpha * p[q];
        for (size_t q = 0; q < n; ++q) residual[q] -= alpha * tmp[q];
        residualNorm2 = dot(residual, residual);
        if (residualNorm2 < threshold) break;
        for (size_t q = 0; q < n; ++q) z[q] = invdiag[q] * residual[q];
        const double absOld = absNew;
        absNew = dot(residual, z);
        const double beta = absNew / absOld;
        for (size_t q = 0; q < n; ++q) p[q] = z[q] + beta * p[q];
        i++;
    }
    tolError = std::sqrt(residualNorm2 / rhsNorm2);
    return i;
}

// initializeGuessVectors / constructGuessVectors (Solver.cpp:512-531) and the guessVector every assemble*() fills
// (AssembleSystem.cpp:421-427, 461-467): [pressureGuess; stressGuess].  Only solveEigenCG uses it (solveWithGuess, :834);
// the matrix-vector PCG starts from zero (:768) — but exportMatrices writes it as Vec_guess.mtx either way (:540).
void Oracle::constructGuessVectors() {
    const int64_t nP = nPressures, nT = nStresses;
    guess.assign((size_t)(nP + nT), 0.);
    if (!P.useWarmStart) return;
    // pressureGuess = -G^T oldActiveVs - JG^T cfit ;  stressGuess = -2 uInv (-Dt^T oldActiveVs - JDt^T cfit)
    std::vector<double> gp((size_t)nP, 0.), gt((size_t)nT, 0.);
    G.mulT_add(oldActiveVs.data(), gp.data());
    Dt.mulT_add(oldActiveVs.data(), gt.data());
    if (regionCount > 0) {
        JG.mulT_add(cfit.data(), gp.data());
        JDt.mulT_add(cfit.data(), gt.data());
    }
    for (int64_t i = 0; i < nP; ++i) guess[(size_t)i] = -gp[(size_t)i];
    for (int64_t i = 0; i < nT; ++i) guess[(size_t)(nP + i)] = -2. * uInv[(size_t)i] * (-gt[(size_t)i]);
}

// Solver.cpp:646-668, 734-812 (solveSPDwithMatrixVectorPCG) / :814-862 (solveEigenCG)
int Oracle::solve() {
    const double tol = P.tolerance;
    const int maxit = P.maxSolverIterations;
    const std::clock_t c0 = std::clock();
    const auto w0 = std::chrono::high_resolution_clock::now();
    int result = PS_NOCHANGE;
    stats.usedBiCGStab = 0;
    std::fill(solution.begin(), solution.end(), 0.);   // :768
    if (P.solverType == PS_EIGEN) {
        assembleSystemPressureStress();
        double e = 0;
        solution = guess;   // solver.solveWithGuess(b, guessVector), :834
        solveIterations = eigenCG(solution, b, tol, maxit, e);
        solveError = e;
        result = e <= tol ? PS_SUCCESS : PS_NOCONVERGE;   // IterativeSolverBase: info = error <= tolerance
    } else {
        double rre = 0;
        solveIterations = pcg(solution, b, tol, maxit, rre);
        if (solveIterations == maxit) {   // :784-799
            std::fill(solution.begin(), solution.end(), 0.);
            solveIterations = bicgstab(solution, b, tol, maxit, rre);
            stats.usedBiCGStab = 1;
        }
        solveError = rre;
        result = solveIterations == maxit ? PS_NOCONVERGE : PS_SUCCESS;   // :808-811
    }
    const auto w1 = std::chrono::high_resolution_clock::now();
    stats.solveData[0] = solveError;
    stats.solveData[1] = solveIterations;
    stats.solveData[2] = 1000.0 * (double)(std::clock() - c0) / CLOCKS_PER_SEC;
    stats.solveData[3] = std::chrono::duration<double, std::milli>(w1 - w0).count();
    return result;
}

// Solver.cpp:492-510
void Oracle::recoverVelocityFromPressureStress() {
    const int64_t nP = nPressures, nA = nActiveVs, nR = nReducedVs;
    const double* ps = solution.data();
    const double* ts = solution.data() + nP;
    std::vector<double> gp((size_t)nA), dtau((size_t)nA), jg((size_t)nR), jd((size_t)nR);
    G.mul(ps, gp.data());
    Dt.mul(ts, dtau.data());
    JG.mul(ps, jg.data());
    JDt.mul(ts, jd.data());
    recovered.assign((size_t)(nA + nR), 0.);
    for (int64_t f = 0; f < nA; ++f)
        recovered[(size_t)f] = dt * McInv[(size_t)f] * (invDt * activeRHS[(size_t)f] - gp[(size_t)f] - dtau[(size_t)f]);
    for (int64_t r = 0; r < regionCount; ++r)
        for (int m = 0; m < RD; ++m) {
            double s = 0;
            for (int n = 0; n < RD; ++n) {
                const size_t q = (size_t)r * RD + n;
                s += Binv[(size_t)r * RD * RD + m * RD + n] * (invDt * reducedRHS[q] - jg[q] - jd[q]);
            }
            recovered[(size_t)(nA + r * RD + m)] = s;
        }
}

// Solver.cpp:937-1028
void Oracle::applySolutionToVelocity() {
    for (int axis = 0; axis < 3; ++axis) {
        const Dim fd = faceDim(axis);
        velOut[axis] = vel[axis];
        for (int k = 0; k < fd.n[2]; ++k)
            for (int j = 0; j < fd.n[1]; ++j)
                for (int i = 0; i < fd.n[0]; ++i) {
                    if (valid[axis].at(i, j, k) == 0.f) continue;
                    const int32_t faceLabel = labels[1 + axis].at(i, j, k);
                    const int64_t localActive = activeIdx[1 + axis].at(i, j, k);
                    const int64_t reducedFaceIndex = reducedIdx[1 + axis].at(i, j, k);
                    double localVelocity = 0.;
                    if (reducedFaceIndex >= 0) {
                        double off[3] = {(double)i, (double)j, (double)k};
                        off[axis] -= 0.5;
                        for (int q = 0; q < 3; ++q) { off[q] *= dx; off[q] -= COM[(size_t)reducedFaceIndex * 3 + q]; }
                        double C[RD];
                        buildConversionCoefficients(off, axis, C);
                        double s = 0;
                        for (int n = 0; n < RD; ++n) s += recovered[(size_t)(nActiveVs + RD * reducedFaceIndex + n)] * C[n];
                        localVelocity = s;
                    } else if (localActive >= 0) {
                        localVelocity = recovered[(size_t)faceVelocityDOF(localActive, axis)];
                    } else if (faceLabel == PS_SOLID) {
                        localVelocity = (double)collisionvel[axis].at(i, j, k);
                    }
                    velOut[axis].at(i, j, k) = (float)localVelocity;
                }
    }
}

int Oracle::load(const ps_params* p, const ps_fields_in* in) {
    P = *p;
    nx = in->nx; ny = in->ny; nz = in->nz;
    dx = in->dx; invDx = 1. / dx; dt = in->dt; invDt = 1. / dt;
    rho = (double)in->density;
    if (nx <= 0 || ny <= 0 || nz <= 0) { err = "bad resolution"; return PS_INVALID; }
    if (!in->surface) { err = "Surface field is missing."; return PS_INVALID; }
    if (!in->collision) { err = "Collision field is missing."; return PS_INVALID; }
    if (!in->viscosity) { err = "Viscosity field is missing."; return PS_INVALID; }
    for (int a = 0; a < 3; ++a) if (!in->vel[a]) { err = "Velocity field is missing."; return PS_INVALID; }
    auto cp = [](Field<float>& f, const Dim& d, const float* src) {
        f.init(d, 0.f);
        if (src) std::memcpy(f.v.data(), src, sizeof(float) * (size_t)d.size());
    };
    cp(surface, centerDim(), in->surface);
    cp(collision, centerDim(), in->collision);
    cp(viscosity, centerDim(), in->viscosity);
    for (int a = 0; a < 3; ++a) { cp(vel[a], faceDim(a), in->vel[a]); cp(collisionvel[a], faceDim(a), in->collisionvel[a]); }
    for (int s = 0; s < 7; ++s) {   // Solver.cpp:86-152
        const Dim d = s == 0 ? centerDim() : (s <= 3 ? faceDim(s - 1) : edgeDim(s - 4));
        labels[s].init(d, PS_UNASSIGNED);
        activeIdx[s].init(d, PS_UNASSIGNED);
        reducedIdx[s].init(d, PS_UNASSIGNED);
    }
    regionCount = 0;
    std::memset(&stats, 0, sizeof(stats));
    stats.result = PS_INCOMPLETE;
    return PS_SUCCESS;
}

// HDK_PolyStokes.C:344-476
int Oracle::setup(const ps_params* p, const ps_fields_in* in) {
    const int rc = load(p, in);
    if (rc != PS_SUCCESS) return rc;
    const std::clock_t c0 = std::clock();
    const auto w0 = std::chrono::high_resolution_clock::now();
    // PS_ORACLE_TIMING=1: wall time of every setup stage on stderr (where the restatement's setup spends its time: bench.py's CPU leg)
    const bool timing = std::getenv("PS_ORACLE_TIMING") != nullptr;
    auto tl = std::chrono::high_resolution_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::high_resolution_clock::now();
        std::fprintf(stderr, "[oracle] %-44s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tl).count());
        tl = t;
    };
    buildIntegrationWeightsAlt(in); lap("buildIntegrationWeightsAlt");
    classifyCells();
    if (P.doReducedRegions) constructReducedRegions(); else constructOnlyActiveRegions();
    classifyFaces();
    classifyEdges(); lap("classify cells / regions / faces / edges");
    if (P.doReducedRegions) {
        constructCenterReducedIndices();
        constructFacesReducedIndices();
        constructEdgesReducedIndices();
    }
    lap("reduced indices (components, fixes)");
    constructActiveIndices(); lap("constructActiveIndices");
    if (P.doReducedRegions) {
        computeCenterOfMasses();
        computeLeastSquaresFits(); lap("COM + least-squares fits");
        computeReducedMassMatrices(); lap("computeReducedMassMatrices");
        computeReducedViscosityMatricesInteriorOnly(); lap("computeReducedViscosityMatricesInteriorOnly");
    } else {
        COM.clear(); cfit.clear(); Mr.clear(); K.clear();
    }
    constructMatrixBlocks(); lap("constructMatrixBlocks");
    // HDK_PolyStokes.C:462-467: initializeGuessVectors(); if (getUseWarmStart()) constructGuessVectors();
    constructGuessVectors(); lap("constructGuessVectors");
    assembleSystemPressureStressFactored(); lap("assembleSystemPressureStressFactored");
    if (P.preconditioner == PS_PRE_DIAGONAL || P.preconditioner == PS_PRE_CHEBYSHEV || P.preconditioner == PS_PRE_CHEBYSHEV_F32) buildJacobiDiagonal();
    if (P.preconditioner == PS_PRE_CHEBYSHEV || P.preconditioner == PS_PRE_CHEBYSHEV_F32) estimateLambdaMax();
    lap("preconditioner");
    const auto w1 = std::chrono::high_resolution_clock::now();
    stats.solveData[4] = 1000.0 * (double)(std::clock() - c0) / CLOCKS_PER_SEC;
    stats.solveData[5] = std::chrono::duration<double, std::milli>(w1 - w0).count();
    // dimData, Solver.cpp:578-593
    double* dd = stats.dimData;
    dd[0] = (double)nCenter; dd[1] = (double)nFace[0]; dd[2] = (double)nFace[1]; dd[3] = (double)nFace[2];
    dd[4] = (double)nEdge[0]; dd[5] = (double)nEdge[1]; dd[6] = (double)nEdge[2];
    dd[7] = (double)nActiveVs; dd[8] = (double)nFace[0]; dd[9] = (double)nFace[1]; dd[10] = (double)nFace[2];
    dd[11] = (double)nReducedVs; dd[12] = (double)nPressures; dd[13] = (double)nStresses;
    dd[14] = dd[15] = dd[16] = (double)nCenter;
    dd[17] = (double)nEdge[0]; dd[18] = (double)nEdge[1]; dd[19] = (double)nEdge[2];
    dd[20] = (double)nTotalDOFs; dd[21] = (double)nSystemSize; dd[22] = 1.; dd[23] = 0.;
    dd[24] = (double)regionCount; dd[25] = dx; dd[26] = dt;
    return PS_SUCCESS;
}

int Oracle::run(const ps_params* p, const ps_fields_in* in, bool doSolveStage) {
    int rc = setup(p, in);
    if (rc != PS_SUCCESS) { stats.result = rc; return rc; }
    int result = PS_INCOMPLETE;
    if (doSolveStage && P.doSolve) result = solve();
    buildValidFaces();
    for (int a = 0; a < 3; ++a) velOut[a] = vel[a];
    if (doSolveStage && (result == PS_SUCCESS || P.keepNonConvergedResults)) {
        recoverVelocityFromPressureStress();
        applySolutionToVelocity();
    }
    stats.result = result;
    registerArrays();
    return result;
}

void Oracle::registerArrays() {
    arrays.clear();
    static const char* sname[7] = {"center", "faceX", "faceY", "faceZ", "edgeYZ", "edgeXZ", "edgeXY"};
    auto reg = [&](const std::string& n, const void* p, int64_t c, int32_t e) { arrays[n] = ArrayRef{p, c, e}; };
    for (int s = 0; s < 7; ++s) {
        reg(std::string(sname[s]) + "LiquidWeights", liquidW[s].v.data(), (int64_t)liquidW[s].v.size(), 4);
        reg(std::string(sname[s]) + "FluidWeights", fluidW[s].v.data(), (int64_t)fluidW[s].v.size(), 4);
        reg(std::string(sname[s]) + "Labels", labels[s].v.data(), (int64_t)labels[s].v.size(), 4);
        reg(std::string(sname[s]) + "ActiveIndices", activeIdx[s].v.data(), (int64_t)activeIdx[s].v.size(), 4);
        reg(std::string(sname[s]) + "ReducedIndices", reducedIdx[s].v.data(), (int64_t)reducedIdx[s].v.size(), 4);
    }
    auto regv = [&](const std::string& n, const std::vector<double>& v) { reg(n, v.data(), (int64_t)v.size(), 8); };
    regv("reducedRegionCOM", COM);
    regv("reducedRegionBestFitVectors", cfit);
    regv("reducedRegionBestFitSystems", fitN); regv("reducedRegionBestFitRHS", fitRhs);
    regv("reducedMassMatrices", Mr);
    regv("reducedViscosityMatrices", K);
    regv("Inv_Mr_plus_2JDtuDJ", Binv);
    regv("reducedRHSVector", reducedRHS);
    regv("Mc", Mc); regv("McInv", McInv); regv("uInv", uInv); regv("u", u);
    regv("activeRHSVector", activeRHS); regv("pressureRHSVector", pressureRHS); regv("stressRHSVector", stressRHS);
    regv("oldActiveVs", oldActiveVs);
    regv("guessVector", guess);
    regv("b", b); regv("solutionVector", solution); regv("recoveredVelocity", recovered); regv("diagA", diagA);
    auto regcsr = [&](const std::string& n, const CSR& m) {
        reg(n + ".ptr", m.ptr.data(), (int64_t)m.ptr.size(), 8);
        reg(n + ".col", m.col.data(), (int64_t)m.col.size(), 4);
        reg(n + ".val", m.val.data(), (int64_t)m.val.size(), 8);
    };
    regcsr("G", G); regcsr("Dt", Dt); regcsr("JG", JG); regcsr("JDt", JDt); regcsr("A", A);
    static const char* ax[3] = {"X", "Y", "Z"};
    for (int a = 0; a < 3; ++a) {
        reg(std::string("vel") + ax[a], velOut[a].v.data(), (int64_t)velOut[a].v.size(), 4);
        reg(std::string("valid") + ax[a], valid[a].v.data(), (int64_t)valid[a].v.size(), 4);
    }
}

}  // namespace psoracle

// ---------------------------------------------------------------------------------------------
// C API (ctypes) — prefix po_ ("polystokes oracle")
// ---------------------------------------------------------------------------------------------
using psoracle::Oracle;
extern "C" {

void* po_create() { return new Oracle(); }
void po_destroy(void* h) { delete (Oracle*)h; }
void po_params_default(ps_params* p) {
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
int32_t po_run(void* h, const ps_params* p, const ps_fields_in* in, int32_t doSolve, ps_stats* stats) {
    Oracle* o = (Oracle*)h;
    const int rc = o->run(p, in, doSolve != 0);
    if (stats) *stats = o->stats;
    return rc;
}
const char* po_last_error(void* h) { return ((Oracle*)h)->err.c_str(); }
int64_t po_query_array(void* h, const char* name, int32_t* elem) {
    Oracle* o = (Oracle*)h;
    auto it = o->arrays.find(name);
    if (it == o->arrays.end()) return -1;
    if (elem) *elem = it->second.elem;
    return it->second.count;
}
int32_t po_read_array(void* h, const char* name, void* dst, int64_t bytes) {
    Oracle* o = (Oracle*)h;
    auto it = o->arrays.find(name);
    if (it == o->arrays.end()) return -1;
    const int64_t need = it->second.count * it->second.elem;
    if (bytes < need) return -2;
    if (need) std::memcpy(dst, it->second.ptr, (size_t)need);
    return 0;
}
int32_t po_apply_operator(void* h, const double* x, double* y, int32_t fair) {
    Oracle* o = (Oracle*)h;
    if (fair) o->applyOperatorFair(x, y); else o->applyOperator(x, y);
    return 0;
}
int32_t po_build_explicit_A(void* h) {
    Oracle* o = (Oracle*)h;
    o->assembleSystemPressureStress();
    o->registerArrays();
    return 0;
}
int32_t po_build_jacobi(void* h) {
    Oracle* o = (Oracle*)h;
    o->buildJacobiDiagonal();
    o->registerArrays();
    return 0;
}
// CPU baseline for bench.py: average wall ms of one operator application + one CG iteration's
// vector work (reference-shaped, `fair`=0; or fused, `fair`=1).
double po_time_cg_iterations(void* h, int32_t iters, int32_t fair) {
    Oracle* o = (Oracle*)h;
    const size_t n = (size_t)(o->nPressures + o->nStresses);
    std::vector<double> x(n, 0.), r = o->b, p = o->b, Ap(n);
    double rsold = 0; for (size_t i = 0; i < n; ++i) rsold += r[i] * r[i];
    const auto w0 = std::chrono::high_resolution_clock::now();
    for (int it = 0; it < iters; ++it) {
        if (fair) o->applyOperatorFair(p.data(), Ap.data()); else o->applyOperator(p.data(), Ap.data());
        double pAp = 0; for (size_t i = 0; i < n; ++i) pAp += p[i] * Ap[i];
        const double alpha = rsold / pAp;
        double rsnew = 0, xm = 0;
        for (size_t i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; }
        for (size_t i = 0; i < n; ++i) { rsnew += r[i] * r[i]; xm += x[i] * x[i]; }
        const double beta = rsnew / rsold;
        for (size_t i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
        rsold = rsnew;
        (void)xm;
    }
    const auto w1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::milli>(w1 - w0).count() / (double)iters;
}
void po_precondition(void* h, const double* r, double* z) {
    Oracle* o = (Oracle*)h;
    const size_t n = (size_t)(o->nPressures + o->nStresses);
    std::vector<double> in(r, r + n), out;
    o->precondition(in, out);
    std::copy(out.begin(), out.end(), z);
}
double po_cheb_lmax(void* h) { return ((Oracle*)h)->chebLmax; }
void po_set_exact_diagonal(void* h, int32_t on) { ((Oracle*)h)->exactDiagonal = on != 0; }
void po_set_setup_threads(void* h, int32_t n) { ((Oracle*)h)->setupThreads = n > 1 ? n : 1; }
int32_t po_reduced_dof(void) { return psoracle::RD; }
void po_basis(const double* off, int32_t axis, double* out) { psoracle::buildConversionCoefficients(off, axis, out); }
int32_t po_fullpivlu_solve(const double* N, const double* rhs, double* x) { return psoracle::fullPivLuSolve(N, rhs, x) ? 1 : 0; }
int32_t po_partialpiv_inverse(const double* B, double* Binv) { return psoracle::partialPivInverse(B, Binv) ? 1 : 0; }

}  // extern "C"
