// TEST INFRASTRUCTURE (see ps_oracle.hpp) — multi-threaded CPU baseline used ONLY by bench.py's cpu_baseline leg.
// Same operator as Oracle::applyOperatorFair ("fair CPU": one pass per block, no per-call matrix products,
// ApplyPressureStressMatrix.h:102-179 restated) and the vector work of pcg_external_matrix_A (pcg.h:311-335), with every
// row loop split over OpenMP threads (the reference itself runs three `omp sections` inside apply,
// ApplyPressureStressMatrix.h:122-154, and TBB over regions, util.h:163-168).  Reductions use OpenMP's order: this is a
// timing harness, not a parity path — the parity paths stay single-threaded and deterministic.
#include <chrono>
#include <omp.h>
#include "ps_oracle.hpp"

namespace psoracle {
namespace {
void mulRows(const CSR& M, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M.rows; ++r) {
        double s = 0;
        for (int64_t p = M.ptr[(size_t)r]; p < M.ptr[(size_t)r + 1]; ++p) s += M.val[(size_t)p] * x[M.col[(size_t)p]];
        y[r] = s;
    }
}
void mulRowsAdd(const CSR& M, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M.rows; ++r) {
        double s = 0;
        for (int64_t p = M.ptr[(size_t)r]; p < M.ptr[(size_t)r + 1]; ++p) s += M.val[(size_t)p] * x[M.col[(size_t)p]];
        y[r] += s;
    }
}
}  // namespace
}  // namespace psoracle

extern "C" double po_time_cg_iterations_mt(void* h, int32_t iters, int32_t threads, int32_t* threads_used) {
    using namespace psoracle;
    Oracle* o = (Oracle*)h;
    if (threads > 0) omp_set_num_threads(threads);
    int used = 1;
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
    if (threads_used) *threads_used = used;
    const int64_t nP = o->nPressures, nT = o->nStresses, nA = o->nActiveVs, nR = o->nReducedVs, R = o->regionCount;
    const int64_t n = nP + nT;
    // transposes of the reduced blocks, built once (setupMatrixVectorProducts builds GtJt / DJt the same way, :24-68)
    const CSR JGt = o->JG.transposed(), JDtt = o->JDt.transposed();
    std::vector<double> x((size_t)n, 0.), r = o->b, p = o->b, Ap((size_t)n), t((size_t)nA), t2((size_t)nA), w((size_t)nR), w2((size_t)nR), v((size_t)nR);
    double rsold = 0;
    for (int64_t i = 0; i < n; ++i) rsold += r[(size_t)i] * r[(size_t)i];
    const double dt = o->dt;
    const auto w0 = std::chrono::high_resolution_clock::now();
    for (int it = 0; it < iters; ++it) {
        mulRows(o->G, p.data(), t.data());
        mulRows(o->Dt, p.data() + nP, t2.data());
#pragma omp parallel for schedule(static)
        for (int64_t f = 0; f < nA; ++f) t[(size_t)f] = -dt * o->McInv[(size_t)f] * (t[(size_t)f] + t2[(size_t)f]);
        mulRows(o->Gt, t.data(), Ap.data());
        mulRows(o->D, t.data(), Ap.data() + nP);
        mulRows(o->JG, p.data(), w.data());
        mulRows(o->JDt, p.data() + nP, w2.data());
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < R; ++q)
            for (int m = 0; m < RD; ++m) {
                double s = 0;
                for (int k = 0; k < RD; ++k) s += o->Binv[(size_t)q * RD * RD + m * RD + k] * (w[(size_t)q * RD + k] + w2[(size_t)q * RD + k]);
                v[(size_t)q * RD + m] = -s;
            }
        mulRowsAdd(JGt, v.data(), Ap.data());
        mulRowsAdd(JDtt, v.data(), Ap.data() + nP);
        double pAp = 0;
#pragma omp parallel for schedule(static) reduction(+ : pAp)
        for (int64_t i = 0; i < n; ++i) {
            if (i >= nP) Ap[(size_t)i] += -0.5 * o->uInv[(size_t)(i - nP)] * p[(size_t)i];
            pAp += p[(size_t)i] * Ap[(size_t)i];
        }
        const double alpha = rsold / pAp;
        double rsnew = 0, xm = 0;
#pragma omp parallel for schedule(static) reduction(+ : rsnew, xm)
        for (int64_t i = 0; i < n; ++i) {
            x[(size_t)i] += alpha * p[(size_t)i];
            r[(size_t)i] -= alpha * Ap[(size_t)i];
            rsnew += r[(size_t)i] * r[(size_t)i];
            xm += x[(size_t)i] * x[(size_t)i];
        }
        const double beta = rsnew / rsold;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) p[(size_t)i] = r[(size_t)i] + beta * p[(size_t)i];
        rsold = rsnew;
        (void)xm;
    }
    const auto w1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::milli>(w1 - w0).count() / (double)iters;
}

// BASELINE.md section 2 "baseline A": the reference's own pass structure — applyMatrixVectorProducts with its three bodies
// under `#pragma omp parallel sections` (ApplyPressureStressMatrix.h:122-164, so at most 3 threads work inside an apply,
// and McInv*G / McInv*Dt are re-formed on every call) and the vector part of pcg_external_matrix_A as Eigen runs it
// (expression per line, single thread, pcg.h:311-335).  Returns ms per CG iteration.
extern "C" double po_time_cg_iterations_sections(void* h, int32_t iters, int32_t* threads_used) {
    using namespace psoracle;
    Oracle* o = (Oracle*)h;
    const int64_t n = o->nPressures + o->nStresses;
    omp_set_num_threads(3);
    int used = 1;
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
    if (threads_used) *threads_used = used;
    std::vector<double> x((size_t)n, 0.), r = o->b, p = o->b, Ap((size_t)n);
    auto dot = [&](const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (int64_t i = 0; i < n; ++i) s += a[(size_t)i] * b[(size_t)i]; return s; };
    double rsold = dot(r, r);
    const auto w0 = std::chrono::high_resolution_clock::now();
    for (int it = 0; it < iters; ++it) {
        std::vector<double> A11_1, A21_1, tp, tt, A12_1, A22_1;
#pragma omp parallel sections
        {
#pragma omp section
            o->applySection1(p.data(), A11_1, A21_1);
#pragma omp section
            o->applySection2(p.data(), tp, tt);
#pragma omp section
            o->applySection3(p.data(), A12_1, A22_1);
        }
        o->applyCombine(p.data(), A11_1, A21_1, tp, tt, A12_1, A22_1, Ap.data());
        const double alpha = rsold / dot(p, Ap);
        for (int64_t i = 0; i < n; ++i) x[(size_t)i] = x[(size_t)i] + alpha * p[(size_t)i];
        for (int64_t i = 0; i < n; ++i) r[(size_t)i] = r[(size_t)i] - alpha * Ap[(size_t)i];
        const double rsnew = dot(r, r);
        const double xmag = dot(x, x);
        (void)xmag;
        const double beta = rsnew / rsold;
        for (int64_t i = 0; i < n; ++i) p[(size_t)i] = r[(size_t)i] + beta * p[(size_t)i];
        rsold = rsnew;
    }
    const auto w1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::milli>(w1 - w0).count() / (double)iters;
}
