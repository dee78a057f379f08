// Exported-system path: MatrixMarket import + general CSR PCG, and its C ABI entry point.
// Part of the single translation unit ps_solve.hip (included there, inside its anonymous namespace where noted).
#pragma once

// =====================================================================================================
// Exported-system path: solve a component set written by exportComponentMatrices()
// (exec/HDK_PolyStokesSolver.cpp:543-566) — Mat_G, Mat_Dt, Mat_JG, Mat_JDt, Mat_McInv, Mat_uInv,
// Mat_Inv_Mr_plus_2JDtuDJ, Vec_b — with the same PCG.  JG / JDt arrive materialised, so the operator is applied
// literally as in ApplyPressureStressMatrix.h:102-179 with general CSR SpMVs (a sub-wave group per row).
// =====================================================================================================
#include <fstream>
#include <sstream>

namespace {

struct HostCSR {
    int64_t rows = 0, cols = 0;
    std::vector<int32_t> ptr, col;
    std::vector<double> val;
};
// MatrixMarket "coordinate real general" as Eigen's saveMarket writes it (MarketIO.h:310-340): 1-based triplets;
// loadMarket semantics: setFromTriplets (duplicates summed, rows sorted by column).
HostCSR readMarketSparse(const std::string& fn) {
    std::ifstream in(fn.c_str());
    if (!in) throw Error("cannot open " + fn);
    std::string line;
    do { if (!std::getline(in, line)) throw Error("empty file " + fn); } while (!line.empty() && line[0] == '%');
    std::istringstream hs(line);
    int64_t R, Cc, N;
    if (!(hs >> R >> Cc >> N)) throw Error("bad size line in " + fn);
    // sizes come from a file: refuse what cannot be a matrix before allocating for it
    if (R < 0 || Cc < 0 || N < 0 || R >= 0x7fffffffLL || Cc >= 0x7fffffffLL || N >= 0x7fffffffLL) throw Error("bad sizes in " + fn);
    if (R == 0 || Cc == 0 ? N != 0 : N / R > Cc + 1 + (int64_t)1e6) throw Error("more entries than the matrix can hold in " + fn);
    struct T { int32_t r, c; double v; };
    std::vector<T> t((size_t)N);
    for (int64_t k = 0; k < N; ++k) {
        int64_t i, j; double v;
        if (!(in >> i >> j >> v)) throw Error("truncated " + fn);
        if (i < 1 || i > R || j < 1 || j > Cc) throw Error("index out of range in " + fn);
        t[(size_t)k] = {(int32_t)(i - 1), (int32_t)(j - 1), v};
    }
    std::stable_sort(t.begin(), t.end(), [](const T& a, const T& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    HostCSR M;
    M.rows = R; M.cols = Cc;
    M.ptr.assign((size_t)R + 1, 0);
    size_t p = 0;
    for (int64_t r = 0; r < R; ++r) {
        M.ptr[(size_t)r] = (int32_t)M.val.size();
        while (p < t.size() && t[p].r == r) {
            const int32_t c = t[p].c;
            double s = 0;
            while (p < t.size() && t[p].r == r && t[p].c == c) { s += t[p].v; ++p; }
            M.col.push_back(c); M.val.push_back(s);
        }
    }
    M.ptr[(size_t)R] = (int32_t)M.val.size();
    return M;
}
std::vector<double> readMarketVector(const std::string& fn) {   // "array real general", column major (MarketIO.h:349-371)
    std::ifstream in(fn.c_str());
    if (!in) throw Error("cannot open " + fn);
    std::string line;
    do { if (!std::getline(in, line)) throw Error("empty file " + fn); } while (!line.empty() && line[0] == '%');
    std::istringstream hs(line);
    int64_t R, Cc = 1;
    if (!(hs >> R)) throw Error("bad size line in " + fn);
    hs >> Cc;
    if (R < 0 || Cc < 0 || R >= 0x7fffffffLL || Cc > 1024 || R * Cc >= 0x7fffffffLL) throw Error("bad sizes in " + fn);
    std::vector<double> v((size_t)(R * Cc));
    for (auto& x : v) if (!(in >> x)) throw Error("truncated " + fn);
    return v;
}
HostCSR hcat(const HostCSR& A, const HostCSR& B) {   // concatenate_h (lib/include/util.h:442-459)
    if (A.rows != B.rows) throw Error("hcat: row mismatch");
    HostCSR M;
    M.rows = A.rows; M.cols = A.cols + B.cols;
    M.ptr.assign((size_t)A.rows + 1, 0);
    for (int64_t r = 0; r < A.rows; ++r) {
        M.ptr[(size_t)r] = (int32_t)M.val.size();
        for (int p = A.ptr[(size_t)r]; p < A.ptr[(size_t)r + 1]; ++p) { M.col.push_back(A.col[(size_t)p]); M.val.push_back(A.val[(size_t)p]); }
        for (int p = B.ptr[(size_t)r]; p < B.ptr[(size_t)r + 1]; ++p) { M.col.push_back((int32_t)(B.col[(size_t)p] + A.cols)); M.val.push_back(B.val[(size_t)p]); }
    }
    M.ptr[(size_t)A.rows] = (int32_t)M.val.size();
    return M;
}
HostCSR transpose(const HostCSR& A) {
    HostCSR T;
    T.rows = A.cols; T.cols = A.rows;
    T.ptr.assign((size_t)A.cols + 1, 0);
    for (int32_t c : A.col) T.ptr[(size_t)c + 1]++;
    for (int64_t c = 0; c < A.cols; ++c) T.ptr[(size_t)c + 1] += T.ptr[(size_t)c];
    T.col.resize(A.val.size()); T.val.resize(A.val.size());
    std::vector<int32_t> pos(T.ptr.begin(), T.ptr.end() - 1);
    for (int64_t r = 0; r < A.rows; ++r)
        for (int p = A.ptr[(size_t)r]; p < A.ptr[(size_t)r + 1]; ++p) {
            const int q = pos[(size_t)A.col[(size_t)p]]++;
            T.col[(size_t)q] = (int32_t)r; T.val[(size_t)q] = A.val[(size_t)p];
        }
    return T;
}
std::vector<double> diagOf(const HostCSR& A) {
    std::vector<double> d((size_t)A.rows, 0.);
    for (int64_t r = 0; r < A.rows; ++r)
        for (int p = A.ptr[(size_t)r]; p < A.ptr[(size_t)r + 1]; ++p) if (A.col[(size_t)p] == r) d[(size_t)r] = A.val[(size_t)p];
    return d;
}

struct GenCSR {   // general CSR on the device
    int64_t rows = 0, cols = 0, nnz = 0;
    DevBuf<int32_t> ptr, col;
    DevBuf<double> val;
    int tpr = 1;   // threads per row (power of two <= 64)
    void upload(const HostCSR& H, hipStream_t s) {
        rows = H.rows; cols = H.cols; nnz = (int64_t)H.val.size();
        ptr.alloc(H.ptr.size()); col.alloc(H.col.size()); val.alloc(H.val.size());
        HIP_CHECK(hipMemcpyAsync(ptr.p, H.ptr.data(), H.ptr.size() * 4, hipMemcpyHostToDevice, s));
        if (nnz) {
            HIP_CHECK(hipMemcpyAsync(col.p, H.col.data(), H.col.size() * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(val.p, H.val.data(), H.val.size() * 8, hipMemcpyHostToDevice, s));
        }
        const double avg = rows ? (double)nnz / (double)rows : 0.;
        tpr = 1;
        while (tpr < 64 && tpr < avg) tpr <<= 1;
    }
};
// y[row] = beta*y[row] + alpha * scale[row] * (M x)[row]; TPR threads cooperate on a row (CSR-vector), shuffle reduce
template <int TPR>
__global__ void __launch_bounds__(BS) k_gen_spmv(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                 const double* __restrict__ x, int rows, double alpha, const double* __restrict__ scale,
                                                 double beta, double* __restrict__ y) {
    const int gid = blockIdx.x * BS + threadIdx.x;
    const int row = gid / TPR, sub = gid % TPR;
    double s = 0.;
    if (row < rows)
        for (int p = ptr[row] + sub; p < ptr[row + 1]; p += TPR) s += val[p] * x[col[p]];
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) s += __shfl_down(s, o, TPR);
    if (row < rows && sub == 0) {
        double v = alpha * s;
        if (scale) v *= scale[row];
        y[row] = (beta != 0. ? beta * y[row] : 0.) + v;
    }
}
void genSpmv(const GenCSR& M, const double* x, double alpha, const double* scale, double beta, double* y, hipStream_t st) {
    if (M.rows == 0) return;
    const int64_t threads = M.rows * M.tpr;
    const dim3 gr(gridFor(threads, BS)), bl(BS);
#define PS_GEN(T_) hipLaunchKernelGGL(k_gen_spmv<T_>, gr, bl, 0, st, M.ptr.p, M.col.p, M.val.p, x, (int)M.rows, alpha, scale, beta, y)
    switch (M.tpr) { case 1: PS_GEN(1); break; case 2: PS_GEN(2); break; case 4: PS_GEN(4); break; case 8: PS_GEN(8); break;
                     case 16: PS_GEN(16); break; case 32: PS_GEN(32); break; default: PS_GEN(64); }
#undef PS_GEN
}
// v[26 r + m] = sum_n BInv[r][m][n] w[26 r + n]
__global__ void k_binv_apply(const double* __restrict__ Binv, const double* __restrict__ w, double* __restrict__ v, int64_t nR) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nR) return;
    const int64_t r = i / PS_RD;
    const int m = (int)(i % PS_RD);
    const double* B = Binv + r * PS_RD * PS_RD + m * PS_RD;
    double s = 0.;
#pragma unroll
    for (int n = 0; n < PS_RD; ++n) s += B[n] * w[r * PS_RD + n];
    v[i] = s;
}
// y[nP + i] -= 0.5 uInv[i] x[nP + i]
__global__ void k_uinv_term(const double* __restrict__ uInv, const double* __restrict__ x, double* __restrict__ y, int64_t nP, int64_t nT) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nT) y[nP + i] -= 0.5 * uInv[i] * x[nP + i];
}
// Jacobi diagonal of the imported operator (thread per column, over the transposed blocks)
__global__ void k_gen_jacobi(const int32_t* __restrict__ ctp, const int32_t* __restrict__ ctc, const double* __restrict__ ctv,
                             const int32_t* __restrict__ jtp, const int32_t* __restrict__ jtc, const double* __restrict__ jtv, int n, int nP,
                             double dt, const double* __restrict__ McInv, const double* __restrict__ uInv, const double* __restrict__ Binv,
                             double* __restrict__ dinv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double diag = 0.;
    for (int p = ctp[j]; p < ctp[j + 1]; ++p) diag += -dt * McInv[ctc[p]] * ctv[p] * ctv[p];
    double q[PS_RD];
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
        const double* B = Binv + (int64_t)cur * PS_RD * PS_RD;
        double s = 0.;
        for (int m = 0; m < PS_RD; ++m) { double t = 0.; for (int k = 0; k < PS_RD; ++k) t += B[m * PS_RD + k] * q[k]; s += q[m] * t; }
        diag -= s;
    };
    for (int p = jtp[j]; p < jtp[j + 1]; ++p) {
        const int r = jtc[p] / PS_RD, m = jtc[p] % PS_RD;
        if (r != cur) { flush(); cur = r; for (int k = 0; k < PS_RD; ++k) q[k] = 0.; }
        q[m] += jtv[p];
    }
    flush();
    if (j >= nP) diag += -0.5 * uInv[j - nP];
    dinv[j] = diag != 0. ? 1. / diag : 1.;
}

}  // namespace

extern "C" int32_t ps_solve_exported_system(ps_context* c, const char* prefix, const ps_params* params, double dt, double* x_out,
                                            int64_t x_len, ps_stats* stats) {
    if (!c || !prefix || !params) return PS_FAILED;
    ps::SinkScope sinkScope_(&c->deferred);
    try {
        HIP_CHECK(hipSetDevice(c->device));
        const std::string pre(prefix);
        HostCSR G = readMarketSparse(pre + "Mat_G.mtx"), Dt = readMarketSparse(pre + "Mat_Dt.mtx");
        HostCSR JG = readMarketSparse(pre + "Mat_JG.mtx"), JDt = readMarketSparse(pre + "Mat_JDt.mtx");
        HostCSR McInvM = readMarketSparse(pre + "Mat_McInv.mtx"), uInvM = readMarketSparse(pre + "Mat_uInv.mtx");
        HostCSR BinvM = readMarketSparse(pre + "Mat_Inv_Mr_plus_2JDtuDJ.mtx");
        std::vector<double> bh = readMarketVector(pre + "Vec_b.mtx");
        const int64_t nA = G.rows, nP = G.cols, nT = Dt.cols, n = nP + nT, nR = JG.rows, R = nR / PS_RD;
        if (Dt.rows != nA || JG.cols != nP || JDt.cols != nT || JDt.rows != nR || nR % PS_RD) throw Error("inconsistent block sizes");
        if (McInvM.rows != nA || uInvM.rows != nT || BinvM.rows != nR || (int64_t)bh.size() != n) throw Error("inconsistent diagonal / rhs sizes");
        if (x_out && x_len < n) throw Error("x_out too small");
        // blocks (setupMatrixVectorProducts, ApplyPressureStressMatrix.h:24-68): cat_G_Dt, its transpose, cat_JG_JDt, its transpose
        HostCSR C = hcat(G, Dt), J = hcat(JG, JDt);
        HostCSR Ct = transpose(C), Jt = transpose(J);
        std::vector<double> mc = diagOf(McInvM), ui = diagOf(uInvM), bi((size_t)R * PS_RD * PS_RD, 0.);
        for (int64_t r = 0; r < nR; ++r)
            for (int p = BinvM.ptr[(size_t)r]; p < BinvM.ptr[(size_t)r + 1]; ++p) {
                const int64_t cc = BinvM.col[(size_t)p];
                if (cc / PS_RD != r / PS_RD) throw Error("Mat_Inv_Mr_plus_2JDtuDJ is not block diagonal");
                bi[(size_t)((r / PS_RD) * PS_RD * PS_RD + (r % PS_RD) * PS_RD + cc % PS_RD)] = BinvM.val[(size_t)p];
            }
        hipStream_t st = c->stream;
        GenCSR dC, dCt, dJ, dJt;
        dC.upload(C, st); dCt.upload(Ct, st); dJ.upload(J, st); dJt.upload(Jt, st);
        DevBuf<double> dMc, dUi, dBi, db, dx, dr, dp, dAp, ds, dw, dv, ddinv, part;
        DevBuf<CGScalars> dsc;
        auto up = [&](DevBuf<double>& d, const std::vector<double>& h) { d.alloc(h.size()); if (!h.empty()) HIP_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, st)); };
        up(dMc, mc); up(dUi, ui); up(dBi, bi); up(db, bh);
        dx.alloc((size_t)n); dr.alloc((size_t)n); dp.alloc((size_t)n); dAp.alloc((size_t)n);
        ds.alloc((size_t)nA + 1); dw.alloc((size_t)nR + 1); dv.alloc((size_t)nR + 1); dsc.alloc(1);
        part.alloc(3 * VGRID + 16);
        const bool jac = params->preconditioner == PS_PRE_DIAGONAL;
        if (jac) {
            ddinv.alloc((size_t)n);
            hipLaunchKernelGGL(k_gen_jacobi, dim3(gridFor(n, 128)), dim3(128), 0, st, dCt.ptr.p, dCt.col.p, dCt.val.p, dJt.ptr.p, dJt.col.p, dJt.val.p,
                               (int)n, (int)nP, dt, dMc.p, dUi.p, dBi.p, ddinv.p);
        }
        auto apply = [&](const double* x, double* y) {   // y = A x
            genSpmv(dC, x, dt, dMc.p, 0., ds.p, st);                 // s = dt McInv [G Dt] x
            genSpmv(dCt, ds.p, -1., nullptr, 0., y, st);             // y = -[G Dt]^T s
            if (nR > 0) {
                genSpmv(dJ, x, 1., nullptr, 0., dw.p, st);           // w = [JG JDt] x
                hipLaunchKernelGGL(k_binv_apply, dim3(gridFor(nR, BS)), dim3(BS), 0, st, dBi.p, dw.p, dv.p, nR);
                genSpmv(dJt, dv.p, -1., nullptr, 1., y, st);         // y -= [JG JDt]^T BInv w
            }
            if (nT > 0) hipLaunchKernelGGL(k_uinv_term, dim3(gridFor(nT, BS)), dim3(BS), 0, st, dUi.p, x, y, nP, nT);
        };
        const auto w0 = std::chrono::high_resolution_clock::now();
        const int vb = dotBlocks(n);
        const double* dvp = jac ? ddinv.p : nullptr;
        const int maxit = params->maxSolverIterations;
        hipLaunchKernelGGL(k_cg_init, dim3(vb), dim3(BS), 0, st, db.p, dvp, dx.p, dr.p, dp.p, n, part.p);
        hipLaunchKernelGGL(k_cg_scal0, dim3(1), dim3(BS), 0, st, dsc.p, part.p, vb, params->tolerance, maxit, 1);
        CGScalars h{};
        int it = 0;
        bool finished = false;
        while (it < maxit && !finished) {
            const int upto = std::min(maxit, it + 25);
            for (; it < upto; ++it) {
                apply(dp.p, dAp.p);
                hipLaunchKernelGGL(k_dot, dim3(vb), dim3(BS), 0, st, dp.p, dAp.p, n, part.p);
                hipLaunchKernelGGL(k_cg_scal1, dim3(1), dim3(BS), 0, st, dsc.p, part.p, vb);
                hipLaunchKernelGGL(k_cg_update_xr, dim3(vb), dim3(BS), 0, st, dsc.p, dp.p, dAp.p, dvp, dx.p, dr.p, n, part.p);
                hipLaunchKernelGGL(k_cg_scal2, dim3(1), dim3(BS), 0, st, dsc.p, part.p, vb, jac ? 1 : 0, it);
                hipLaunchKernelGGL(k_cg_update_p, dim3(vb), dim3(BS), 0, st, dsc.p, dr.p, dvp, dp.p, n);
            }
            HIP_CHECK(hipMemcpyAsync(&h, dsc.p, sizeof(h), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (h.done) finished = true;
        }
        const int iters = h.done ? h.iter : maxit;
        if (x_out) { HIP_CHECK(hipMemcpyAsync(x_out, dx.p, (size_t)n * 8, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st)); }
        const auto w1 = std::chrono::high_resolution_clock::now();
        const int result = iters == maxit ? PS_NOCONVERGE : PS_SUCCESS;   // the BiCGStab fallback is not wired into this path
        if (stats) {
            std::memset(stats, 0, sizeof(*stats));
            stats->dimData[7] = (double)nA; stats->dimData[11] = (double)nR; stats->dimData[12] = (double)nP; stats->dimData[13] = (double)nT;
            stats->dimData[21] = (double)n; stats->dimData[24] = (double)R; stats->dimData[26] = dt;
            stats->solveData[0] = std::sqrt(h.rre); stats->solveData[1] = iters;
            stats->solveData[3] = std::chrono::duration<double, std::milli>(w1 - w0).count();
            stats->result = result;
        }
        return result;
    } catch (const ps::Error& e) { c->err = e.msg; return PS_FAILED; }
    catch (const std::exception& e) { c->err = e.what(); return PS_FAILED; }
}

