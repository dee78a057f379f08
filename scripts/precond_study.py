"""CPU study (scipy, oracle blocks): CG iteration / operator-apply counts of candidate preconditioners on the factored
pressure-stress operator.  usage: precond_study.py <scene> <res> [tile]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from oracle import ps_oracle
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc, p = getattr(scenes, name)(n, tile=tile)
o = ps_oracle.Oracle(); o.run(sc, p, solve=False)
G, Dt, JG, JDt = (o.csr(k) for k in ("G", "Dt", "JG", "JDt"))
nP, nT = o.nP, o.nT; N = nP + nT
S = sp.hstack([G, Dt]).tocsr(); J = sp.hstack([JG, JDt]).tocsr()
Mc = o.array("McInv"); U = np.concatenate([np.zeros(nP), o.array("uInv")])
R = o.nRegions
Bi = sp.block_diag([b for b in o.array("Inv_Mr_plus_2JDtuDJ").reshape(R, 26, 26)]).tocsr() if R else sp.csr_matrix((0, 0))
dt = sc.dt
St, Jt = S.T.tocsr(), J.T.tocsr()
napply = [0]
def A(x):   # positive definite form: -A of the reference
    napply[0] += 1
    y = dt * (St @ (Mc * (S @ x))) + 0.5 * U * x
    if R: y += Jt @ (Bi @ (J @ x))
    return y
b = -o.array("b")
# diagonal
diag = dt * np.asarray(S.multiply(S).T @ Mc).ravel() + 0.5 * U
if R:
    JB = (Bi @ J).tocsr()
    diag += np.asarray(J.multiply(JB).sum(axis=0)).ravel()
dinv = np.where(diag != 0, 1.0 / np.where(diag != 0, diag, 1.0), 1.0)
def pcg(prec, tol=1e-3, maxit=20000):
    x = np.zeros(N); r = b.copy(); z = prec(r); pv = z.copy(); rs = r @ z
    napply[0] = 0
    for i in range(maxit):
        Ap = A(pv); al = rs / (pv @ Ap); x += al * pv; r -= al * Ap
        rr = r @ r; xx = x @ x
        if min(rr, rr / xx) < tol * tol: return i, napply[0]
        z = prec(r); rsn = r @ z; pv = z + (rsn / rs) * pv; rs = rsn
    return maxit, napply[0]
def lam_max(its=30):
    v = np.random.RandomState(0).standard_normal(N)
    for _ in range(its):
        w = dinv * A(v); lam = np.linalg.norm(w) / np.linalg.norm(v); v = w / np.linalg.norm(w)
    return lam
def cheb(k, lmax, ratio):
    lmin = lmax / ratio
    th, de = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    def prec(r):   # k steps of Chebyshev iteration on D^-1 A z = D^-1 r from z = 0 (k-1 operator applies)
        rho = de / th * 0 + 1.0 / (th / de)   # sigma^-1
        sig = th / de
        rho = 1.0 / sig
        d = (dinv * r) / th
        z = d.copy()
        for j in range(1, k):
            rho_n = 1.0 / (2 * sig - rho)
            res = dinv * (r - A(z))
            d = rho_n * rho * d + (2 * rho_n / de) * res
            z = z + d; rho = rho_n
        return z
    return prec
print(name, n, "N", N, "regions", R, flush=True)
t0 = time.time(); print("identity", pcg(lambda r: r), round(time.time() - t0, 1), flush=True)
t0 = time.time(); print("jacobi  ", pcg(lambda r: dinv * r), round(time.time() - t0, 1), flush=True)
lm = lam_max(); print("lambda_max(D^-1 A) ~", lm, flush=True)
for k in (2, 3, 4, 6):
    for ratio in (10, 30, 100):
        it, ap = pcg(cheb(k, 1.1 * lm, ratio))
        print("cheb k=%d ratio=%d: iters %d applies %d" % (k, ratio, it, ap), flush=True)
