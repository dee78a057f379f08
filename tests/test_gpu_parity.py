"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bit-exact for every integer / index / label / weight array; stated fp64 tolerances elsewhere.
"""
import os

import numpy as np
import pytest

from polystokes_amd import _abi as abi
from polystokes_amd import scenes

from helpers import materialise_blocks, relerr

pytestmark = pytest.mark.gpu

SCENES = {
    "cavity32": lambda: scenes.cavity(32),
    "cavity20_t10_p1": lambda: scenes.cavity(20, tile=10, pad=1),
    "beam32_uniform": lambda: scenes.beam(32),
    "coil48": lambda: scenes.coil(48),
    "blob0": lambda: scenes.blob(seed=0),
    "blob1_t7": lambda: scenes.blob(30, 26, 22, seed=1, tile=7, pad=2),
    "blob2_notile": lambda: _notile(scenes.blob(seed=2)),
    "spheres40": lambda: scenes.spheres(40, tile=8),
    "droplet24": lambda: scenes.droplet(24),
    "cavity33_linear": lambda: _linear(scenes.cavity(33, tile=11)),
    "blob3_L3S3": lambda: _layers(scenes.blob(28, 28, 28, seed=3, tile=9, pad=2), 3, 3),
    "blob4_L1S0": lambda: _layers(scenes.blob(seed=4), 1, 0),
}


def _notile(sp):
    sp[1].doTile = 0
    return sp


def _linear(sp):
    sp[1].indexOrder = abi.ORDER_LINEAR
    return sp


def _layers(sp, L, S):
    sp[1].activeLiquidBoundaryLayerSize = L
    sp[1].activeSolidBoundaryLayerSize = S
    return sp


@pytest.fixture(scope="module")
def gpu():
    import polystokes_amd
    s = polystokes_amd.Solver(0)
    yield s
    s.close()


@pytest.fixture(scope="module", params=list(SCENES))
def pair(request, gpu, oracle_mod):
    sc, p = SCENES[request.param]()
    o = oracle_mod.Oracle()
    o.run(sc, p, solve=True)
    gpu.upload(sc, p)
    gpu.setup()
    return request.param, sc, p, o, gpu


def test_setup_integer_state_is_bit_exact(pair):
    name, sc, p, o, g = pair
    assert list(g.stats.dimData) == list(o.stats.dimData), name
    for s in abi.SAMPLE_NAMES:
        for kind in ("LiquidWeights", "FluidWeights", "Labels", "ActiveIndices", "ReducedIndices"):
            a, b = g.array(s + kind), o.array(s + kind)
            assert a.dtype == b.dtype and np.array_equal(a, b), f"{name}: {s}{kind} differs in {(a != b).sum()} entries"


def test_tile_blocks_match(pair):
    name, sc, p, o, g = pair
    if o.nRegions == 0:
        pytest.skip("no reduced regions")
    assert np.array_equal(g.array("reducedRegionCOM"), o.array("reducedRegionCOM"))   # exact integer sums
    tol = 1e-12
    assert relerr(g.array("reducedMassMatrices"), o.array("reducedMassMatrices")) < tol
    assert relerr(g.array("reducedViscosityMatrices"), o.array("reducedViscosityMatrices")) < tol
    R = o.nRegions
    # c_fit solves an ill-conditioned LSQ system and BInv inverts B: compare through what they feed
    Mr = o.array("reducedMassMatrices").reshape(R, 26, 26)
    K = o.array("reducedViscosityMatrices").reshape(R, 26, 26)
    Bi = g.array("Inv_Mr_plus_2JDtuDJ").reshape(R, 26, 26)
    Bo = o.array("Inv_Mr_plus_2JDtuDJ").reshape(R, 26, 26)
    for r in range(R):
        B = Mr[r] / sc.dt + 2 * K[r]
        assert np.abs(Bi[r] @ B - np.eye(26)).max() < 1e-7
        assert relerr(Bi[r], Bo[r]) < 1e-6
    assert relerr(g.array("reducedRHSVector"), o.array("reducedRHSVector")) < 1e-8


def test_stencil_blocks_match(pair):
    name, sc, p, o, g = pair
    for nm in ("McInv", "activeRHSVector", "uInv", "pressureRHSVector", "stressRHSVector"):
        assert relerr(g.array(nm), o.array(nm)) < 1e-13, (name, nm)
    G, Dt, JG, JDt = materialise_blocks(g)
    for nm, M in (("G", G), ("Dt", Dt), ("JG", JG), ("JDt", JDt)):
        Mo = o.csr(nm)
        assert M.shape == Mo.shape, (name, nm, M.shape, Mo.shape)
        d = abs(M - Mo)
        scale = abs(Mo).max() if Mo.nnz else 1.0
        assert (d.max() if d.nnz else 0.0) <= 1e-12 * scale, (name, nm)
    # G and Dt have the same sparsity pattern, entry for entry
    assert np.array_equal(G.indptr, o.csr("G").indptr) and np.array_equal(G.indices, o.csr("G").indices)
    S, St = g.S_matrices()
    assert abs(S.T - St).max() == 0.0 if S.nnz else True


def test_rhs_and_operator_match(pair):
    name, sc, p, o, g = pair
    n = o.nP + o.nT
    if n == 0:
        pytest.skip("empty system")
    assert relerr(g.array("b"), o.array("b")) < 1e-9, name
    rng = np.random.RandomState(7)
    for _ in range(2):
        x = rng.randn(n)
        yo, yg = o.apply(x), g.apply(x)
        assert relerr(yg, yo) < 1e-10, name
    # symmetry / definiteness of the device operator itself
    x, y = rng.randn(n), rng.randn(n)
    Ax, Ay = g.apply(x), g.apply(y)
    assert abs(x @ Ay - y @ Ax) < 1e-9 * abs(x @ Ay)
    assert x @ Ax < 0


def test_solve_matches(pair):
    name, sc, p, o, g = pair
    rc = g.solve()
    assert rc == o.result, name
    it_o, it_g = o.stats.solveData[1], g.stats.solveData[1]
    assert abs(it_g - it_o) <= max(2, 0.02 * it_o), (name, it_g, it_o)   # CG is order sensitive: +-2 %
    xo, xg = o.array("solutionVector"), g.array("solutionVector")
    if np.linalg.norm(xo) > 0:
        assert np.linalg.norm(xg - xo) <= 10 * p.tolerance * np.linalg.norm(xo), name   # SURVEY §8c tolerance
    vel, valid = g.download()
    for a in range(3):
        assert np.array_equal(valid[a].ravel(), o.array("valid" + "XYZ"[a])), name
        vo = o.array("vel" + "XYZ"[a])
        scale = max(np.abs(vo).max(), 1e-30)
        assert np.abs(vel[a].ravel() - vo).max() <= 20 * p.tolerance * scale, name


@pytest.mark.parametrize("scene", ["spheres48", "spheres64", "coil48"])
def test_velocity_converges_to_the_oracle_with_the_tolerance(gpu, oracle_mod, scene):
    """The deliverable is the velocity (Solver.cpp:492-510: u = dt McInv (rhs/dt - [G Dt] x), :937-1028 write-back).  On stiff scenes
    (mu = 1e4 spheres, mu = 100 coil) that recovery differences 1e5-sized terms, so two solves that both satisfy the reference's stop
    rule at tol 1e-3 agree in x to 10 tol and may still differ by per cents in u ("AMP").  This pins that the
    spread IS the stop rule and not the recovery / write-back kernels: as the tolerance is tightened the HIP velocities converge to the
    oracle's: at tol 1e-8 the max-norm difference is <= 1e-4 of the largest velocity (1e-3 on the 64^3 spheres) and at least ten times
    smaller than on the worst rung of the ladder."""
    make = {"spheres48": lambda: scenes.spheres(48), "spheres64": lambda: scenes.spheres(64), "coil48": lambda: scenes.coil(48)}[scene]
    errs = []
    for tol in ((1e-4, 1e-6, 1e-8) if scene != "spheres64" else (1e-4, 1e-8)):   # (the 64^3 oracle solves are the slow ones)
        sc, p = make()
        p.tolerance = tol
        p.maxSolverIterations = 200000
        o = oracle_mod.Oracle()
        assert o.run(sc, p) == abi.SUCCESS
        assert gpu.step(sc, p) == abi.SUCCESS
        it_o, it_g = o.stats.solveData[1], gpu.stats.solveData[1]
        assert abs(it_g - it_o) <= max(2, 0.05 * it_o), (scene, tol, it_g, it_o)   # (thousands of iterations at 1e-8: the +-2 % of the standard tolerance widens)
        e = 0.0
        for a in range(3):
            vo = o.array("vel" + "XYZ"[a])
            e = max(e, np.abs(gpu.vel[a].ravel() - vo).max() / max(np.abs(vo).max(), 1e-30))
        errs.append(e)
    floor = 2e-6                                             # fp32 output fields
    # measured (r03): spheres48 3e-2 -> 4e-4 -> 4e-6, coil48 below 1e-5 throughout, spheres64 0.21 -> 4.9e-4: the amplification grows with
    # the size of the stiff scene (about 5e4 x tol there), so its bound at 1e-8 is 1e-3
    assert errs[-1] <= (1e-3 if scene == "spheres64" else 1e-4), (scene, errs)
    # The ladder need not be monotone rung by rung (coil48: 3e-9 at 1e-4 — both solves stop after the same few iterations — then 6e-4
    # at 1e-6 and 3e-6 at 1e-8): what must hold is that the tightest solve is far better than the worst rung
    assert errs[-1] <= max(0.1 * max(errs[:-1]), floor), (scene, errs)


def test_jacobi_pcg_extension(gpu, oracle_mod):
    sc, p = scenes.blob(seed=5)
    p.preconditioner = abi.PRE_DIAGONAL
    o = oracle_mod.Oracle()
    o.run(sc, p)
    gpu.upload(sc, p)
    gpu.setup()
    dinv = gpu.array("dinv")
    do = o.array("diagA")
    nz = do != 0
    assert np.all(dinv[~nz] == 1.0)        # Eigen DiagonalPreconditioner convention for empty rows
    assert relerr(1.0 / dinv[nz], do[nz]) < 1e-9
    rc = gpu.solve()
    assert rc == o.result == abi.SUCCESS
    assert abs(gpu.stats.solveData[1] - o.stats.solveData[1]) <= max(2, 0.02 * o.stats.solveData[1])


def test_step_host_boundary_and_zero_rhs(gpu, oracle_mod):
    # polystokes_step == solveGasSubclass on host buffers; b == 0 guard (documented deviation)
    sc, p = scenes.cavity(20, tile=10)
    sc.vel[0][:] = 0
    rc = gpu.step(sc, p)
    assert rc == abi.SUCCESS and gpu.stats.solveData[1] == 0
    for a in range(3):
        assert np.all(gpu.vel[a] == 0)
    sc, p = scenes.droplet(24)
    for a, v in enumerate((0.25, -0.5, 0.125)):
        sc.vel[a][:] = v
    rc = gpu.step(sc, p)
    assert rc == abi.SUCCESS
    for a, v in enumerate((0.25, -0.5, 0.125)):
        ok = gpu.valid[a] > 0
        np.testing.assert_allclose(gpu.vel[a][ok], v, rtol=1e-6)


def test_noconverge_falls_back_to_bicgstab(gpu, oracle_mod):
    sc, p = scenes.blob(seed=6)
    p.maxSolverIterations = 5
    o = oracle_mod.Oracle()
    o.run(sc, p)
    rc = gpu.step(sc, p)
    assert o.stats.usedBiCGStab == 1 and gpu.stats.usedBiCGStab == 1
    assert rc == o.result


@pytest.mark.parametrize("scene,maxit,tol", [("blob6", 20, 1e-3), ("spheres24", 12, 1e-3), ("spheres24", 30, 1e-4), ("coil32", 30, 1e-3)])
def test_bicgstab_fallback_converges_like_the_oracle(gpu, oracle_mod, scene, maxit, tol):
    """A PCG that runs out of iterations falls back to bicgstab_external_matrix_A restarted from zero (Solver.cpp:784-799,
    pcg.h:134-200) — and that BiCGStab CONVERGES here: same verdict, same 0-based iteration index, same error measure
    (min(||e||^2, ||e|| / ||x||), pcg.h:190-196) and the same solution as the oracle's restatement."""
    if scene == "blob6":
        sc, p = scenes.blob(seed=6)
    elif scene == "coil32":
        sc, p = scenes.coil(32, tile=8)
    else:
        sc, p = scenes.spheres(24, tile=8)
    p.maxSolverIterations = maxit
    p.tolerance = tol
    o = oracle_mod.Oracle()
    o.run(sc, p)
    rc = gpu.step(sc, p)
    assert o.stats.usedBiCGStab == 1 and gpu.stats.usedBiCGStab == 1
    assert rc == o.result == abi.SUCCESS, (rc, o.result, o.stats.solveData[1])
    assert abs(int(gpu.stats.solveData[1]) - int(o.stats.solveData[1])) <= 1 and int(gpu.stats.solveData[1]) < maxit   # BiCGStab is erratic: +-1
    # BiCGStab amplifies rounding differences (the two implementations sum their dot products in other orders): after 7-38
    # iterations the error measures agree to 1e-3 ... 6e-2 relative, the iteration index exactly
    assert gpu.stats.solveData[0] < tol and o.stats.solveData[0] < tol
    assert abs(gpu.stats.solveData[0] - o.stats.solveData[0]) <= 0.25 * abs(o.stats.solveData[0])
    xo, xg = o.array("solutionVector"), gpu.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * tol * np.linalg.norm(xo)
    # (the bound that matters is the one on x: the velocities difference 1e5-sized terms and, after an erratic BiCGStab run,
    # differ by up to 1e-2 relative at tol 1e-4 although x agrees to 10 tol — DESIGN.md section 4, AMP)
    for a in range(3):
        assert np.array_equal(gpu.valid[a].ravel(), o.array("valid" + "XYZ"[a]))


@pytest.mark.parametrize("case", ["beam32", "beam16_nowarm", "blob_reduced"])
def test_solver_type_eigen_matches_oracle(gpu, oracle_mod, tmp_path, case):
    """solverType = EIGEN (BASELINE config 1: uniform 32^3 beam): Eigen's ConjugateGradient semantics — diagonal preconditioner,
    ||r||^2 < tol^2 ||b||^2, solveWithGuess from the warm-start vector (Solver.cpp:814-862, 512-531) — on the factored device
    operator, against the oracle's Eigen-CG restatement on its explicitly assembled A; and the explicit A itself (Mat_A.mtx,
    AssembleSystem.cpp:351-430) against the oracle's."""
    import scipy.io
    if case == "beam32":
        sc, p = scenes.beam(32)
    elif case == "beam16_nowarm":
        sc, p = scenes.beam(16)
        p.useWarmStart = 0
    else:
        sc, p = scenes.blob(16, 14, 18, seed=5, tile=6)
    p.solverType = abi.EIGEN
    p.tolerance = 1e-8
    p.maxSolverIterations = 20000
    o = oracle_mod.Oracle()
    o.run(sc, p)
    rc = gpu.step(sc, p)
    assert rc == o.result == abi.SUCCESS
    g, go = gpu.array("guessVector"), o.array("guessVector")
    if case == "beam16_nowarm":
        assert not go.any() and not g.any()
    else:
        assert np.abs(go).max() > 0 and np.abs(g - go).max() <= 1e-11 * np.abs(go).max()
    it, ito = int(gpu.stats.solveData[1]), int(o.stats.solveData[1])
    assert abs(it - ito) <= max(2, 0.02 * ito), (it, ito)
    assert gpu.stats.solveData[0] <= p.tolerance
    xo, xg = o.array("solutionVector"), gpu.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * p.tolerance * np.linalg.norm(xo)
    for a in range(3):
        vo = o.array("vel" + "XYZ"[a])
        assert np.abs(gpu.vel[a].ravel() - vo).max() <= 1e-4 * max(np.abs(vo).max(), 1e-30)
    pre = str(tmp_path) + "/e."
    gpu.export_matrices(pre)
    A = scipy.io.mmread(pre + "Mat_A.mtx").tocsr()
    Ao = o.csr("A")
    assert A.shape == Ao.shape and A.nnz > 0
    assert abs(A - Ao).max() <= 1e-9 * abs(Ao).max()
    assert abs(A - A.T).max() <= 1e-9 * abs(Ao).max()


@pytest.mark.parametrize("scene", ["cavity32", "coil32", "spheres32", "blob6", "cavity24_k2", "beam16_uniform_k6"])
def test_chebyshev_preconditioner_matches_oracle(gpu, oracle_mod, scene):
    """PS_PRE_CHEBYSHEV (SURVEY 8f-3 extension; the reference's own preconditioners are stubs / dead code, Preconditioners.cpp:4-41):
    the interval estimate, z = M^-1 r on a random r, the iteration count and the solution against the oracle's restatement;
    and the point of it — at least 2.5x fewer CG iterations than Jacobi at the default degree."""
    deg = 0
    if scene == "cavity32":
        sc, p = scenes.cavity(32)
    elif scene == "coil32":
        sc, p = scenes.coil(32, tile=8)
    elif scene == "spheres32":
        sc, p = scenes.spheres(32, tile=8)
    elif scene == "blob6":
        sc, p = scenes.blob(seed=6)
    elif scene == "cavity24_k2":
        (sc, p), deg = scenes.cavity(24, tile=12), 2
    else:
        (sc, p), deg = scenes.beam(16), 6
    p.preconditioner = abi.PRE_CHEBYSHEV
    p.preconditionerDegree = deg
    p.tolerance = 1e-6
    o = oracle_mod.Oracle()
    o.run(sc, p)
    rc = gpu.step(sc, p)
    assert rc == o.result == abi.SUCCESS
    r = np.random.RandomState(11).standard_normal(gpu.nP + gpu.nT)
    zo, zg = o.precondition(r), gpu.precondition(r)
    assert np.abs(zo - zg).max() <= 1e-10 * np.abs(zo).max()
    ito, itg = int(o.stats.solveData[1]), int(gpu.stats.solveData[1])
    assert abs(itg - ito) <= max(2, 0.02 * ito), (itg, ito)
    xo, xg = o.array("solutionVector"), gpu.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * p.tolerance * np.linalg.norm(xo)
    if deg == 0:
        p.preconditioner = abi.PRE_DIAGONAL
        assert gpu.step(sc, p) == abi.SUCCESS
        assert int(gpu.stats.solveData[1]) >= 2.5 * itg, (int(gpu.stats.solveData[1]), itg)


@pytest.mark.parametrize("scene", ["cavity32", "coil32", "spheres32", "blob6", "cavity24_k2", "cavity48_k6", "cavity24_k1"])
def test_chebyshev_f32_inner_vectors_match_oracle(gpu, oracle_mod, scene):
    """PS_PRE_CHEBYSHEV_F32 (r06): the polynomial with its inner vectors STORED as fp32, restated in the oracle first
    (ps_oracle_solve.cpp: chebyshev32 — rounding at the iterates and the active face rows; the tile rows' storage is not restatable there, so
    the comparison is to the rounding LEVEL).  z = M^-1 r within 5e-6 of max |z| (fp64 form: 1e-10), the iteration count within 2 % (or 2) of the
    oracle's AND of the product's own fp64 polynomial (growth <= 5 % accepted, none seen), x within 10 tol.  blob6 has a viscosity FIELD: its stress
    diagonal is not value-set coded and the two-unit kernels read it as fp64 (UC = false, r06) — the fp32 inner vectors run there too (array chebInner32)."""
    deg = 0
    if scene == "cavity32":
        sc, p = scenes.cavity(32)
    elif scene == "coil32":
        sc, p = scenes.coil(32, tile=8)
    elif scene == "spheres32":
        sc, p = scenes.spheres(32, tile=8)
    elif scene == "blob6":
        sc, p = scenes.blob(seed=6)
    elif scene == "cavity24_k2":
        (sc, p), deg = scenes.cavity(24, tile=12), 2
    elif scene == "cavity24_k1":
        (sc, p), deg = scenes.cavity(24, tile=12), 1          # one term: z = fl32(dinv r / theta), no inner apply
    else:
        (sc, p), deg = scenes.cavity(48), 6
    p.preconditionerDegree = deg
    p.tolerance = 1e-6
    p.preconditioner = abi.PRE_CHEBYSHEV
    assert gpu.step(sc, p) == abi.SUCCESS
    it64 = int(gpu.stats.solveData[1])
    assert int(gpu.array("chebInner32")[0]) == 0
    p.preconditioner = abi.PRE_CHEBYSHEV_F32
    o = oracle_mod.Oracle()
    o.run(sc, p)
    assert gpu.step(sc, p) == o.result == abi.SUCCESS
    ran32 = int(gpu.array("chebInner32")[0])
    assert ran32 == 1, scene
    assert int(gpu.array("diagonalsCoded")[0]) == (2 if scene == "blob6" else 3)          # bit 0: uInv coded, bit 1: McInv coded
    r = np.random.RandomState(11).standard_normal(gpu.nP + gpu.nT)
    zo, zg = o.precondition(r), gpu.precondition(r)
    assert int(gpu.array("chebInner32")[0]) == ran32
    assert np.abs(zo - zg).max() <= 5e-6 * np.abs(zo).max()
    if ran32:
        assert np.array_equal(zg, zg.astype(np.float32).astype(np.float64))          # the result IS an fp32 vector
        if deg != 1:
            assert np.abs(zo - zg).max() > 1e-12 * np.abs(zo).max()                 # ... and not the fp64 polynomial's (one term: the two roundings coincide)
    ito, itg = int(o.stats.solveData[1]), int(gpu.stats.solveData[1])
    assert abs(itg - ito) <= max(2, 0.02 * ito), (itg, ito)
    assert itg <= max(it64 + 2, 1.05 * it64), (itg, it64)
    xo, xg = o.array("solutionVector"), gpu.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * p.tolerance * np.linalg.norm(xo)
    # the stop rule holds on the TRUE fp64 residual (the recurrence residual does not drift away from it under an inexact preconditioner)
    b = gpu.array("b")
    res = b - gpu.apply(xg)
    assert min(res @ res, (res @ res) / (xg @ xg)) < p.tolerance ** 2 * 1.01


@pytest.mark.parametrize("scene", ["cavity32", "blob6_variable_viscosity"])
def test_jacobi_on_the_stored_diagonal_matches_oracle(gpu, oracle_mod, scene):
    """The Jacobi extension reads 1 / A_jj as the upper 16 bits of its fp32 value, rounded to nearest even (ps_common.hpp: diag_t);
    the oracle restates the rounding (ps_oracle_solve.cpp: storedDinv).  The stored values are EQUAL bit for bit (the rounding itself is
    0.4 %: a restatement without it would miss by that much), every one within 2^-8 of the fp64 reciprocal, z = M^-1 r equal, the
    iteration counts equal (± 1) and x within the tolerance."""
    sc, p = scenes.cavity(32) if scene == "cavity32" else scenes.blob(seed=6)
    p.preconditioner = abi.PRE_DIAGONAL
    p.tolerance = 1e-6
    o = oracle_mod.Oracle()
    o.run(sc, p)
    assert gpu.step(sc, p) == o.result == abi.SUCCESS
    n = gpu.nP + gpu.nT
    one = np.ones(n)
    zo, zg = o.precondition(one), gpu.precondition(one)          # the stored diagonal itself
    same = zo == zg                                               # (the two diagonals differ by 1e-9 before the rounding: an entry next to a
    assert same.mean() >= 0.9999                                  #  rounding midpoint may land on the other neighbour — one step of 2^-8)
    assert np.abs(zg / zo - 1.0).max() <= 2.0 ** -7
    dg = o.array("diagA")                                         # (negative: the system is assembled with the reference's signs; 0 -> 1 where a row is empty)
    exact = np.where(dg != 0., 1.0 / np.where(dg != 0., dg, 1.), 1.0)
    assert 1e-5 < np.abs(zg / exact - 1.0).max() <= 2.0 ** -8 * (1 + 1e-6)
    r = np.random.RandomState(5).standard_normal(n)
    assert np.array_equal(o.precondition(r)[same], gpu.precondition(r)[same])
    assert abs(int(gpu.stats.solveData[1]) - int(o.stats.solveData[1])) <= 1, (int(gpu.stats.solveData[1]), int(o.stats.solveData[1]))
    xo, xg = o.array("solutionVector"), gpu.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * p.tolerance * np.linalg.norm(xo)


@pytest.mark.parametrize("tol", [1e-3, 1e-6])
@pytest.mark.parametrize("scene", ["cavity32", "coil48", "blob6"])
def test_stored_diagonal_jacobi_is_equivalent_to_exact_jacobi(gpu, oracle_mod, scene, tol):
    """BASELINE config 3 says "Jacobi-PCG".  The product reads 1 / A_jj in 16 bits; THIS test compares it with the
    oracle's Jacobi-PCG on the EXACT fp64 diagonal (ps_oracle.Oracle.set_exact_diagonal: no restated rounding in the loop) — iterations
    within 2 % (or 2), x within 10 tol, at the shipped tolerance and at 1e-6.  (The same comparison at 5.9 M DOFs against a committed
    fixture: tests/test_golden.py::test_hip_jacobi_matches_the_exact_diagonal_oracle_at_real_size.)"""
    sc, p = {"cavity32": lambda: scenes.cavity(32), "coil48": lambda: scenes.coil(48), "blob6": lambda: scenes.blob(seed=6)}[scene]()
    p.preconditioner = abi.PRE_DIAGONAL
    p.tolerance = tol
    p.maxSolverIterations = 20000
    o = oracle_mod.Oracle()
    o.set_exact_diagonal(True)
    o.run(sc, p)
    assert gpu.step(sc, p) == o.result == abi.SUCCESS
    n = gpu.nP + gpu.nT
    dg = o.array("diagA")
    exact = np.where(dg != 0., 1.0 / np.where(dg != 0., dg, 1.), 1.0)
    assert np.array_equal(o.precondition(np.ones(n)), exact)                  # the oracle really is on the unrounded diagonal ...
    zg = gpu.precondition(np.ones(n))
    assert 1e-5 < np.abs(zg / exact - 1.0).max() <= 2.0 ** -8 * (1 + 1e-6)    # ... and the product on the 16-bit one
    ito, itg = int(o.stats.solveData[1]), int(gpu.stats.solveData[1])
    assert abs(itg - ito) <= max(2, 0.02 * ito), (scene, tol, itg, ito)
    xo, xg = o.array("solutionVector"), gpu.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * tol * np.linalg.norm(xo), (scene, tol)


def test_export_component_matrices_roundtrip(gpu, oracle_mod, tmp_path):
    """exportComponentMatrices / exportStats (Solver.cpp:543-606): MatrixMarket files with the reference's names,
    read back with scipy and compared with the oracle's blocks (reference numbering)."""
    import scipy.io
    sc, p = scenes.blob(seed=7)
    p.exportComponentMatrices = 1
    p.exportMatrices = 1
    o = oracle_mod.Oracle()
    o.run(sc, p)
    rc = gpu.step(sc, p)
    assert rc == o.result
    pre = str(tmp_path) + "/frame0001."
    gpu.export_component_matrices(pre)
    gpu.export_matrices(pre)
    gpu.export_stats(pre)
    # the reduced blocks: B = Mr/dt + 2K and its inverse are both in the file set; MrInv / A are written empty, the guess zero
    B = scipy.io.mmread(pre + "Mat_Mr_plus_2JDtuDJ.mtx").tocsr()
    Bi = scipy.io.mmread(pre + "Mat_Inv_Mr_plus_2JDtuDJ.mtx").tocsr()
    Mr = scipy.io.mmread(pre + "Mat_Mr.mtx").tocsr()
    K = scipy.io.mmread(pre + "Mat_JDtuDJ.mtx").tocsr()
    assert abs(B - (Mr / sc.dt + 2 * K)).max() <= 1e-12 * abs(B).max()
    nR = B.shape[0]
    assert nR == 26 * int(gpu.stats.dimData[24]) and nR > 0
    assert abs((B @ Bi) - __import__("scipy.sparse").sparse.identity(nR)).max() <= 1e-6
    assert scipy.io.mmread(pre + "Mat_MrInv.mtx").shape == (0, 0)
    n = int(gpu.stats.dimData[21])
    A = scipy.io.mmread(pre + "Mat_A.mtx")
    assert A.shape == (n, n) and A.nnz == 0
    # Vec_guess is the warm-start vector of constructGuessVectors (Solver.cpp:512-531, useWarmStart defaults to 1)
    g = np.asarray(scipy.io.mmread(pre + "Vec_guess.mtx")).ravel()
    go = o.array("guessVector")
    assert np.abs(go).max() > 0 and np.abs(g - go).max() <= 1e-11 * np.abs(go).max()
    for nm in ("G", "Dt", "JG", "JDt"):
        M = scipy.io.mmread(pre + "Mat_%s.mtx" % nm).tocsr()
        Mo = o.csr(nm)
        assert M.shape == Mo.shape
        d = abs(M - Mo)
        assert (d.max() if d.nnz else 0.0) <= 1e-12 * max(abs(Mo).max(), 1e-300), nm
    for nm, arr in (("Mat_McInv", "McInv"), ("Mat_uInv", "uInv"), ("Mat_Mc", "Mc"), ("Mat_u", "u")):
        M = scipy.io.mmread(pre + nm + ".mtx").tocsr()
        assert relerr(M.diagonal(), o.array(arr)) < 1e-13, nm
    for nm, arr in (("Vec_activeRHS", "activeRHSVector"), ("Vec_pressureRHS", "pressureRHSVector"),
                    ("Vec_stressRHS", "stressRHSVector"), ("Vec_b", "b")):
        v = scipy.io.mmread(pre + nm + ".mtx").ravel()
        assert relerr(v, o.array(arr)) < 1e-9, nm
    dd = scipy.io.mmread(pre + "dimData.mtx").ravel()
    assert list(dd) == list(o.stats.dimData)
    sd = scipy.io.mmread(pre + "solveData.mtx").ravel()
    assert sd[1] == gpu.stats.solveData[1]
    head = open(pre + "Mat_G.mtx").readline()
    assert head.startswith("%%MatrixMarket matrix coordinate  real general")     # MarketIO.h header text
    assert open(pre + "Vec_b.mtx").readline().startswith("%%MatrixMarket matrix array real general")


def test_no_liquid_and_all_solid(gpu, oracle_mod):
    """Edge cases: an empty domain (no liquid anywhere) and liquid entirely inside a solid."""
    n = 20
    for surface, collision in ((1.0, 1.0), (-1.0, -1.0)):
        sc = abi.Scene(n, n, n, 1.0 / n, 0.01, 1.0, [0.3, 0.0, 0.0], np.float32(surface), np.float32(collision), 1.0)
        p = abi.default_params()
        o = oracle_mod.Oracle()
        ro = o.run(sc, p)
        rg = gpu.step(sc, p)
        assert rg == ro
        assert list(gpu.stats.dimData) == list(o.stats.dimData)
        for a in range(3):
            assert np.array_equal(gpu.valid[a].ravel(), o.array("valid" + "XYZ"[a]))
            assert np.array_equal(gpu.vel[a].ravel(), o.array("vel" + "XYZ"[a]))


def test_input_weights_are_used_verbatim(gpu, oracle_mod):
    """The shim may hand over the 14 volume-fraction fields sampled by HDK itself (ps_fields_in.weights)."""
    sc, p = scenes.blob(seed=8)
    o = oracle_mod.Oracle()
    o.run(sc, p, solve=False)
    w = [o.array(s + "LiquidWeights") for s in abi.SAMPLE_NAMES] + [o.array(s + "FluidWeights") for s in abi.SAMPLE_NAMES]
    # hand them in under a different SDF: the result must follow the weights, not the SDF
    sh = abi.grid_shapes(sc.nx, sc.ny, sc.nz)
    sc2 = abi.Scene(sc.nx, sc.ny, sc.nz, sc.dx, sc.dt, sc.density, sc.vel, np.float32(5.0), np.float32(5.0), sc.viscosity,
                    collisionvel=sc.collisionvel, weights=[w[i].reshape(sh[abi.SAMPLE_NAMES[i % 7]]) for i in range(14)])
    o2 = oracle_mod.Oracle()
    o2.run(sc2, p)
    rc = gpu.step(sc2, p)
    assert rc == o2.result
    assert np.array_equal(gpu.array("centerLabels"), o.array("centerLabels"))
    assert np.array_equal(gpu.array("edgeXZActiveIndices"), o2.array("edgeXZActiveIndices"))
    assert abs(gpu.stats.solveData[1] - o2.stats.solveData[1]) <= 2


def test_non_dyadic_weights_take_the_fp64_stream(gpu, oracle_mod):
    """Volume fractions that are not multiples of 1/8 (a sampler other than the 2x2x2 one, e.g. HDK's own handed in through
    ps_fields_in.weights): the value coding fails its bit-for-bit check at setup and the pipelined kernels run on the
    fp64-value stream (16-bit windowed columns + fp64 values) — by itself, no environment switch.  Checked against the
    oracle given the same weights."""
    sc, p = scenes.blob(seed=8)
    o = oracle_mod.Oracle()
    o.run(sc, p, solve=False)
    rng = np.random.RandomState(3)
    w = []
    for s_ in abi.SAMPLE_NAMES:
        a = o.array(s_ + "LiquidWeights").copy()
        m = a > 0
        a[m] = np.clip(a[m] * rng.uniform(0.55, 1.0, m.sum()), 0.03, 1.0).astype(np.float32)   # arbitrary fractions, same support
        w.append(a)
    w += [o.array(s_ + "FluidWeights") for s_ in abi.SAMPLE_NAMES]
    sh = abi.grid_shapes(sc.nx, sc.ny, sc.nz)
    sc2 = abi.Scene(sc.nx, sc.ny, sc.nz, sc.dx, sc.dt, sc.density, sc.vel, sc.surface, sc.collision, sc.viscosity,
                    collisionvel=sc.collisionvel, weights=[w[i].reshape(sh[abi.SAMPLE_NAMES[i % 7]]) for i in range(14)])
    p.tolerance = 1e-6
    p.maxSolverIterations = 20000
    o2 = oracle_mod.Oracle()
    o2.run(sc2, p)
    rc = gpu.step(sc2, p)
    assert rc == o2.result == abi.SUCCESS
    assert int(gpu.array("valuesCoded")[0]) == 0 and int(gpu.array("columns16")[0]) == 3
    assert np.array_equal(gpu.array("faceXActiveIndices"), o2.array("faceXActiveIndices"))
    x = np.random.RandomState(4).standard_normal(gpu.nP + gpu.nT)
    yo, yg = o2.apply(x), gpu.apply(x)
    assert np.abs(yo - yg).max() <= 1e-10 * np.abs(yo).max()
    assert abs(gpu.stats.solveData[1] - o2.stats.solveData[1]) <= max(2, 0.02 * o2.stats.solveData[1])
    xs, xo = gpu.array("solutionVector"), o2.array("solutionVector")
    # two converged iterates of a system with arbitrary (badly scaled) volume fractions: they agree to a multiple of the stop
    # tolerance that depends on where each run's last iteration lands (observed 4 to 16 tol with different dot-product orders)
    assert np.linalg.norm(xs - xo) <= 30 * p.tolerance * np.linalg.norm(xo)


def test_exported_system_solved_independently(gpu, tmp_path):
    """North-star parity route: export the component matrices as .mtx (the reference's file set), rebuild the explicit
    operator A = -dt [G Dt]^T McInv [G Dt] - [JG JDt]^T BInv [JG JDt] - 1/2 diag(0,uInv) (AssembleSystem.cpp:381-389)
    with scipy — no oracle, no product code — solve A x = b directly and compare with the HIP PCG solution."""
    import scipy.io
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    sc, p = scenes.blob(20, 18, 22, seed=9, tile=8)
    p.tolerance = 1e-9
    p.maxSolverIterations = 20000
    rc = gpu.step(sc, p)
    assert rc == abi.SUCCESS
    pre = str(tmp_path) + "/sys."
    gpu.export_component_matrices(pre)
    rd = lambda n: scipy.io.mmread(pre + n + ".mtx")
    G, Dt, JG, JDt = (rd("Mat_" + n).tocsr() for n in ("G", "Dt", "JG", "JDt"))
    McInv, uInv, BInv = rd("Mat_McInv").tocsr(), rd("Mat_uInv").tocsr(), rd("Mat_Inv_Mr_plus_2JDtuDJ").tocsr()
    b = np.asarray(rd("Vec_b")).ravel()
    nP, nT = G.shape[1], Dt.shape[1]
    C = sp.hstack([G, Dt]).tocsr()
    J = sp.hstack([JG, JDt]).tocsr()
    U = sp.block_diag([sp.csr_matrix((nP, nP)), uInv]).tocsr()
    A = (-sc.dt * (C.T @ McInv @ C) - J.T @ BInv @ J - 0.5 * U).tocsc()
    assert abs(A - A.T).max() <= 1e-10 * abs(A).max()
    x = np.asarray(rd("solutionVector")).ravel()
    # DOFs with an empty row (e.g. a pressure cell whose centre liquid weight is 0, so every G coefficient is skipped,
    # ConstructMatrixBlocks.cpp:414) are decoupled: b is 0 there and CG leaves x at 0
    keep = np.diff(A.tocsr().indptr) > 0
    assert np.all(b[~keep] == 0) and np.all(x[~keep] == 0)
    Ak = A.tocsr()[keep][:, keep].tocsc()
    x_ref = np.zeros_like(b)
    x_ref[keep] = spla.spsolve(Ak, b[keep])
    assert np.linalg.norm(x - x_ref) <= 1e-5 * np.linalg.norm(x_ref)
    assert np.linalg.norm(A @ x - b) <= 1e-7 * np.linalg.norm(b)


def test_rigid_rotation_is_preserved_on_the_gpu(gpu):
    """Known answer, no oracle involved (tests/test_oracle_kat.py holds the same for the oracle): a liquid ball spinning rigidly about a tilted
    axis has zero strain rate — b = 0 and the velocity comes back unchanged, on the active faces and through the 26-DOF fit of the reduced
    tile (whose basis contains the rigid modes)."""
    from helpers import rigid_rotation_scene
    sc, p, ref = rigid_rotation_scene()
    assert gpu.step(sc, p) == abi.SUCCESS and gpu.nRegions >= 1
    assert np.abs(gpu.array("b")).max() < 1e-9 * (np.abs(gpu.array("activeRHSVector")).max() / sc.dx)
    vmax = max(np.abs(r).max() for r in ref)
    for a in range(3):
        ok = gpu.valid[a].ravel() > 0
        assert ok.sum() > 100
        assert np.abs(gpu.vel[a].ravel()[ok] - ref[a][ok]).max() <= 1e-6 * vmax


def test_interrupt_callback_stops_the_solve(gpu):
    sc, p = scenes.cavity(32)
    calls = []
    gpu.set_interrupt(lambda: calls.append(1) or len(calls) >= 2)
    try:
        rc = gpu.step(sc, p)
    finally:
        gpu.set_interrupt(None)
    assert rc == abi.INCOMPLETE and len(calls) == 2
    assert 25 <= gpu.stats.solveData[1] <= 75
    for a in range(3):
        assert np.array_equal(gpu.vel[a], sc.vel[a])       # velocity untouched
    assert gpu.step(sc, p) == abi.SUCCESS                   # and the context is reusable afterwards


@pytest.mark.parametrize("precond", [abi.PRE_IDENTITY, abi.PRE_DIAGONAL])
def test_exported_system_import_and_solve(gpu, tmp_path, precond):
    """ps_solve_exported_system: read back the .mtx component set (the reference's exportComponentMatrices file set,
    Solver.cpp:543-566) and run the same PCG on it with general CSR SpMVs.  Must agree with the in-memory solve: same
    iteration count (the operator is the same up to summation order) and the same solution."""
    import scipy.io
    sc, p = scenes.blob(20, 18, 22, seed=9, tile=8)
    p.tolerance = 1e-8
    p.maxSolverIterations = 20000
    p.preconditioner = precond
    assert gpu.step(sc, p) == abi.SUCCESS
    its = gpu.stats.solveData[1]
    pre = str(tmp_path) + "/sys."
    gpu.export_component_matrices(pre)
    x_mem = np.asarray(scipy.io.mmread(pre + "solutionVector.mtx")).ravel()
    n = x_mem.size
    assert n == int(gpu.stats.dimData[21])
    rc, x = gpu.solve_exported_system(pre, p, sc.dt, n)
    assert rc == abi.SUCCESS
    assert abs(gpu.stats.solveData[1] - its) <= 2
    assert int(gpu.stats.dimData[21]) == n and int(gpu.stats.dimData[24]) > 0
    assert np.linalg.norm(x - x_mem) <= 1e-6 * np.linalg.norm(x_mem)


def test_exported_system_import_errors(gpu, tmp_path):
    with pytest.raises(RuntimeError, match="cannot open"):
        gpu.solve_exported_system(str(tmp_path) + "/nothing.", abi.default_params(), 0.1, 4)


@pytest.mark.parametrize("env", [{"PS_COL32": "1"}, {"PS_FORCE_FP64_VALUES": "1"}, {"PS_FORCE_FP64_VALUES": "1", "PS_COL32": "1"},
                                 {"PS_PIPE_GRID": "0"}, {"PS_XCD": "0"}, {"PS_NO_DIAG_CODES": "1"}, {"PS_TILE_SPLIT": "1"},
                                 {"PS_FUSED_R": "1"},
                                 {"PS_FUSED_R": "1", "PS_TILE_SPLIT": "1"}, {"PS_FUSED_R": "1", "PS_NO_DIAG_CODES": "1"},
                                 {"PS_NT_LEVEL": "1"}, {"PS_NT_LEVEL": "2"}, {"PS_FUSED_R": "1", "PS_NT_LEVEL": "2"},
                                 {"PS_NO_SHARED_RUNS": "1"}, {"PS_CHUNK_PLAIN": "1"}, {"PS_WEAK_CHUNK_HASH": "1"}, {"PS_CHUNK_PLAIN": "1", "PS_FUSED_R": "1", "PS_NT_LEVEL": "2"},
                                 {"PS_NO_ELL": "1"}, {"PS_IL": "0"}, {"PS_IL": "3", "PS_NO_ELL": "1"}, {"PS_WG_RUN": "0"}, {"PS_WG_RUN": "2", "PS_FUSED_R": "1"},
                                 {"PS_NO_ELL": "1", "PS_FUSED_R": "1"}, {"PS_NO_SHARED_RUNS": "1", "PS_FUSED_R": "1"}])
def test_fallback_kernel_paths_agree(gpu, tmp_path, env):
    """The SpMV has four storage formats chosen at setup — compressed stream with int8 value codes (3 B/nnz) or with fp64
    values (10 B/nnz: values that are not code * scale), both on the pipelined kernels; int8-coded CSR and fp64 CSR on the
    one-shot kernels — plus switches read once per process (numbering lattice, chunk schedule, walk; the four-kernel PCG step
    and the cache-policy level, which default by system size: this grid runs five kernels at level 0).  Run the
    alternatives in a child process: same iteration count, same velocities to rounding (the formats reproduce the same
    fp64 products; only summation orders of the dot products differ)."""
    import os
    import subprocess
    import sys
    sc, p = scenes.blob(20, 18, 22, seed=9, tile=8)
    p.tolerance = 1e-8
    p.maxSolverIterations = 20000
    assert gpu.step(sc, p) == abi.SUCCESS
    assert int(gpu.array("fusedStep")[0]) == 0    # small system: five-kernel PCG step (the four-kernel one runs from 1.2 M rows, or PS_FUSED_R=1)
    out = str(tmp_path / "alt.npz")
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import polystokes_amd\nfrom polystokes_amd import scenes\n"
        "sc, p = scenes.blob(20, 18, 22, seed=9, tile=8)\np.tolerance = 1e-8\np.maxSolverIterations = 20000\n"
        "s = polystokes_amd.Solver(0)\nrc = s.step(sc, p)\n"
        f"np.savez({out!r}, rc=rc, it=s.stats.solveData[1], vx=s.vel[0], vy=s.vel[1], vz=s.vel[2], c16=s.array('columns16'), coded=s.array('valuesCoded'), dc=s.array('diagonalsCoded'), fused=s.array('fusedStep'), rpl=s.array('rowPerLane'))\n"
    )
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, **env), timeout=300)
    alt = np.load(out)
    assert int(alt["rc"]) == abi.SUCCESS
    assert int(alt["dc"][0]) == (0 if "PS_NO_DIAG_CODES" in env else 2)   # blob: variable viscosity -> uInv stays fp64, McInv is coded
    assert int(alt["fused"][0]) == (1 if "PS_FUSED_R" in env else 0)
    # row-per-lane kernels + kind-major numbering by default on coded values; the switches that take them away
    no_ell = any(k in env for k in ("PS_NO_ELL", "PS_COL32", "PS_FORCE_FP64_VALUES")) or env.get("PS_PIPE_GRID") == "0"
    assert int(alt["rpl"][0]) == (0 if no_ell else 3), env
    assert int(alt["rpl"][1]) == (int(env["PS_IL"]) if "PS_IL" in env else (0 if no_ell else 3)), env
    assert int(gpu.array("rowPerLane")[0]) == 3
    if "PS_COL32" in env:
        assert int(alt["c16"][0]) == 0
    if "PS_FORCE_FP64_VALUES" in env:
        assert int(alt["coded"][0]) == 0 and int(alt["c16"][0]) == (0 if "PS_COL32" in env else 3)
    assert abs(float(alt["it"]) - gpu.stats.solveData[1]) <= 1
    for a, k in enumerate(("vx", "vy", "vz")):
        scale = max(np.abs(gpu.vel[a]).max(), 1e-30)
        assert np.abs(alt[k] - gpu.vel[a]).max() <= 1e-6 * scale


@pytest.mark.parametrize("precond", [abi.PRE_IDENTITY, abi.PRE_DIAGONAL])
def test_repeated_solves_are_bit_identical(gpu, precond):
    """Every reduction runs in a fixed order, so a solve is reproducible bit for bit.  Small grids are where a race between
    the blocks of the fused step kernels would show (blocks finish before others start): 150 solves, one outcome."""
    sc, p = scenes.cavity(32, precond=precond)
    gpu.upload(sc, p)
    seen = set()
    for _ in range(150):
        rc = gpu.step_device()
        seen.add((rc, int(gpu.stats.solveData[1]), float(gpu.stats.solveData[0]).hex()))
    assert len(seen) == 1, seen
    vel0, _ = gpu.download()
    gpu.step_device()
    vel1, _ = gpu.download()
    for a in range(3):
        assert np.array_equal(vel0[a], vel1[a])


@pytest.mark.parametrize("precond,degree", [(abi.PRE_IDENTITY, 0), (abi.PRE_DIAGONAL, 0), (abi.PRE_CHEBYSHEV, 4), (abi.PRE_CHEBYSHEV, 1)])
def test_four_kernel_step_matches_and_is_reproducible(gpu, tmp_path, precond, degree):
    """The four-kernel PCG step (residual update inside the St kernel, p.Ap from the factored form; default from 1.2 M rows)
    forced on a small grid in a child process: same iteration count as the five-kernel step to +-1, same x to rounding,
    and 100 solves give one outcome bit for bit.  With the Chebyshev preconditioner the same St kernel also forms the
    polynomial's first term on the rows it updates (degree 1: that term is the whole preconditioner)."""
    import os
    import subprocess
    import sys
    sc, p = scenes.cavity(32, precond=precond)
    p.tolerance = 1e-8
    p.preconditionerDegree = degree
    assert gpu.step(sc, p) == abi.SUCCESS
    assert int(gpu.array("fusedStep")[0]) == 0
    x_ref = gpu.array("solutionVector").copy()
    out = str(tmp_path / "fused.npz")
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import polystokes_amd\nfrom polystokes_amd import scenes\n"
        f"sc, p = scenes.cavity(32, precond={int(precond)})\np.tolerance = 1e-8\np.preconditionerDegree = {int(degree)}\n"
        "s = polystokes_amd.Solver(0)\ns.upload(sc, p)\nseen = set()\n"
        "for _ in range(100):\n"
        "    rc = s.step_device()\n"
        "    seen.add((rc, int(s.stats.solveData[1]), float(s.stats.solveData[0]).hex(), s.array('solutionVector').tobytes()))\n"
        f"np.savez({out!r}, outcomes=len(seen), rc=rc, it=s.stats.solveData[1], x=s.array('solutionVector'), fused=s.array('fusedStep'))\n"
    )
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, PS_FUSED_R="1"), timeout=300)
    alt = np.load(out)
    assert int(alt["fused"][0]) == 1 and int(alt["rc"]) == abi.SUCCESS
    assert int(alt["outcomes"]) == 1
    assert abs(float(alt["it"]) - gpu.stats.solveData[1]) <= 1
    assert np.linalg.norm(alt["x"] - x_ref) <= 1e-6 * np.linalg.norm(x_ref)


@pytest.mark.skipif(os.environ.get("PS_TEST_CHILD") == "1", reason="this IS the child run")
def test_oracle_parity_with_the_large_system_switches_forced(tmp_path):
    """The four-kernel PCG step and the non-temporal cache policies switch on by system size (2 M / 4 M / 10 M rows), above
    what the oracle can check.  Run the oracle-parity tests of the solve (all fixed scenes: free surfaces, no tiles, ragged grids,
    layer variants; Jacobi; BiCGStab fallback; Chebyshev) once more in a child pytest with both forced on."""
    import subprocess
    import sys
    env = dict(os.environ, PS_FUSED_R="1", PS_NT_LEVEL="2", PS_TEST_CHILD="1")
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                         "-k", "solve_matches or jacobi_pcg or bicgstab or chebyshev or rhs_and_operator or zero_rhs"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert pr.returncode == 0, pr.stdout[-4000:]
    assert " passed" in pr.stdout and "failed" not in pr.stdout, pr.stdout[-2000:]


def test_shared_runs_on_a_periodic_scene(gpu, oracle_mod):
    """Chunks of the compressed stream start at lattice-block / tile boundaries, and chunks with byte-identical runs share one run
    (array `streamRuns` = distinct / all entries of S, then St).  A 64^3 cavity has 64 tiles in 27 neighbourhood classes: most of
    the stream must be shared, and the operator must still be the oracle's."""
    sc, p = scenes.cavity(64)
    o = oracle_mod.Oracle()
    o.run(sc, p, solve=False)
    gpu.upload(sc, p)
    gpu.setup()
    r = gpu.array("streamRuns")
    assert r[1] > 0 and r[3] > 0
    assert r[0] <= 0.6 * r[1] and r[2] <= 0.6 * r[3], r          # the interior classes repeat
    x = np.random.RandomState(5).standard_normal(gpu.nP + gpu.nT)
    yo, yg = o.apply(x), gpu.apply(x)
    assert np.abs(yo - yg).max() <= 1e-10 * np.abs(yo).max()


def test_bad_parameters_are_refused(gpu):
    """A raw ABI caller gets an error (ps_last_error) instead of a division by zero or a non-finite system."""
    sc, p = scenes.cavity(16)
    for field, value, msg in (("tileSize", 0, "tileSize"), ("tilePadding", -1, "negative"), ("tolerance", -1.0, "negative"),
                              ("maxSolverIterations", -5, "negative"), ("preconditioner", 3, "preconditioner")):
        q = abi.default_params()
        setattr(q, field, value)
        with pytest.raises(RuntimeError, match=msg):
            gpu.upload(sc, q)
    bad = abi.Scene(16, 16, 16, 0.0, sc.dt, 1.0, [0.0, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0)
    with pytest.raises(RuntimeError, match="dx and dt"):
        gpu.upload(bad, p)
    bad = abi.Scene(16, 16, 16, sc.dx, sc.dt, 0.0, [0.0, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0)
    with pytest.raises(RuntimeError, match="density"):
        gpu.upload(bad, p)
    assert gpu.step(sc, p) == abi.SUCCESS   # the context stays usable


@pytest.mark.parametrize("keep", [1, 0])
def test_do_solve_off_follows_the_reference(gpu, oracle_mod, keep):
    """HDK_PolyStokes.C:513,566-583: with doSolve off the result stays INCOMPLETE; the valid field is still built; with
    keepNonConvergedResults (the default) recovery and write-back run all the same — from the ZERO solution vector of assemble()
    (AssembleSystem.cpp:469): u = McInv rhs_a on active faces, the tile's smoothed fit on reduced ones; with it off the velocity
    field is left alone."""
    sc, p = scenes.spheres(32, tile=8)
    p.doSolve, p.keepNonConvergedResults = 0, keep
    o = oracle_mod.Oracle()
    ro = o.run(sc, p)
    rc = gpu.step(sc, p)
    assert rc == ro == abi.INCOMPLETE
    for a in range(3):
        assert np.array_equal(gpu.valid[a].ravel(), o.array("valid" + "XYZ"[a]))
        vo = o.array("vel" + "XYZ"[a])
        if keep:
            assert np.abs(gpu.vel[a].ravel() - vo).max() <= 1e-6 * max(np.abs(vo).max(), 1e-30)
            assert not np.array_equal(vo, np.asarray(sc.vel[a], np.float32).ravel())      # (the reference does change the field here)
        else:
            assert np.array_equal(gpu.vel[a].ravel(), np.asarray(sc.vel[a], np.float32).ravel())
            assert np.array_equal(vo, np.asarray(sc.vel[a], np.float32).ravel())


@pytest.mark.parametrize("keep", [1, 0])
def test_noconverge_with_and_without_keeping_the_results(gpu, oracle_mod, keep):
    """HDK_PolyStokes.C:566,590-605: a solve that runs out of iterations (PCG, then the BiCGStab retry) returns NOCONVERGE; its
    iterate is written back only with keepNonConvergedResults."""
    sc, p = scenes.blob(seed=6)
    p.maxSolverIterations, p.keepNonConvergedResults = 5, keep
    o = oracle_mod.Oracle()
    ro = o.run(sc, p)
    rc = gpu.step(sc, p)
    assert rc == ro == abi.NOCONVERGE
    assert o.stats.usedBiCGStab == 1 and gpu.stats.usedBiCGStab == 1
    for a in range(3):
        assert np.array_equal(gpu.valid[a].ravel(), o.array("valid" + "XYZ"[a]))
        vo = o.array("vel" + "XYZ"[a])
        if keep:      # five BiCGStab iterations from zero on both sides: the same iterate up to the summation order of the dots
            assert np.abs(gpu.vel[a].ravel() - vo).max() <= 1e-3 * max(np.abs(vo).max(), 1e-30)
        else:
            assert np.array_equal(gpu.vel[a].ravel(), np.asarray(sc.vel[a], np.float32).ravel())
            assert np.array_equal(vo, np.asarray(sc.vel[a], np.float32).ravel())


_TILE_CLASS_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import polystokes_amd
from polystokes_amd import scenes
from polystokes_amd._abi import Scene, default_params
n = 64
dx = 1.0 / n
z, y, x = np.meshgrid((np.arange(n) + 0.5) * dx, (np.arange(n) + 0.5) * dx, (np.arange(n) + 0.5) * dx, indexing="ij", sparse=True)
surface = (y - 0.64) + 0.0 * x + 0.0 * z                                   # a pool with a flat free surface (cuts the fifth tile layer)
visc = 50.0 * (1.0 + 0.5 * np.sin(2 * np.pi * x * (n / 16.0)) * np.cos(2 * np.pi * z * (n / 16.0))) + 0.0 * y   # varies inside a tile, repeats tile to tile
sc = Scene(n, n, n, dx, 1.0 / 24.0, 1000.0, [0.0, -1.0, 0.0], surface, np.float32(1.0), visc, name="pool_varvisc")
p = default_params(tileSize=16, tilePadding=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
np.savez(sys.argv[2], K=s.array("reducedViscosityMatrices"), Binv=s.array("Inv_Mr_plus_2JDtuDJ"), Mr=s.array("reducedMassMatrices"), R=np.int64(s.nRegions))
s.close()
'''


@pytest.mark.skipif(os.environ.get("PS_TEST_CHILD") == "1", reason="child run")
def test_shared_tile_blocks_equal_the_tiles_own_sums(tmp_path):
    """K is shared between tiles of one class (same labels, regions, viscosity samples; ps_tiles.hip:buildTileClasses) — on by
    default.  On a free-surface scene whose viscosity varies INSIDE every tile (and repeats from tile to tile, so that classes exist) the
    shared blocks must equal the blocks every tile sums for itself (PS_NO_TILE_CLASSES=1) to rounding: K and BInv <= 1e-12 relative per
    block; Mr is never shared: bit-identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, env in (("shared", {}), ("own", {"PS_NO_TILE_CLASSES": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        pr = subprocess.run([sys.executable, "-c", _TILE_CLASS_CHILD, root, f], env=dict(os.environ, PS_VERBOSE="1", **env), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True, timeout=600)
        assert pr.returncode == 0, pr.stdout[-3000:]
        out[tag] = (np.load(f), pr.stdout)
    a, b = out["shared"][0], out["own"][0]
    R = int(a["R"])
    assert R == int(b["R"]) and R >= 32
    import re
    m = re.search(r"tile classes: (\d+) of (\d+) tiles sum their own", out["shared"][1])
    assert m and int(m.group(1)) < int(m.group(2)) // 2, out["shared"][1][-500:]            # classes do exist on this scene
    assert np.array_equal(a["Mr"], b["Mr"])
    for nm in ("K", "Binv"):
        A, B = a[nm].reshape(R, -1), b[nm].reshape(R, -1)
        rel = np.abs(A - B).max(axis=1) / np.abs(B).max(axis=1)
        assert rel.max() <= 1e-12, (nm, rel.max())
