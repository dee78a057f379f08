"""BASELINE-size checks (256^3 cavity, tile 16 / pad 2) through size-independent properties: the oracle cannot run
this size in seconds, so the HIP path is checked against closed forms and against its own invariants."""
import numpy as np
import pytest

from polystokes_amd import _abi as abi
from polystokes_amd import scenes

pytestmark = pytest.mark.gpu

N = 256


@pytest.fixture(scope="module")
def big():
    import polystokes_amd
    sc, p = scenes.cavity(N, precond=abi.PRE_DIAGONAL)
    s = polystokes_amd.Solver(0)
    s.upload(sc, p)
    s.setup()
    yield sc, p, s
    s.close()


def test_closed_form_dof_counts(big):
    sc, p, s = big
    T, P, S = 16, 2, 2
    # per axis: tiles of 14 reduced cells, the last tile loses S layers to the domain-boundary solid layer (first tile's
    # boundary layer coincides with its padding)
    red_axis = (N // T) * (T - P) - S
    dd = s.stats.dimData
    assert dd[24] == (N // T) ** 3                                   # one region per tile
    assert dd[0] == N ** 3 - red_axis ** 3                           # active cells = pressures
    lab = s.array("centerLabels")
    assert (lab == abi.REDUCED).sum() == red_axis ** 3
    assert dd[12] == dd[0] and dd[13] == 3 * dd[0] + dd[4] + dd[5] + dd[6]
    # active + reduced faces tile the whole face grid (all-liquid box: no UNSOLVED / SOLID faces)
    fl = s.array("faceXLabels")
    assert (fl == abi.ACTIVEFLUID).sum() + (fl == abi.REDUCED).sum() == (N + 1) * N * N
    assert (fl == abi.ACTIVEFLUID).sum() == dd[1]
    # indices are permutations
    for nm, cnt in (("centerActiveIndices", dd[0]), ("faceZActiveIndices", dd[3]), ("edgeXZActiveIndices", dd[5])):
        idx = s.array(nm)
        v = idx[idx >= 0]
        assert len(v) == cnt and v.min() == 0 and v.max() == cnt - 1 and len(np.unique(v)) == cnt


def test_tile_blocks_closed_forms(big):
    sc, p, s = big
    Mr = s.array("reducedMassMatrices").reshape(-1, 26, 26)
    com = s.array("reducedRegionCOM").reshape(-1, 3)
    # first tile: full 14^3 block, Mr[a,a] = rho * s^2 (s+1); COM = mean integer coordinate * dx
    for a in range(3):
        assert Mr[0, a, a] == pytest.approx(14 * 14 * 15, rel=1e-14)
    np.testing.assert_allclose(com[0], 8.5 * sc.dx, rtol=1e-15)
    assert np.abs(Mr - np.transpose(Mr, (0, 2, 1))).max() <= 1e-9 * np.abs(Mr).max()
    Bi = s.array("Inv_Mr_plus_2JDtuDJ").reshape(-1, 26, 26)
    K = s.array("reducedViscosityMatrices").reshape(-1, 26, 26)
    for r in (0, 17, 4095):
        B = Mr[r] / sc.dt + 2 * K[r]
        assert np.abs(Bi[r] @ B - np.eye(26)).max() < 1e-7
        assert np.abs(K[r][:, :3]).max() < 1e-6 * np.abs(K[r]).max()      # rigid translations carry no viscous stress


def test_operator_symmetric_negative_definite(big):
    sc, p, s = big
    n = s.nP + s.nT
    rng = np.random.RandomState(3)
    x, y = rng.randn(n), rng.randn(n)
    Ax, Ay = s.apply(x), s.apply(y)
    assert abs(x @ Ay - y @ Ax) <= 1e-9 * abs(x @ Ay)
    assert x @ Ax < 0 and y @ Ay < 0
    # linearity
    z = s.apply(2.0 * x - 3.0 * y)
    assert np.abs(z - (2.0 * Ax - 3.0 * Ay)).max() <= 1e-9 * np.abs(Ax).max()


def test_solution_satisfies_reference_stop_rule(big):
    sc, p, s = big
    rc = s.solve()
    assert rc == abi.SUCCESS
    x, b = s.array("solutionVector"), s.array("b")
    r = b - s.apply(x)
    rre = min(r @ r, (r @ r) / (x @ x))                    # pcg.h:319-325
    assert rre < p.tolerance ** 2 * 1.001
    assert s.stats.solveData[0] == pytest.approx(np.sqrt(rre), rel=1e-5)
    assert 0 < s.stats.solveData[1] < p.maxSolverIterations
    vel, valid = s.download()
    assert all(np.all(v == 1.0) for v in valid)
    assert all(np.isfinite(v).all() for v in vel)
    # the one-cell lid impulse is smeared out by the viscous solve but still drives the top more than the bottom
    assert vel[0][N - 1].mean() > 10 * abs(vel[0][0]).mean() and abs(vel[0]).max() <= 1.0


def test_large_system_paths_and_chebyshev_at_full_size(big):
    """What switches on by size ran in the solve above: the four-kernel PCG step, and a compressed stream whose chunks share their
    runs (256^3 cavity: 4096 tiles in 125 neighbourhood classes -> under 2 % distinct entries).  Then the same system with the
    Chebyshev preconditioner: the reference's stop rule on the true residual, and at least 3.5x fewer iterations than Jacobi."""
    sc, p, s = big
    assert s.solve() == abi.SUCCESS
    it_jacobi = int(s.stats.solveData[1])
    assert int(s.array("fusedStep")[0]) == 1
    r = s.array("streamRuns")
    assert 0 < r[0] <= 0.02 * r[1] and 0 < r[2] <= 0.02 * r[3], r
    p2 = type(p).from_buffer_copy(p)
    p2.preconditioner = abi.PRE_CHEBYSHEV
    s.upload(sc, p2)
    s.setup()
    assert s.solve() == abi.SUCCESS
    x, b = s.array("solutionVector"), s.array("b")
    res = b - s.apply(x)
    rre = min(res @ res, (res @ res) / (x @ x))
    assert rre < p.tolerance ** 2 * 1.001
    assert 0 < s.stats.solveData[1] * 3.5 <= it_jacobi, (s.stats.solveData[1], it_jacobi)
    it_cheb64, x64 = int(s.stats.solveData[1]), x
    assert int(s.array("chebInner32")[0]) == 0
    # ... and with the polynomial's inner vectors stored as fp32 (PS_PRE_CHEBYSHEV_F32, r06): the kernels for it run at this size, the stop
    # rule holds on the TRUE fp64 residual, the count grows by at most 5 %, x agrees with the fp64 polynomial's within 10 tol
    p3 = type(p).from_buffer_copy(p)
    p3.preconditioner = abi.PRE_CHEBYSHEV_F32
    s.upload(sc, p3)
    s.setup()
    assert s.solve() == abi.SUCCESS
    assert int(s.array("chebInner32")[0]) == 1 and int(s.array("fusedStep")[0]) == 1
    x32 = s.array("solutionVector")
    res = b - s.apply(x32)
    assert min(res @ res, (res @ res) / (x32 @ x32)) < p.tolerance ** 2 * 1.001
    assert it_cheb64 - 2 <= s.stats.solveData[1] <= 1.05 * it_cheb64 + 2, (s.stats.solveData[1], it_cheb64)
    assert np.linalg.norm(x32 - x64) <= 10 * p.tolerance * np.linalg.norm(x64)
    s.upload(sc, p)      # leave the module's context as the other tests expect it
    s.setup()


def test_config2_coil_128_properties():
    """BASELINE config 2 stand-in at its real size (128^3 coil, tile 16 / pad 2): free surface, solid floor, air."""
    import polystokes_amd
    sc, p = scenes.coil(128)
    s = polystokes_amd.Solver(0)
    s.upload(sc, p)
    s.setup()
    n = s.nP + s.nT
    dd = s.stats.dimData
    assert dd[24] > 50 and dd[0] > 1e5                      # many tiles, many active cells
    lab = s.array("centerLabels")
    assert set(np.unique(lab)) <= {abi.UNSOLVED, abi.ACTIVEFLUID, abi.SOLID, abi.REDUCED}
    for nm, cnt in (("centerActiveIndices", dd[0]), ("faceYActiveIndices", dd[2]), ("edgeXYActiveIndices", dd[6])):
        idx = s.array(nm)
        v = idx[idx >= 0]
        assert len(v) == cnt and len(np.unique(v)) == cnt
    rng = np.random.RandomState(4)
    x, y = rng.randn(n), rng.randn(n)
    Ax, Ay = s.apply(x), s.apply(y)
    assert abs(x @ Ay - y @ Ax) <= 1e-9 * abs(x @ Ay) and x @ Ax < 0
    rc = s.solve()
    assert rc == abi.SUCCESS
    xs, b = s.array("solutionVector"), s.array("b")
    r = b - s.apply(xs)
    assert min(r @ r, (r @ r) / (xs @ xs)) < p.tolerance ** 2 * 1.001
    vel, valid = s.download()
    assert all(np.isfinite(v).all() for v in vel)
    # air faces are invalid, liquid faces valid
    assert 0 < valid[1].mean() < 1
    s.close()


def _stop_rule_holds(s, p):
    xs, b = s.array("solutionVector"), s.array("b")
    r = b - s.apply(xs)
    return min(r @ r, (r @ r) / (xs @ xs)) < p.tolerance ** 2 * 1.001


@pytest.mark.parametrize("scene,n,worlds", [("coil", 512, (2, 4, (2, 2, 2))), ("spheres", 256, (4, (2, 2, 2)))])
def test_config4_and_5_full_size_single_and_slabs(scene, n, worlds):
    """BASELINE config 4 (coiling column 512^3, 78 M DOFs) and config 5 stand-in (8 moving solid spheres, 256^3, mu = 1e4) at
    their stated sizes: the single-domain step satisfies the reference's stop rule on the operator it solved; the same scene cut
    into 2 / 4 z-slabs — and, config 5, into 2 x 2 x 2 bricks: its 8-rank decomposition — (the distributed algorithm,
    in-process ranks on this one GPU) takes the same number of iterations, marks the
    same faces valid and — where the scene is not of the AMP kind — returns the same velocities.
    The 8-rank run of these configs needs an 8-GPU node (bench.py --gpus 8 --scaling strong --scene coil --res 512)."""
    import polystokes_amd
    sc, p = getattr(scenes, scene)(n)
    p.preconditioner = abi.PRE_DIAGONAL
    s = polystokes_amd.Solver(0)
    rc = s.step(sc, p)
    assert rc == abi.SUCCESS
    it1 = int(s.stats.solveData[1])
    assert s.nP + s.nT > (5e7 if scene == "coil" else 2e7) and s.nRegions > 1000
    assert _stop_rule_holds(s, p)
    vel1, valid1 = s.vel, s.valid
    # the solution vector as seven grid fields (p, txx, tyy, tzz, the three edge stresses): the form in which a slab group's solution
    # can be compared with it, whatever the ranks' numberings
    from helpers import DOF_KINDS, dof_field, merge_dof_field, merge_dof_field_brick
    x1 = s.array("solutionVector")
    x1_norm = float(np.linalg.norm(x1))
    fields1 = {k: dof_field(s, k, x1) for k in DOF_KINDS}
    del x1
    s.close()
    for world in worlds:
        dims = world if isinstance(world, tuple) else None
        grp = polystokes_amd.Group(world, dims=dims) if dims is None else polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
        rc2 = grp.solve_scene(sc, p)
        assert rc2 == abi.SUCCESS
        it2 = int(grp.stats.solveData[1])
        assert abs(it1 - it2) <= max(2, 0.02 * it1), (scene, world, it1, it2)
        # x of the group (every rank's owned DOFs) against the single domain's: the tolerance the north star states, on x
        d2 = 0.0
        for kind in DOF_KINDS:
            merged = np.full(fields1[kind].shape, np.nan, np.float32)
            for r, sl in enumerate(grp.slabs):
                xr = grp.ranks[r].array("solutionVector")
                if dims is None:
                    merge_dof_field(merged, dof_field(grp.ranks[r], kind, xr), sl, kind)
                else:
                    merge_dof_field_brick(merged, dof_field(grp.ranks[r], kind, xr), sl, kind, (sc.nx, sc.ny, sc.nz))
                del xr
            assert np.array_equal(np.isnan(merged), np.isnan(fields1[kind])), (scene, world, kind)   # the same DOFs exist
            d = np.nan_to_num(merged.astype(np.float64) - fields1[kind])
            d2 += float((d * d).sum())
            del merged, d
        assert np.sqrt(d2) <= 10 * p.tolerance * x1_norm, (scene, world, np.sqrt(d2) / x1_norm)
        for a in range(3):
            assert np.array_equal(grp.valid[a], valid1[a]), (scene, world, a)
            # Velocities are NOT compared here: u = dt McInv (rhs/dt - [G Dt] x) differences 1e5-sized terms (coil: mu = 100,
            # rho = 1000; spheres: mu = 1e4), so two solves that both satisfy the reference's stop rule at tol 1e-3 can differ by
            # tens of per cent in u at this size and only agree as the tolerance goes to 1e-7 (scripts/amp_check.py, DESIGN.md
            # section 4, AMP).  The tight comparisons are on x (goldens, oracle parity) and on the small multirank scenes.
            assert np.isfinite(grp.vel[a]).all()
        grp.close()


def test_velocities_of_decompositions_converge_at_full_size():
    """The AMP statement as a test at BASELINE config 5's size: spheres 256^3 (mu = 1e4) solved by the single domain,
    by 4 z-slabs and by 2 x 2 x 2 bricks.  At the node's default tolerance 1e-3 their velocities differ by tens of per cent (every one of
    them satisfies the reference's stop rule); the difference goes away with the tolerance, and at any tolerance it is as large between
    two ROUNDING PATHS of one decomposition as between decompositions — measured with scripts/amp_dual.py over the four combinations of
    the one- / two-unit S and St kernels (same rows, same products, other grouping of the partial sums), single domain vs 4 slabs:
    3.1e-3 ... 9.0e-3 of the largest velocity at tol 1e-7 (12.6 k iterations), 8.3e-4 ... 1.7e-3 at 1e-8 (16.4 k).  Asserted at 1e-8:
    <= 5e-3, iterations within 5 %."""
    import polystokes_amd
    sc, p = scenes.spheres(256)
    p.preconditioner, p.tolerance, p.maxSolverIterations = abi.PRE_DIAGONAL, 1e-8, 100000
    s = polystokes_amd.Solver(0)
    assert s.step(sc, p) == abi.SUCCESS
    it1 = int(s.stats.solveData[1])
    v1 = [v.copy() for v in s.vel]
    s.close()
    for world, dims in ((4, None), (8, (2, 2, 2))):
        grp = polystokes_amd.Group(world, dims=dims)
        assert grp.solve_scene(sc, p) == abi.SUCCESS
        it2 = int(grp.stats.solveData[1])
        assert abs(it1 - it2) <= 0.05 * it1, (world, it1, it2)
        for a in range(3):
            d = float(np.abs(grp.vel[a] - v1[a]).max() / max(np.abs(v1[a]).max(), 1e-30))
            assert d <= 5e-3, (world, dims, a, d)
        grp.close()
