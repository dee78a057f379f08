"""Distributed (slab) solve on ONE GPU: `world` ranks in one process exchanging halos by device copies.
Checks the distributed algorithm (ownership, exchange lists, reduction order) against the single-domain solve."""
import numpy as np
import pytest

from polystokes_amd import _abi as abi
from polystokes_amd import scenes

pytestmark = pytest.mark.gpu


def _tall_cavity(nx, nz, precond=abi.PRE_IDENTITY, tile=16):
    sc0, p = scenes.cavity(nx, tile=tile, precond=precond)
    velx = np.zeros((nz, nx, nx + 1), np.float32)
    velx[nz - 1] = 1.0
    velx[nz // 2, :, : nx // 2] = -0.5          # something to do near the cut as well
    sc = abi.Scene(nx, nx, nz, sc0.dx, sc0.dt, 1.0, [velx, 0.0, 0.0], -1.0, 1.0, 1.0, name=f"tall{nx}x{nz}")
    return sc, p


def _tall_coil(n, nz):
    # liquid column + pool with a free surface crossing the cuts, solid floor
    sc0, p = scenes.coil(n)
    rep = lambda a, extra=0: np.concatenate([a] * (nz // n) + ([a[-1:]] if extra and False else []), axis=0)
    z, y, x = np.meshgrid((np.arange(nz) + 0.5) * sc0.dx, (np.arange(n) + 0.5) * sc0.dx, (np.arange(n) + 0.5) * sc0.dx, indexing="ij")
    r = 0.2
    col = np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2) - r          # column along z
    surface = np.minimum(col, z - 0.3)                           # pool below z = 0.3
    collision = z - 2 * sc0.dx
    sc = abi.Scene(n, n, nz, sc0.dx, sc0.dt, 1000.0, [0.0, 0.0, -1.0], surface, collision, 100.0, name=f"tallcoil{n}x{nz}")
    return sc, p


@pytest.mark.parametrize("case", ["cavity_w2", "cavity_w3_jacobi", "coil_w2", "cavity_w4_tile8", "spheres_w2", "spheres_w4_jacobi",
                                  "cavity_w3_chebyshev", "coil_w2_chebyshev", "spheres_w4_chebyshev_k2"])
def test_group_matches_single_domain(case):
    import polystokes_amd
    if case.startswith("spheres"):
        # BASELINE config 5 stand-in (moving solid spheres, mu = 1e4, mixed uniform / reduced regions) cut into 2 and 4 slabs
        sc, p = scenes.spheres(64)
        world = 2 if case == "spheres_w2" else 4
        if world == 4:
            p.preconditioner = abi.PRE_DIAGONAL
        if case.endswith("chebyshev_k2"):
            p.preconditioner, p.preconditionerDegree = abi.PRE_CHEBYSHEV, 2
    elif case == "cavity_w2":
        (sc, p), world = _tall_cavity(32, 64), 2
    elif case == "cavity_w3_jacobi":
        (sc, p), world = _tall_cavity(24, 96, precond=abi.PRE_DIAGONAL), 3
    elif case == "cavity_w3_chebyshev":        # the polynomial preconditioner across cuts: distributed interval estimate and applies
        (sc, p), world = _tall_cavity(24, 96, precond=abi.PRE_CHEBYSHEV), 3
    elif case == "coil_w2_chebyshev":
        (sc, p), world = _tall_coil(32, 64), 2
        p.preconditioner = abi.PRE_CHEBYSHEV
    elif case == "coil_w2":
        (sc, p), world = _tall_coil(32, 64), 2
    else:
        (sc, p), world = _tall_cavity(24, 64, tile=8), 4
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(world)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS
    it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
    assert abs(it1 - it2) <= max(2, 0.02 * it1), (it1, it2)
    # every owned label equals the global classification
    lab = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx)
    for r, sl in enumerate(grp.slabs):
        ll = grp.ranks[r].array("centerLabels").reshape(sl.nz_local, sc.ny, sc.nx)
        assert np.array_equal(ll[sl.zLoOwned:sl.zHiOwned], lab[sl.z0:sl.z1]), (case, r)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a]), case
        scale = max(np.abs(single.vel[a]).max(), 1e-30)
        assert np.abs(grp.vel[a] - single.vel[a]).max() <= 20 * p.tolerance * scale, case
    grp.close()
    single.close()


@pytest.mark.parametrize("case", ["cavity128_2x2x2_jacobi", "cavity64_2x2x2", "cavity64_2x1x2_jacobi", "cavity_1x2x1", "spheres64_2x2x2_jacobi", "coil64_2x2x1", "cavity48_tile8_3x2x2_chebyshev"])
def test_bricks_match_single_domain(case):
    """The decomposition along all three axes (bricks; ps_set_brick): in-process ranks on one GPU against the single domain —
    labels of every owned cell, valid faces, iteration count, velocities; a tile's matrices do not depend on the decomposition."""
    import polystokes_amd
    if case == "cavity128_2x2x2_jacobi":      # a group of 8 at 128^3
        (sc, p), dims = scenes.cavity(128, tile=16, precond=abi.PRE_DIAGONAL), (2, 2, 2)
    elif case == "cavity64_2x2x2":
        (sc, p), dims = scenes.cavity(64, tile=16), (2, 2, 2)
    elif case == "cavity64_2x1x2_jacobi":
        (sc, p), dims = scenes.cavity(64, tile=16, precond=abi.PRE_DIAGONAL), (2, 1, 2)
    elif case == "cavity_1x2x1":
        (sc, p), dims = _tall_cavity(32, 32), (1, 2, 1)
    elif case == "spheres64_2x2x2_jacobi":
        (sc, p), dims = scenes.spheres(64), (2, 2, 2)
        p.preconditioner = abi.PRE_DIAGONAL
    elif case == "coil64_2x2x1":
        (sc, p), dims = scenes.coil(64), (2, 2, 1)
    else:
        (sc, p), dims = scenes.cavity(48, tile=8, precond=abi.PRE_CHEBYSHEV), (3, 2, 2)
    world = dims[0] * dims[1] * dims[2]
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(world, dims=dims)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS
    it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
    assert abs(it1 - it2) <= max(2, 0.02 * it1), (it1, it2)
    lab = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx)
    for r, b in enumerate(grp.bricks):
        ll = grp.ranks[r].array("centerLabels").reshape(b.n_local[2], b.n_local[1], b.n_local[0])
        own = ll[b.lo[2]:b.hi[2], b.lo[1]:b.hi[1], b.lo[0]:b.hi[0]]
        assert np.array_equal(own, lab[b.g0[2]:b.g1[2], b.g0[1]:b.g1[1], b.g0[0]:b.g1[0]]), (case, r)
    _views_hold_the_global_cell_labels(single, grp, sc)      # the halo blocks too (their labels come from the owners)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a]), case
        scale = max(np.abs(single.vel[a]).max(), 1e-30)
        assert np.abs(grp.vel[a] - single.vel[a]).max() <= 20 * p.tolerance * scale, case
    grp.close()
    single.close()


@pytest.mark.skipif(__import__("os").environ.get("PS_TEST_CHILD") == "1", reason="this IS the child run")
@pytest.mark.parametrize("split", ["", "1"])
def test_bricks_and_slabs_with_the_four_kernel_step_forced(split):
    """The four-kernel PCG step across the cuts (the St kernel corrects r on owned rows, the halo rows' A p travels back axis after axis,
    k_relay2 / k_dist_fixup) switches on from 1.2 M owned rows per rank: force it (PS_FUSED_R=1) through the brick and slab cases in a child.
    r06: the ranks of an in-process group share one stream, so their S and St run as ONE launch each (two units per wave, halo rows included:
    k_spmv_St_ell2<.., HALO>) and an exchange is one kernel for all ranks (k_xchg_direct) — split "": that path; split "1" (PS_DIST_OVERLAP=1):
    the launches split into the chunks next to a cut and the rest, pack / transport / unpack, as one process per GPU runs them."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PS_FUSED_R="1", PS_TEST_CHILD="1")
    if split:
        env["PS_DIST_OVERLAP"] = split
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                         "-k", "test_bricks_match_single_domain or test_group_matches_single_domain"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert pr.returncode == 0, pr.stdout[-3000:]


@pytest.mark.skipif(__import__("os").environ.get("PS_TEST_CHILD") == "1", reason="this IS the child run")
@pytest.mark.parametrize("fused", ["0", "1"])
def test_bricks_with_the_forwarding_rounds_forced(fused):
    """r05: the exchanges of the solve run as ONE round (every list holds the sender's own samples; Dist::decideExchangeMode checks on the
    matrices that no row reaches a diagonal neighbour's sample) and fall back to the three forwarding rounds x -> y -> z of r03 / r04 only
    when one does (tilePadding 1: the seeds of test_bricks_classification_that_reaches_beyond_the_halo).  The brick cases once more in a child
    with the forwarding rounds forced (PS_DIST_FORWARD=1), with the five- and the four-kernel step: both modes must reproduce the single domain."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PS_DIST_FORWARD="1", PS_FUSED_R=fused, PS_TEST_CHILD="1")
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                         "-k", "test_bricks_match_single_domain or test_bricks_without_reduced_regions"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert pr.returncode == 0, pr.stdout[-3000:]


def test_rccl_entry_points_world1():
    """dlopen'ed RCCL on the solver stream: communicator init, all-reduce, grouped send/recv (to self)."""
    import polystokes_amd
    s = polystokes_amd.Solver(0)
    s.comm_init(polystokes_amd.comm_unique_id(), 0, 1)
    s.comm_selftest()
    s.close()


def test_bench_weak_scaling_pieces_match_the_global_cavity():
    """The rank-local scenes bench.py --gpus N feeds (scenes.cavity_slab), run as a 2-rank group, reproduce the
    single-domain solve of the global n x n x 2n cavity."""
    import polystokes_amd
    from polystokes_amd import partition
    n, world = 32, 2
    sc0, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    velx = np.zeros((n * world, n, n + 1), np.float32)
    velx[n * world - 1] = 1.0
    glob = abi.Scene(n, n, n * world, sc0.dx, sc0.dt, 1.0, [velx, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0)
    single = polystokes_amd.Solver(0)
    assert single.step(glob, p) == abi.SUCCESS
    grp = polystokes_amd.Group(world)
    slabs = []
    for r in range(world):
        sc, pr, sl = scenes.cavity_slab(n, world, r, precond=abi.PRE_DIAGONAL)
        grp.ranks[r].upload(sc, pr)
        grp.ranks[r].set_slab(sl)
        slabs.append(sl)
    assert grp.step() == abi.SUCCESS
    it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
    assert abs(it1 - it2) <= max(2, 0.02 * it1), (it1, it2)
    for r, sl in enumerate(slabs):
        lv, lval = grp.ranks[r].download()
        for a in range(3):
            own = grp.ranks[r].array("owned" + "XYZ"[a]).reshape(lv[a].shape) > 0
            ref = single.vel[a][sl.g0:sl.g0 + lv[a].shape[0]]
            scale = max(np.abs(single.vel[a]).max(), 1e-30)
            assert np.abs(lv[a][own] - ref[own]).max() <= 20 * p.tolerance * scale
    grp.close()
    single.close()


def test_group_bicgstab_fallback_matches_single_domain():
    """A slab solve that exhausts maxSolverIterations falls back to the distributed BiCGStab (pcg.h:134-200) like the
    single-domain solve: same verdict, same BiCGStab iteration index (+-1), same velocities to solver tolerance."""
    import polystokes_amd
    sc, p = _tall_cavity(24, 64)
    p.maxSolverIterations = 12
    p.tolerance = 5e-2
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    assert single.stats.usedBiCGStab == 1
    grp = polystokes_amd.Group(2)
    rc2 = grp.solve_scene(sc, p)
    assert grp.stats.usedBiCGStab == 1
    assert rc1 == rc2
    assert abs(single.stats.solveData[1] - grp.stats.solveData[1]) <= 1
    if rc1 == abi.SUCCESS:
        for a in range(3):
            scale = max(np.abs(single.vel[a]).max(), 1e-30)
            assert np.abs(grp.vel[a] - single.vel[a]).max() <= 20 * p.tolerance * scale
    grp.close()
    single.close()


def _compare_with_single(single, grp, sc, p, tag):
    rc1 = single.step(sc, p)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS, tag
    it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
    assert abs(it1 - it2) <= max(2, 0.02 * it1), (tag, it1, it2)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a]), tag
        scale = max(np.abs(single.vel[a]).max(), 1e-30)
        assert np.isfinite(grp.vel[a]).all(), tag
        assert np.abs(grp.vel[a] - single.vel[a]).max() <= 20 * p.tolerance * scale, tag


@pytest.mark.skipif(__import__("os").environ.get("PS_TEST_REUSE_CHILD") != "1", reason="runs in the child of the test below")
@pytest.mark.parametrize("dims", [None, (2, 2, 1), (2, 1, 2)])
def test_reused_group_child(dims):
    """One Group (and one single-domain context) stepped through scenes with DIFFERENT free surfaces — so a different DOF numbering,
    different halo rows and chunk lists in buffers the previous step sized and filled (halo rows of A p that no launch of
    the overlapped four-kernel step writes were packed for the neighbours all the same)."""
    import polystokes_amd
    world = 2 if dims is None else dims[0] * dims[1] * dims[2]
    single = polystokes_amd.Solver(0)
    grp = polystokes_amd.Group(world, dims=dims)
    seq = [scenes.spheres(64), scenes.coil(64), _tall_coil(32, 64) if dims is None else scenes.cavity(64, tile=16), scenes.spheres(64)]
    for i, (sc, p) in enumerate(seq):
        p.preconditioner = abi.PRE_DIAGONAL if i % 2 == 0 else abi.PRE_IDENTITY
        _compare_with_single(single, grp, sc, p, (dims, i, sc.name))
    grp.close()
    single.close()


@pytest.mark.skipif(__import__("os").environ.get("PS_TEST_CHILD") == "1", reason="this IS the child run")
@pytest.mark.parametrize("poison", ["0", "1"])
def test_group_reused_across_scenes_with_the_four_kernel_step(poison):
    """ADVICE r03 (high / medium / low): contexts reused across uploads with another numbering, the overlapped four-kernel step forced;
    with PS_DEBUG_POISON=1 every buffer alloc() hands out is filled with NaN / -1 first, so a word nobody wrote this step shows."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PS_FUSED_R="1", PS_TEST_CHILD="1", PS_TEST_REUSE_CHILD="1", PS_DEBUG_POISON=poison)
    if poison == "1":
        env["PS_DIST_OVERLAP"] = "1"      # (r06: an in-process group runs unsplit launches by default; the poisoned run walks the split ones)
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                         "-k", "test_reused_group_child"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert pr.returncode == 0, pr.stdout[-3000:]


@pytest.mark.parametrize("mode", ["dosolve_off_keep", "dosolve_off_drop", "noconverge_keep", "noconverge_drop"])
def test_group_parameter_paths_match_single_domain(mode):
    """doSolve = 0 and keepNonConvergedResults = 0 across a cut (HDK_PolyStokes.C:513,566,590-605; ps_dist.hpp:distStep): result code,
    valid faces and the velocity field of a 2-rank group against the single domain."""
    import polystokes_amd
    sc, p = scenes.spheres(32, tile=8)
    if mode.startswith("dosolve_off"):
        p.doSolve = 0
    else:
        p.maxSolverIterations = 4
    p.keepNonConvergedResults = 1 if mode.endswith("keep") else 0
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(2)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == (abi.INCOMPLETE if p.doSolve == 0 else abi.NOCONVERGE)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a])
        if mode.endswith("drop"):
            assert np.array_equal(grp.vel[a], np.asarray(sc.vel[a], np.float32).reshape(grp.vel[a].shape))
            assert np.array_equal(single.vel[a], grp.vel[a])
        else:
            scale = max(np.abs(single.vel[a]).max(), 1e-30)
            assert np.abs(grp.vel[a] - single.vel[a]).max() <= (1e-6 if p.doSolve == 0 else 1e-2) * scale
    grp.close()
    single.close()


@pytest.mark.parametrize("seed", [500, 504, 507, 508, 520])
def test_fuzz_brick_mismatches_vanish_with_the_tolerance(seed):
    """The five cases of `scripts/fuzz_bricks.py 40 500` whose velocities differ by 0.2 - 1.3 % between the single domain and the bricks at
    tol 1e-6: ill-conditioned solves of 800 - 4100 iterations where the reference's stop rule on the
    pressure-stress system leaves the velocities that loose (AMP).  Replayed at tol 1e-9 the two solves agree —
    the decomposition is not what separates them."""
    import polystokes_amd
    from helpers import fuzz_brick_case
    sc, p, dims, n, tile = fuzz_brick_case(seed, 1e-9)
    p.maxSolverIterations = 60000
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS, (seed, rc1, rc2)
    it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
    assert abs(it1 - it2) <= max(3, 0.05 * it1), (seed, it1, it2)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a])
        scale = max(np.abs(single.vel[a]).max(), 1e-30)
        assert np.abs(grp.vel[a] - single.vel[a]).max() <= 1e-4 * scale, (seed, a, np.abs(grp.vel[a] - single.vel[a]).max() / scale)
    grp.close()
    single.close()


@pytest.mark.parametrize("seed", [7026, 9042, 9003, 9004, 9029, 7058, 4237, 11404, 11417])
def test_fuzz_seeds_keep_the_iteration_count_at_the_reference_tolerance(seed):
    """the fuzz seeds whose brick solves differ from the single domain's by up to 13 % in ITERATIONS at tol 1e-6 (7026:
    1060 vs 1201; 4237: 4784 vs 4090; `profiles/r04_fuzz_summary.txt`) — replayed at the reference's own tolerance, 1e-3, where SURVEY 8c's
    +-2 % is claimed.  What holds there (`profiles/r05_seeds_tol1e-3.txt`): the counts are EQUAL on every one of them (121 / 11 / 117 / 122 /
    28 / 40 / 183) — and on r05's own sweep one case is not: seed 11417 stops after 26 iterations as a single domain and after 29 as 2 x 2 x 3
    bricks (`profiles/r05_fuzz_summary.txt`; 11404: 18 / 18).  Asserted: +-2 % or +-3 iterations.  The spread at 1e-6 is the stop rule hovering at
    its threshold over hundreds of iterations of an ill-conditioned system; at 1e-3 it is at most a few iterations."""
    import polystokes_amd
    from helpers import fuzz_brick_case
    sc, p, dims, n, tile = fuzz_brick_case(seed, 1e-3)
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS, (seed, rc1, rc2)
    it1, it2 = int(single.stats.solveData[1]), int(grp.stats.solveData[1])
    assert abs(it1 - it2) <= max(3, 0.02 * it1), (seed, it1, it2)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a])
    grp.close()
    single.close()


def _views_hold_the_global_cell_labels(single, grp, sc):
    """every rank's WHOLE view (owned box + halo blocks) carries the single domain's cell labels after the owners' exchange"""
    G = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx)
    for r, b in enumerate(grp.bricks):
        nx, ny, nz = b.n_local
        ox, oy, oz = b.origin
        loc = grp.ranks[r].array("centerLabels").reshape(nz, ny, nx)
        assert np.array_equal(loc, G[oz:oz + nz, oy:oy + ny, ox:ox + nx]), r


@pytest.mark.parametrize("seed", [4219, 4202, 4215])
def test_bricks_classification_that_reaches_beyond_the_halo(seed):
    """Found by `scripts/fuzz_bricks.py 40 4200` (seed 4219; refused with "exchange lists disagree" until r04): with tilePadding = 1 the
    reference's fixReducedRegionBoundaries (Classifier.cpp:1073-1172) demotes the REDUCED cells on both sides of a one-cell padding
    layer, so the classification of the tile next to a cut depends on the cell just OUTSIDE the halo block — and, through
    fixSmallReducedRegions (:1174-1262), a whole thin region next to the cut can exist in a rank's own view and not in the global one
    (`scripts/brick_diag.py 4219`).  The ranks now take the cell labels of their halo blocks from the owners (Dist::exchangeLabels,
    before the components and after the boundary fix): every view holds the global labels, and the bricks solve the single domain's
    system."""
    import polystokes_amd
    from helpers import fuzz_brick_case
    sc, p, dims, n, tile = fuzz_brick_case(seed, 1e-9)
    p.maxSolverIterations = 60000
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS, (seed, rc1, rc2)
    _views_hold_the_global_cell_labels(single, grp, sc)
    if seed == 4219:
        assert p.tilePadding == 1
        assert sum(grp.ranks[r].dist_stats()["halo_label_changes"] for r in range(grp.world)) > 0
    it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
    assert abs(it1 - it2) <= max(3, 0.05 * it1), (seed, it1, it2)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a])
        scale = max(np.abs(single.vel[a]).max(), 1e-30)
        assert np.abs(grp.vel[a] - single.vel[a]).max() <= 1e-4 * scale, (seed, a, np.abs(grp.vel[a] - single.vel[a]).max() / scale)
    grp.close()
    single.close()


@pytest.mark.parametrize("dims", [(2, 1, 2), (1, 1, 3)])
def test_bricks_without_reduced_regions_match_single_domain(dims):
    """doReducedRegions off (the uniform Stokes solve, BASELINE config 1's mode) under a decomposition: only the first of the two label
    exchanges runs (there are no regions to fix); iterations equal, velocities to 1e-6 at tol 1e-8, every view holds the global labels."""
    import polystokes_amd
    sc, p = scenes.blob(48, 32, 48, seed=5, tile=16, pad=2)
    p.doReducedRegions = 0
    p.preconditioner = abi.PRE_DIAGONAL
    p.tolerance = 1e-8
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
    rc2 = grp.solve_scene(sc, p)
    assert rc1 == rc2 == abi.SUCCESS
    assert abs(int(single.stats.solveData[1]) - int(grp.stats.solveData[1])) <= 1      # (at tol 1e-8 the partial sums' grouping may move the count by one)
    _views_hold_the_global_cell_labels(single, grp, sc)
    for a in range(3):
        assert np.array_equal(grp.valid[a], single.valid[a])
        scale = max(np.abs(single.vel[a]).max(), 1e-30)
        assert np.abs(grp.vel[a] - single.vel[a]).max() <= 1e-6 * scale
    grp.close()
    single.close()
