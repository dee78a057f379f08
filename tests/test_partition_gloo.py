"""world_size-2 (and 3) gloo test of the host-side slab decomposition on CPU: every rank builds its slab (+halo)
from the global scene, runs the CPU oracle's classification on it and must reproduce the global labels, indices'
counts and weights inside its owned range; owned counts all-reduce to the global counts."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polystokes_amd import _abi as abi
from polystokes_amd import partition, scenes


def _scene():
    sc0, p = scenes.coil(32)
    nz, n = 64, 32
    z, y, x = np.meshgrid((np.arange(nz) + 0.5) * sc0.dx, (np.arange(n) + 0.5) * sc0.dx, (np.arange(n) + 0.5) * sc0.dx, indexing="ij")
    col = np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2) - 0.2
    surface = np.minimum(col, z - 0.3)
    sc = abi.Scene(n, n, nz, sc0.dx, sc0.dt, 1000.0, [0.0, 0.0, -1.0], surface, z - 2 * sc0.dx, 100.0)
    return sc, p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ps_oracle
        sc, p = _scene()
        og = ps_oracle.Oracle()
        og.run(sc, p, solve=False)
        sl = partition.make_slab(sc.nz, world, rank, p.tileSize)
        loc = partition.local_scene(sc, sl)
        ol = ps_oracle.Oracle()
        ol.run(loc, p, solve=False)
        ok = True
        own_active = 0
        for name, extra in (("centerLabels", 0), ("faceXLabels", 0), ("faceYLabels", 0), ("edgeXYLabels", 0),
                            ("centerLiquidWeights", 0), ("faceZLabels", 0), ("edgeYZLabels", 0)):
            g = og.array(name)
            l = ol.array(name)
            plane = g.size // (sc.nz + (1 if name.startswith(("faceZ", "edgeYZ", "edgeXZ")) else 0))
            g = g.reshape(-1, plane)
            l = l.reshape(-1, plane)
            a, b = sl.zLoOwned, sl.zHiOwned
            ok = ok and np.array_equal(l[a:b], g[sl.z0:sl.z1])
            if name == "centerLabels":
                own_active = int((l[a:b] == abi.ACTIVEFLUID).sum())
        t = torch.tensor([own_active, int(ok)], dtype=torch.int64)
        dist.all_reduce(t)
        total_active = int((og.array("centerLabels") == abi.ACTIVEFLUID).sum())
        q.put((rank, int(t[0]) == total_active, int(t[1]) == world))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_reproduce_global_classification(world, oracle_mod):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(60)
    assert all(a and b for _, a, b in res), res


def test_slab_ranges_and_alignment():
    assert partition.slab_ranges(256, 8, 16) == [(32 * r, 32 * (r + 1)) for r in range(8)]
    assert partition.slab_ranges(80, 2, 16) == [(0, 48), (48, 80)]
    assert partition.alignment(16) == 16 and partition.alignment(8) == 16 and partition.alignment(12) == 48
    with pytest.raises(ValueError):
        partition.slab_ranges(32, 4, 16)
    sl = partition.make_slab(96, 3, 1, 16)
    assert (sl.z0, sl.z1, sl.g0, sl.zLoOwned, sl.zHiOwned, sl.hasLower, sl.hasUpper, sl.nz_local) == (32, 64, 16, 16, 48, 1, 1, 64)


@pytest.mark.parametrize("world", [2, 4])
def test_cavity_slab_is_the_cut_of_the_global_weak_scaling_cavity(world):
    """bench.py --gpus N builds each rank's piece directly (scenes.cavity_slab); it must be exactly the slab that
    partition.local_scene cuts out of the global n x n x (n*world) cavity with the lid on the global top plane."""
    from polystokes_amd import _abi as abi
    from polystokes_amd import partition, scenes
    n = 32
    sc0, p0 = scenes.cavity(n)
    velx = np.zeros((n * world, n, n + 1), np.float32)
    velx[n * world - 1] = 1.0
    glob = abi.Scene(n, n, n * world, sc0.dx, sc0.dt, 1.0, [velx, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0)
    for rank in range(world):
        sc, p, sl = scenes.cavity_slab(n, world, rank)
        ref = partition.local_scene(glob, partition.make_slab(n * world, world, rank, p.tileSize))
        assert (sc.nx, sc.ny, sc.nz) == (ref.nx, ref.ny, ref.nz)
        assert (sl.z0, sl.z1, sl.g0, sl.nz_local, sl.zLoOwned, sl.zHiOwned) == (rank * n, (rank + 1) * n, max(0, rank * n - 16),
                                                                              n + 16 * ((rank > 0) + (rank < world - 1)),
                                                                              16 * (rank > 0), 16 * (rank > 0) + n)
        assert sc.dx == ref.dx and sc.dt == ref.dt and sc.density == ref.density
        for a in range(3):
            assert np.array_equal(sc.vel[a], ref.vel[a]) and np.array_equal(sc.collisionvel[a], ref.collisionvel[a])
        for f in ("surface", "collision", "viscosity"):
            assert np.array_equal(getattr(sc, f), getattr(ref, f))
        assert p.tileSize == p0.tileSize and p.tilePadding == p0.tilePadding


def _brick_scene():
    """a 64 x 64 x 32 coil-like scene: the free surface crosses cuts along x and along y"""
    sc0, p = scenes.coil(32)
    nx, ny, nz = 64, 64, 32
    z, y, x = np.meshgrid((np.arange(nz) + 0.5) * sc0.dx, (np.arange(ny) + 0.5) * sc0.dx, (np.arange(nx) + 0.5) * sc0.dx, indexing="ij")
    col = np.sqrt((x - 1.0) ** 2 + (y - 0.9) ** 2) - 0.45           # a column across all four bricks
    surface = np.minimum(col, z - 0.3)
    return abi.Scene(nx, ny, nz, sc0.dx, sc0.dt, 1000.0, [0.0, 0.0, -1.0], surface, z - 2 * sc0.dx, 100.0), p


def _brick_worker(rank, world, port, q, dims):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ps_oracle
        sc, p = _brick_scene()
        og = ps_oracle.Oracle()
        og.run(sc, p, solve=False)
        b = partition.make_brick((sc.nx, sc.ny, sc.nz), dims, rank, p.tileSize)
        ol = ps_oracle.Oracle()
        ol.run(partition.local_scene_brick(sc, b), p, solve=False)
        gs, ls = abi.grid_shapes(sc.nx, sc.ny, sc.nz), abi.grid_shapes(*b.n_local)
        ok = True
        own_active = 0
        for name, key in (("centerLabels", "center"), ("faceXLabels", "faceX"), ("faceYLabels", "faceY"), ("faceZLabels", "faceZ"), ("edgeXYLabels", "edgeXY"),
                          ("edgeYZLabels", "edgeYZ"), ("edgeXZLabels", "edgeXZ"), ("centerLiquidWeights", "center")):
            g = og.array(name).reshape(gs[key])
            l = ol.array(name).reshape(ls[key])
            own = l[b.lo[2]:b.hi[2], b.lo[1]:b.hi[1], b.lo[0]:b.hi[0]]          # (the entities at the lower index of every owned cell)
            ok = ok and np.array_equal(own, g[b.g0[2]:b.g1[2], b.g0[1]:b.g1[1], b.g0[0]:b.g1[0]])
            if name == "centerLabels":
                own_active = int((own == abi.ACTIVEFLUID).sum())
        t = torch.tensor([own_active, int(ok)], dtype=torch.int64)
        dist.all_reduce(t)
        total_active = int((og.array("centerLabels") == abi.ACTIVEFLUID).sum())
        q.put((rank, int(t[0]) == total_active, int(t[1]) == world))
    finally:
        dist.destroy_process_group()


def test_bricks_reproduce_global_classification(oracle_mod):
    """The decomposition along x AND y (ps_set_brick; partition.make_brick): four gloo ranks, every one classifies its brick (+ halo
    blocks) with the CPU oracle and reproduces the global labels and weights on its owned box; owned counts all-reduce to the global one."""
    world, dims = 4, (2, 2, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_brick_worker, args=(r, world, 29671, q, dims)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(60)
    assert all(a and b for _, a, b in res), res


def test_brick_ranges_and_generators():
    b = partition.make_brick((64, 96, 64), (2, 3, 2), 1 + 2 * (2 + 3 * 1), 16)       # the brick at (1, 2, 1)
    assert b.coord == (1, 2, 1) and b.world == 12
    assert (b.g0, b.g1) == ([32, 64, 32], [64, 96, 64])
    assert (b.hasLower, b.hasUpper) == ([1, 1, 1], [0, 0, 0])
    assert (b.lo, b.hi, b.n_local, b.origin) == ([16, 16, 16], [48, 48, 48], [48, 48, 48], [16, 48, 16])
    # the rank-local generators of bench.py --bricks are the cuts of the global scene
    for name in ("cavity", "coil", "spheres"):
        glob, p = getattr(scenes, name)(48)
        for rank in range(4):
            loc, pl, bk = scenes.scene_brick(name, 48, (2, 2, 1) if name != "coil" else (1, 2, 2), rank)
            ref = partition.local_scene_brick(glob, bk)
            assert (loc.nx, loc.ny, loc.nz) == (ref.nx, ref.ny, ref.nz) == tuple(bk.n_local)
            for f in ("surface", "collision", "viscosity"):
                assert np.array_equal(getattr(loc, f), getattr(ref, f)), (name, rank, f)
            for a in range(3):
                assert np.array_equal(loc.vel[a], ref.vel[a]) and np.array_equal(loc.collisionvel[a], ref.collisionvel[a]), (name, rank, a)
    # weak scaling: every rank owns n^3 cells of the (n dx) x (n dy) x (n dz) cavity, lid on the global top plane
    n, dims = 32, (2, 1, 2)
    velx = np.zeros((n * 2, n, n * 2 + 1), np.float32)
    velx[-1] = 1.0
    glob = abi.Scene(n * 2, n, n * 2, 1.0 / n, 1.0e-2, 1.0, [velx, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0)
    for rank in range(4):
        loc, pl, bk = scenes.scene_brick("cavity", n, dims, rank, weak=True)
        ref = partition.local_scene_brick(glob, bk)
        assert [bk.g1[a] - bk.g0[a] for a in range(3)] == [n, n, n]
        assert loc.dx == ref.dx and np.array_equal(loc.vel[0], ref.vel[0]) and np.array_equal(loc.surface, ref.surface)

