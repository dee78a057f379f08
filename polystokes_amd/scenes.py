"""Deterministic synthetic scenes as numpy fields in the C-ABI layout.

The scenes BASELINE.json names (viscous_beam, honey_coil, armadillos) are Houdini .hipnc files that are
not in the reference snapshot; these are stand-ins with the same character.  Conventions:
surface SDF < 0 inside liquid; collision SDF < 0 inside solids (Houdini convention, negateCollision=1).
All arrays are float32, shape (z, y, x).
"""
import numpy as np

from ._abi import Scene, default_params


def _centers(nx, ny, nz, dx, k0=0):
    """cell-centre coordinates of layers k0 .. k0 + nz - 1 (a z-slab of a taller grid generates exactly the global values)"""
    # open (broadcastable) grids: every scene below is built from elementwise operations, so the fields come out bit-identical to
    # the dense-meshgrid form at half the time and a fraction of the memory (Scene() broadcasts to the full shapes)
    z, y, x = np.meshgrid((np.arange(k0, k0 + nz) + 0.5) * dx, (np.arange(ny) + 0.5) * dx, (np.arange(nx) + 0.5) * dx,
                          indexing="ij", sparse=True)
    return x, y, z


def _box_sdf(x, y, z, lo, hi):
    """Exact signed distance to an axis-aligned box [lo, hi] (negative inside)."""
    c = [(lo[a] + hi[a]) * 0.5 for a in range(3)]
    h = [(hi[a] - lo[a]) * 0.5 for a in range(3)]
    q = [np.abs(p - c[a]) - h[a] for a, p in enumerate((x, y, z))]
    outside = np.sqrt(sum(np.maximum(qa, 0.0) ** 2 for qa in q))
    inside = np.minimum(np.maximum(np.maximum(q[0], q[1]), q[2]), 0.0)
    return outside + inside


def beam(n=32):
    """Config 1 stand-in: viscous beam clamped to a wall.  Uniform Stokes (doReducedRegions=0)."""
    dx, dt = 1.0 / n, 1.0 / 24.0
    s = n / 32.0
    x, y, z = _centers(n, n, n, dx)
    surface = _box_sdf(x, y, z, (0.0, 12 * s * dx, 12 * s * dx), (28 * s * dx, 20 * s * dx, 20 * s * dx))
    collision = x - 4 * s * dx            # solid (negative) for x < 4 cells
    vel = [0.0, np.float32(-9.8 * dt), 0.0]
    sc = Scene(n, n, n, dx, dt, 1000.0, vel, surface, collision, 1.0e3, name=f"beam{n}")
    return sc, default_params(doReducedRegions=0)


def coil(n=64, tile=16, pad=2, zrange=None):
    """Config 2/4 stand-in: a liquid column falling into a pool over a solid floor.
    zrange = (k0, nz): only the layers k0 .. k0 + nz - 1 of the n^3 scene (a rank's slab + halo, see scene_slab)."""
    dx, dt = 1.0 / n, 1.0 / 24.0
    s = n / 128.0
    k0, nzl = zrange if zrange is not None else (0, n)
    x, y, z = _centers(n, n, nzl, dx, k0)
    cx = cz = 0.5
    r = 12 * s * dx
    col = np.maximum(np.sqrt((x - cx) ** 2 + (z - cz) ** 2) - r, -(y - 0.0))   # infinite-up cylinder
    pool = _box_sdf(x, y, z, (-1.0, -1.0, -1.0), (2.0, (24 * s + 2) * dx, 2.0))
    surface = np.minimum(col, pool)
    collision = y - 2 * dx                 # floor slab of 2 cells
    vel = [0.0, -1.0, 0.0]
    sc = Scene(n, n, nzl, dx, dt, 1000.0, vel, surface, collision, 100.0, name=f"coil{n}")
    return sc, default_params(tileSize=tile, tilePadding=pad)


def cavity(n=64, tile=16, pad=2, precond=1):
    """Config 3 (roofline run): all-liquid box, lid row z=nz-1 moves with u_x=1."""
    dx, dt = 1.0 / n, 1.0e-2
    surface = np.float32(-1.0)
    collision = np.float32(1.0)
    velx = np.zeros((n, n, n + 1), dtype=np.float32)
    velx[n - 1, :, :] = 1.0
    sc = Scene(n, n, n, dx, dt, 1.0, [velx, 0.0, 0.0], surface, collision, 1.0, name=f"cavity{n}")
    return sc, default_params(tileSize=tile, tilePadding=pad, preconditioner=precond)


def spheres(n=64, tile=16, pad=2, nspheres=8, seed=12345, zrange=None):
    """Config 5 stand-in: half-filled pool with moving solid spheres, mixed uniform/reduced regions.
    zrange = (k0, nz): only the layers k0 .. k0 + nz - 1 of the n^3 scene (a rank's slab + halo, see scene_slab)."""
    dx, dt = 1.0 / n, 1.0 / 48.0
    rng = np.random.RandomState(seed)
    k0, nzl = zrange if zrange is not None else (0, n)
    x, y, z = _centers(n, n, nzl, dx, k0)
    surface = y - 0.5
    collision = np.full_like(x, 10.0)
    fx = (np.arange(n + 1) * dx, (np.arange(n) + 0.5) * dx)
    fz = (np.arange(k0, k0 + nzl + 1) * dx, (np.arange(k0, k0 + nzl) + 0.5) * dx)
    cv = [np.zeros((nzl, n, n + 1), np.float32), np.zeros((nzl, n + 1, n), np.float32), np.zeros((nzl + 1, n, n), np.float32)]
    for _ in range(nspheres):
        c = rng.uniform(0.2, 0.8, 3)
        c[1] = rng.uniform(0.25, 0.55)
        rad = rng.uniform(0.06, 0.11)
        v = rng.uniform(-1.0, 1.0, 3)
        collision = np.minimum(collision, np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - rad)
        for a in range(3):
            zz, yy, xx = np.meshgrid(fz[0] if a == 2 else fz[1], fx[0] if a == 1 else fx[1],
                                     fx[0] if a == 0 else fx[1], indexing="ij")
            inside = np.sqrt((xx - c[0]) ** 2 + (yy - c[1]) ** 2 + (zz - c[2]) ** 2) < rad + 1.5 * dx
            cv[a][inside] = v[a]
    vel = [0.0, np.float32(-9.8 * dt), 0.0]
    sc = Scene(n, n, nzl, dx, dt, 1000.0, vel, surface, collision, 1.0e4, collisionvel=cv, name=f"spheres{n}")
    return sc, default_params(tileSize=tile, tilePadding=pad)


def blob(nx=24, ny=20, nz=28, seed=0, tile=8, pad=2, variable_viscosity=True):
    """Fuzz scene: irregular liquid blob (sum of sines), a solid sphere, smooth random velocity and
    viscosity, non-cubic grid whose sizes are not multiples of the tile or the 16^3 voxel tile."""
    rng = np.random.RandomState(seed)
    dx, dt = 1.0 / max(nx, ny, nz), 1.0 / 30.0
    x, y, z = _centers(nx, ny, nz, dx)
    ph = rng.uniform(0, 2 * np.pi, 6)
    cen = np.array([nx, ny, nz]) * dx * 0.5
    rad = 0.36 * min(nx, ny, nz) * dx
    d = np.sqrt((x - cen[0]) ** 2 + (y - cen[1]) ** 2 + (z - cen[2]) ** 2)
    surface = d - rad * (1.0 + 0.25 * np.sin(7 * x + ph[0]) * np.sin(6 * y + ph[1]) * np.sin(5 * z + ph[2]))
    sc_c = cen + rng.uniform(-0.15, 0.15, 3)
    collision = np.sqrt((x - sc_c[0]) ** 2 + (y - sc_c[1]) ** 2 + (z - sc_c[2]) ** 2) - 0.12
    visc = 10.0 * (1.5 + np.sin(9 * x + ph[3]) * np.cos(8 * z + ph[4])) if variable_viscosity else 10.0
    shp = [(nz, ny, nx + 1), (nz, ny + 1, nx), (nz + 1, ny, nx)]
    vel = []
    for a in range(3):
        zz, yy, xx = np.meshgrid(np.arange(shp[a][0]) * dx, np.arange(shp[a][1]) * dx, np.arange(shp[a][2]) * dx,
                                 indexing="ij")
        vel.append(np.sin(5 * xx + ph[a]) * np.cos(4 * yy - ph[5]) + 0.3 * np.sin(6 * zz))
    cv = [np.full(shp[a], 0.2 * (a - 1), np.float32) for a in range(3)]
    sc = Scene(nx, ny, nz, dx, dt, 900.0, vel, surface, collision, visc, collisionvel=cv, name=f"blob{seed}")
    return sc, default_params(tileSize=tile, tilePadding=pad)


def droplet(n=24, tile=8, pad=2, radius=0.33):
    """A liquid ball floating in air, no solids: free surface all around (rigid-motion KAT)."""
    dx, dt = 1.0 / n, 1.0 / 24.0
    x, y, z = _centers(n, n, n, dx)
    surface = np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - radius
    sc = Scene(n, n, n, dx, dt, 1000.0, [0.0, 0.0, 0.0], surface, np.float32(10.0), 50.0, name=f"droplet{n}")
    return sc, default_params(tileSize=tile, tilePadding=pad)


def scene_slab(name, n, world, rank, tile=16, pad=2, precond=1):
    """Rank-local piece (slab + halo) of the n^3 scene `name` ("coil" | "spheres" | "cavity") for STRONG scaling: the
    global problem is fixed, every rank generates only its own layers — bit-identical to cutting the global arrays
    (tests/test_scenes_slab.py).  Returns (local Scene, params, Slab)."""
    from . import partition
    sl = partition.make_slab(n, world, rank, tile)
    if name == "cavity":
        sc, p = cavity(n, tile, pad, precond)
        return partition.local_scene(sc, sl), p, sl
    fn = {"coil": coil, "spheres": spheres}[name]
    sc, p = fn(n, tile, pad, zrange=(sl.g0, sl.nz_local))
    p.preconditioner = precond
    return sc, p, sl


def cavity_slab(n, world, rank, tile=16, pad=2, precond=1):
    """Rank-local piece of the weak-scaling cavity n x n x (n*world) (lid on the global top plane), generated
    directly without materialising the global grid.  Returns (local Scene, params, Slab)."""
    from . import partition
    dx, dt = 1.0 / n, 1.0e-2
    sl = partition.make_slab(n * world, world, rank, tile)
    nzl = sl.nz_local
    velx = np.zeros((nzl, n, n + 1), dtype=np.float32)
    top = n * world - 1 - sl.g0          # local index of the global lid layer
    if 0 <= top < nzl:
        velx[top, :, :] = 1.0
    sc = Scene(n, n, nzl, dx, dt, 1.0, [velx, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0, name=f"cavity{n}x{world}.r{rank}")
    return sc, default_params(tileSize=tile, tilePadding=pad, preconditioner=precond), sl


def scene_brick(name, n, dims, rank, tile=16, pad=2, precond=1, weak=False):
    """Rank-local piece (brick + halo) for a dims[0] x dims[1] x dims[2] decomposition (ps_set_brick).  weak = False: the n^3 scene `name`
    cut into bricks (strong scaling); weak = True (cavity only): every rank owns n^3 cells of the (n dims[0]) x (n dims[1]) x (n dims[2])
    cavity with the lid on the global top plane and cells of size 1 / n.  Returns (local Scene, params, Brick)."""
    from . import partition
    G = (n * dims[0], n * dims[1], n * dims[2]) if weak else (n, n, n)
    b = partition.make_brick(G, dims, rank, tile)
    nx, ny, nz = b.n_local
    if name == "cavity":
        dx, dt = 1.0 / n, 1.0e-2
        velx = np.zeros((nz, ny, nx + 1), dtype=np.float32)
        top = G[2] - 1 - b.origin[2]          # local index of the global lid layer
        if 0 <= top < nz:
            velx[top, :, :] = 1.0
        sc = Scene(nx, ny, nz, dx, dt, 1.0, [velx, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0, name=f"cavity{G[0]}x{G[1]}x{G[2]}.b{rank}")
        return sc, default_params(tileSize=tile, tilePadding=pad, preconditioner=precond), b
    if weak:
        raise ValueError("weak scaling is defined for the cavity scene")
    fn = {"coil": coil, "spheres": spheres}[name]
    full, p = fn(n, tile, pad, zrange=(b.origin[2], nz))      # the rank's z-layers of the global scene, then its x / y range
    p.preconditioner = precond
    zb = partition.Brick.__new__(partition.Brick)
    zb.__dict__.update(b.__dict__)
    zb.origin = [b.origin[0], b.origin[1], 0]
    return partition.local_scene_brick(full, zb), p, b

