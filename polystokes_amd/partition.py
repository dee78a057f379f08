"""Host-side slab / brick decomposition for the multi-GPU path.

The global grid is cut along z at multiples of `align` = lcm(16, tileSize); rank r owns layers
[z0, z1) and is handed those layers plus one halo block of `align` layers on each interior side, so that every
label inside the owned range comes out identical to the global classification (boundary layers reach at most
L+S cells, tiles are aligned, regions are tile-local).
"""
import math

import numpy as np

from ._abi import Scene


class Slab:
    def __init__(self, rank, world, z0, z1, lo_halo, hi_halo):
        self.rank, self.world = rank, world
        self.z0, self.z1 = z0, z1                    # owned global layers [z0, z1)
        self.lo_halo, self.hi_halo = lo_halo, hi_halo
        self.g0 = z0 - lo_halo                       # global layer of local layer 0
        self.zLoOwned, self.zHiOwned = lo_halo, lo_halo + (z1 - z0)
        self.hasLower, self.hasUpper = int(rank > 0), int(rank < world - 1)
        self.nz_local = (z1 - z0) + lo_halo + hi_halo


def slab_ranges(nz, world, align):
    """Contiguous owned ranges, multiples of `align`, as even as possible; the last one takes the remainder."""
    blocks = nz // align
    if blocks < world:
        raise ValueError(f"{nz} layers cannot be cut into {world} slabs of multiples of {align}")
    base, extra = divmod(blocks, world)
    out, z = [], 0
    for r in range(world):
        nb = base + (1 if r < extra else 0)
        z1 = z + nb * align
        if r == world - 1:
            z1 = nz
        out.append((z, z1))
        z = z1
    return out


def alignment(tile_size, do_tile=True):
    return 16 * tile_size // math.gcd(16, tile_size) if do_tile else 16


def make_slab(nz, world, rank, tile_size=16):
    al = alignment(tile_size)
    z0, z1 = slab_ranges(nz, world, al)[rank]
    return Slab(rank, world, z0, z1, al if rank > 0 else 0, al if rank < world - 1 else 0)


def local_scene(scene, slab):
    """The slab (+halo) of `scene` as an ordinary Scene."""
    a, b = slab.g0, slab.g0 + slab.nz_local
    cut = lambda arr, extra=0: np.ascontiguousarray(arr[a:b + extra])
    sc = Scene(scene.nx, scene.ny, slab.nz_local, scene.dx, scene.dt, scene.density,
               [cut(scene.vel[0]), cut(scene.vel[1]), cut(scene.vel[2], 1)],
               cut(scene.surface), cut(scene.collision), cut(scene.viscosity),
               collisionvel=[cut(scene.collisionvel[0]), cut(scene.collisionvel[1]), cut(scene.collisionvel[2], 1)],
               name=f"{scene.name}.r{slab.rank}")
    return sc


def merge_faces(global_out, local_out, owned_mask, slab, axis):
    """Copy the faces a rank is responsible for (owned_mask > 0, local layout) into the global array."""
    a = slab.g0
    n = local_out.shape[0]
    view = global_out[a:a + n]
    m = owned_mask.reshape(local_out.shape) > 0
    view[m] = local_out[m]


# ---- bricks: the same along all three axes (ps_set_brick) ----------------------------------------------
class Brick:
    """Rank `rank` of a dims[0] x dims[1] x dims[2] decomposition: owned global cells [g0, g1) per axis, one halo block of `align`
    cells on every side that has a neighbour.  rank = c0 + dims[0] * (c1 + dims[1] * c2)."""

    def __init__(self, rank, dims, ranges, align):
        self.rank, self.dims, self.world = rank, tuple(dims), dims[0] * dims[1] * dims[2]
        self.coord = (rank % dims[0], (rank // dims[0]) % dims[1], rank // (dims[0] * dims[1]))
        self.g0 = [ranges[a][self.coord[a]][0] for a in range(3)]          # owned global range (x, y, z)
        self.g1 = [ranges[a][self.coord[a]][1] for a in range(3)]
        self.hasLower = [int(self.coord[a] > 0) for a in range(3)]
        self.hasUpper = [int(self.coord[a] + 1 < dims[a]) for a in range(3)]
        self.lo_halo = [align if self.hasLower[a] else 0 for a in range(3)]
        self.hi_halo = [align if self.hasUpper[a] else 0 for a in range(3)]
        self.origin = [self.g0[a] - self.lo_halo[a] for a in range(3)]      # global index of the local cell 0
        self.lo = list(self.lo_halo)                                         # owned range in local coordinates
        self.hi = [self.lo_halo[a] + self.g1[a] - self.g0[a] for a in range(3)]
        self.n_local = [self.hi[a] + self.hi_halo[a] for a in range(3)]
        # slab-compatible names (tests written for slabs read these)
        self.z0, self.z1, self.zLoOwned, self.zHiOwned, self.nz_local = self.g0[2], self.g1[2], self.lo[2], self.hi[2], self.n_local[2]


def make_brick(n, dims, rank, tile_size=16):
    """n = (nx, ny, nz) cells of the global grid."""
    al = alignment(tile_size)
    ranges = [slab_ranges(n[a], dims[a], al) for a in range(3)]
    return Brick(rank, dims, ranges, al)


def local_scene_brick(scene, b):
    """The brick (+halo) of `scene` as an ordinary Scene (arrays are (z, y, x), x fastest)."""
    ox, oy, oz = b.origin
    nx, ny, nz = b.n_local

    def cut(arr, ex=0, ey=0, ez=0):
        return np.ascontiguousarray(arr[oz:oz + nz + ez, oy:oy + ny + ey, ox:ox + nx + ex])

    faces = lambda v: [cut(v[0], ex=1), cut(v[1], ey=1), cut(v[2], ez=1)]
    return Scene(nx, ny, nz, scene.dx, scene.dt, scene.density, faces(scene.vel), cut(scene.surface), cut(scene.collision), cut(scene.viscosity),
                 collisionvel=faces(scene.collisionvel), name=f"{scene.name}.b{b.rank}")


def merge_faces_brick(global_out, local_out, owned_mask, b, axis):
    """Copy the faces a rank is responsible for (owned_mask > 0, local layout) into the global array."""
    ox, oy, oz = b.origin
    nz, ny, nx = local_out.shape
    view = global_out[oz:oz + nz, oy:oy + ny, ox:ox + nx]
    m = owned_mask.reshape(local_out.shape) > 0
    view[m] = local_out[m]
