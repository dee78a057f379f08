"""Scenes shared by the multi-process rank script and its test (deterministic: both sides build the same arrays)."""
import numpy as np

from polystokes_amd import _abi as abi
from polystokes_amd import scenes


def tall_cavity(nx, nz, precond=abi.PRE_IDENTITY, tile=16):
    sc0, p = scenes.cavity(nx, tile=tile, precond=precond)
    velx = np.zeros((nz, nx, nx + 1), np.float32)
    velx[nz - 1] = 1.0
    velx[nz // 2, :, : nx // 2] = -0.5          # something to do near the cut as well
    sc = abi.Scene(nx, nx, nz, sc0.dx, sc0.dt, 1.0, [velx, 0.0, 0.0], -1.0, 1.0, 1.0, name=f"tall{nx}x{nz}")
    return sc, p


def tall_coil(n, nz):
    """liquid column + pool with a free surface crossing the cuts, solid floor"""
    sc0, p = scenes.coil(n)
    z, y, x = np.meshgrid((np.arange(nz) + 0.5) * sc0.dx, (np.arange(n) + 0.5) * sc0.dx, (np.arange(n) + 0.5) * sc0.dx, indexing="ij")
    col = np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2) - 0.2
    surface = np.minimum(col, z - 0.3)
    collision = z - 2 * sc0.dx
    return abi.Scene(n, n, nz, sc0.dx, sc0.dt, 1000.0, [0.0, 0.0, -1.0], surface, collision, 100.0, name=f"tallcoil{n}x{nz}"), p


def make(case):
    base = case.replace("_interrupt", "").replace("_failrank", "")
    if base == "cavity_w2":
        return tall_cavity(32, 64)
    if base == "cavity_w3_jacobi":
        return tall_cavity(24, 96, precond=abi.PRE_DIAGONAL)
    if base == "coil_w2":
        return tall_coil(32, 64)
    if base == "cavity_w2_chebyshev":
        return tall_cavity(32, 64, precond=abi.PRE_CHEBYSHEV)
    if base == "cavity_w2_bicgstab":
        sc, p = tall_cavity(24, 64)
        p.maxSolverIterations = 12
        p.tolerance = 5e-2
        return sc, p
    if base in ("cavity64_b2x2x1", "cavity64_b2x2x2"):          # bricks: cuts along x and y, four processes; along all three axes, eight ranks
        return scenes.cavity(64, tile=16, precond=abi.PRE_DIAGONAL)
    if base == "cavity32_b2x2x2":          # the small eight-rank case (tile 8: bricks of 16^3 cells)
        sc, p = scenes.cavity(32, tile=8, precond=abi.PRE_DIAGONAL)
        p.tolerance = 1e-2                 # eight ranks time-slice one GPU: every iteration costs milliseconds there
        return sc, p
    if base == "cavity_b2x1x2":
        return tall_cavity(32, 64)
    raise KeyError(case)


DIMS = {"cavity64_b2x2x1": (2, 2, 1), "cavity_b2x1x2": (2, 1, 2), "cavity64_b2x2x2": (2, 2, 2), "cavity32_b2x2x2": (2, 2, 2)}      # brick cases: ranks per axis
WORLD = {"cavity64_b2x2x2": 8, "cavity32_b2x2x2": 8, "cavity64_b2x2x1": 4, "cavity_b2x1x2": 4, "cavity_w2": 2, "cavity_w3_jacobi": 3, "coil_w2": 2, "cavity_w2_bicgstab": 2, "cavity_w2_chebyshev": 2}
