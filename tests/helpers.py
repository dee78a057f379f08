import numpy as np
import scipy.sparse as sp

from polystokes_amd import _abi as abi
from polystokes_amd import scenes


def basis_rows(off, axis):
    """Vectorised C_a(x) (exec/HDK_PolyStokesSolver.cpp:2105-2149).  off: (n,3), axis: (n,) -> (n,26)."""
    off = np.asarray(off, np.float64)
    axis = np.asarray(axis)
    x, y, z = off[:, 0], off[:, 1], off[:, 2]
    n = len(x)
    C = np.zeros((n, 26))
    quad = np.stack([x, y, z, x * x, x * y, x * z, y * y, y * z, z * z], axis=1)
    m0, m1, m2 = axis == 0, axis == 1, axis == 2
    C[m0, 0] = 1
    C[np.ix_(m0, range(3, 12))] = quad[m0]
    C[m1, 1] = 1
    C[np.ix_(m1, range(12, 21))] = quad[m1]
    C[m2, 2] = 1
    C[m2, 3] = -z[m2]
    C[m2, 6] = -2 * x[m2] * z[m2]
    C[m2, 7] = -y[m2] * z[m2]
    C[m2, 8] = -0.5 * z[m2] * z[m2]
    C[m2, 13] = -z[m2]
    C[m2, 16] = -x[m2] * z[m2]
    C[m2, 18] = -2 * y[m2] * z[m2]
    C[m2, 19] = -0.5 * z[m2] * z[m2]
    C[m2, 21] = x[m2]
    C[m2, 22] = y[m2]
    C[m2, 23] = x[m2] * x[m2]
    C[m2, 24] = x[m2] * y[m2]
    C[m2, 25] = y[m2] * y[m2]
    return C


def materialise_blocks(solver):
    """G, Dt, JG, JDt (scipy CSR) from the device's factored storage S = [G Dt; Ghat Dhat] and the
    on-the-fly basis — what exportComponentMatrices() writes (Solver.cpp:557-560)."""
    S, _ = solver.S_matrices()
    nA, nP, R = solver.nA, solver.nP, solver.nRegions
    G, Dt = S[:nA, :nP].tocsr(), S[:nA, nP:].tocsr()
    packed = solver.array("reducedRowFace")
    reg = solver.array("reducedRowRegion")
    nRr = len(packed)
    if nRr == 0:
        JG = sp.csr_matrix((R * 26, nP))
        JDt = sp.csr_matrix((R * 26, S.shape[1] - nP))
        return G, Dt, JG, JDt
    i, j, k, a = packed & 1023, (packed >> 10) & 1023, (packed >> 20) & 1023, packed >> 30
    pos = np.stack([i, j, k], axis=1).astype(np.float64)
    pos[np.arange(nRr), a] -= 0.5
    com = solver.array("reducedRegionCOM").reshape(-1, 3)
    off = pos * solver.scene.dx - com[reg]
    Cm = basis_rows(off, a)
    rows = (reg[:, None] * 26 + np.arange(26)[None, :]).ravel()
    cols = np.repeat(np.arange(nRr), 26)
    J = sp.csr_matrix((Cm.ravel(), (rows, cols)), shape=(R * 26, nRr))
    Sr = S[nA:, :]
    JG = (J @ Sr[:, :nP]).tocsr()
    JDt = (J @ Sr[:, nP:]).tocsr()
    return G, Dt, JG, JDt


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    s = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return d / s


# ---- the system vector as grid fields (for comparisons between decompositions) -------------------------------------------
DOF_KINDS = ("p", "txx", "tyy", "tzz", "eYZ", "eXZ", "eXY")


def dof_field(solver, kind, x, dtype=np.float32):
    """One kind of DOF of a system vector x (reference numbering, Solver.h:586-606: pressures, then txx, tyy, tzz by cell, then the
    YZ / XZ / XY edge stresses) as a dense array over its sample grid, NaN where there is no DOF."""
    cidx = solver.array("centerActiveIndices")
    nP = int((cidx >= 0).sum())                    # (a slab rank's own count: its stats object holds the group's)
    if kind in ("p", "txx", "tyy", "tzz"):
        idx = cidx
        off = {"p": 0, "txx": nP, "tyy": 2 * nP, "tzz": 3 * nP}[kind]
    else:
        nE = [int((solver.array(s + "ActiveIndices") >= 0).sum()) for s in ("edgeYZ", "edgeXZ")]
        idx = solver.array({"eYZ": "edgeYZ", "eXZ": "edgeXZ", "eXY": "edgeXY"}[kind] + "ActiveIndices")
        off = 4 * nP + {"eYZ": 0, "eXZ": nE[0], "eXY": nE[0] + nE[1]}[kind]
    out = np.full(idx.shape, np.nan, dtype)
    m = idx >= 0
    out[m] = x[idx[m].astype(np.int64) + off]
    return out


def merge_dof_field(global_out, local, slab, kind):
    """Copy the DOFs a rank OWNS (cells and XY edges of its layers; YZ / XZ edges on its planes, the plane on a cut belonging to
    the rank above: DESIGN.md section 6) from its local field (x-fastest, local layers) into the global one."""
    nzl = slab.nz_local + (1 if kind in ("eYZ", "eXZ") else 0)
    loc = local.reshape(nzl, -1)
    glo = global_out.reshape(global_out.size // loc.shape[1], loc.shape[1])
    k0, k1 = slab.zLoOwned, slab.zHiOwned
    if kind in ("eYZ", "eXZ") and not slab.hasUpper:
        k1 += 1                                   # the top plane of the whole domain
    glo[slab.g0 + k0:slab.g0 + k1] = loc[k0:k1]


def merge_dof_field_brick(global_out, local, b, kind, global_n):
    """The same for a brick (partition.Brick): along every axis a kind lives in the cell layers or on the planes (plane-type axes per
    kind below), owned range [lo, hi) plus the last plane of the domain where there is no upper neighbour.  global_n = (nx, ny, nz)."""
    planes = {"p": (), "txx": (), "tyy": (), "tzz": (), "eYZ": (1, 2), "eXZ": (0, 2), "eXY": (0, 1)}[kind]
    lshape = tuple(b.n_local[a] + (1 if a in planes else 0) for a in (2, 1, 0))
    gshape = tuple(global_n[a] + (1 if a in planes else 0) for a in (2, 1, 0))
    loc, glo = local.reshape(lshape), global_out.reshape(gshape)
    sl_l, sl_g = [], []
    for a in (2, 1, 0):
        k0, k1 = b.lo[a], b.hi[a] + (1 if (a in planes and not b.hasUpper[a]) else 0)
        sl_l.append(slice(k0, k1)); sl_g.append(slice(b.origin[a] + k0, b.origin[a] + k1))
    glo[tuple(sl_g)] = loc[tuple(sl_l)]



# ---- the random brick cases of scripts/fuzz_bricks.py (seed -> scene, parameters, decomposition) ---------------------------------
FUZZ_BRICK_DIMS = [(2, 2, 1), (2, 1, 2), (1, 2, 2), (2, 2, 2), (3, 1, 2), (1, 3, 2), (2, 2, 3)]


def fuzz_brick_case(seed, tol=1e-6):
    """One random blob scene with a free surface, its parameters and a brick decomposition, all drawn from RandomState(seed)."""
    from polystokes_amd import scenes
    rng = np.random.RandomState(seed)
    dims = FUZZ_BRICK_DIMS[int(rng.randint(len(FUZZ_BRICK_DIMS)))]
    tile = int(rng.choice([8, 16, 16]))
    n = [16 * int(rng.randint(d, d + 2)) if d > 1 else int(rng.randint(16, 40)) for d in dims]
    n = [max(v, 16 * d) for v, d in zip(n, dims)]
    sc, p = scenes.blob(n[0], n[1], n[2], seed=seed, tile=tile, pad=int(rng.choice([1, 2])), variable_viscosity=bool(rng.randint(2)))
    p.preconditioner = int(rng.choice([abi.PRE_IDENTITY, abi.PRE_DIAGONAL, abi.PRE_CHEBYSHEV]))
    p.activeLiquidBoundaryLayerSize = int(rng.choice([1, 2, 3])); p.activeSolidBoundaryLayerSize = int(rng.choice([0, 1, 2]))
    p.tolerance = float(tol)
    p.maxSolverIterations = 20000
    return sc, p, dims, n, tile


def rigid_rotation_scene(n=24, w=(0.3, -0.2, 0.5)):
    """the droplet scene with u = w x (x - centre) sampled on its faces; returns (scene, params, the three face arrays flattened)"""
    sc, p = scenes.droplet(n)
    dx = sc.dx
    w = np.array(w)
    c = np.array([0.5, 0.5, 0.5])
    ref = []
    for a in range(3):
        shp = [n, n, n]
        shp[a] += 1
        i, j, k = np.meshgrid(np.arange(shp[0]), np.arange(shp[1]), np.arange(shp[2]), indexing="ij")      # (x, y, z) index order
        pos = [(i + (0.0 if a == 0 else 0.5)) * dx - c[0], (j + (0.0 if a == 1 else 0.5)) * dx - c[1], (k + (0.0 if a == 2 else 0.5)) * dx - c[2]]
        u = np.cross(w, np.stack(pos, axis=-1))[..., a]
        ua = np.ascontiguousarray(u.transpose(2, 1, 0)).astype(np.float32)                                 # stored z-major, x fastest
        assert ua.shape == np.asarray(sc.vel[a]).shape
        sc.vel[a][:] = ua
        ref.append(ua.ravel())
    return sc, p, ref
