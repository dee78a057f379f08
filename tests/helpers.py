import numpy as np
import scipy.sparse as sp

from polystokes_amd import _abi as abi


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
