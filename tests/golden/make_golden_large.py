"""Larger golden fixtures in DIGEST form (tests/golden/large_*.npz), generated from the CPU oracle like make_golden.py:
the scenes of BASELINE configs 2/4 (coil) and 5 (spheres) at 96^3, and the reference's only shipped parameter set
(scenes/jelly_jam: tileSize 32, tilePadding 3, layer sizes 3/3) on a 64^3 cavity and on an irregular blob.
Per case: SHA-256 of every integer / label / index / valid array (bit-exact state), dimData, iteration count, solve error,
every 5th entry of the solution x in float32 (the comparison is at 10*tol), and float64 checksums of b, x and of the operator
applied to a seeded vector w: norms and projections on w (||b||, b.w, ||x||, x.w, ||A w||, w.A w).  Output velocities are not part of THAT
digest: on the coil they difference 1e5-sized terms (AMP) and are decided by rounding at tol 1e-3.
They have their own fixture, large_vel_*.npz: the same scenes solved to tol 1e-8 (where the amplification no longer matters), every
5th entry of the three output velocity fields in float32 + the iteration count (build_velocity; the GPU test compares at 1e-4
of the largest velocity).  The tight solves are not repeated by the CPU suite (minutes of oracle time): the fixture is what
the oracle produced when this script last ran.

    python tests/golden/make_golden_large.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from polystokes_amd import _abi as abi  # noqa: E402
from polystokes_amd import scenes  # noqa: E402


def _shipped(sp):
    sc, p = sp
    p.activeLiquidBoundaryLayerSize = p.activeSolidBoundaryLayerSize = 3
    return sc, p


CASES = {
    "coil96": lambda: scenes.coil(96),
    "spheres96": lambda: scenes.spheres(96),
    "cavity64_t32p3_L3S3": lambda: _shipped(scenes.cavity(64, tile=32, pad=3)),
    "blob_t32p3_L3S3": lambda: _shipped(scenes.blob(52, 44, 48, seed=3, tile=32, pad=3)),
}
# Oracle-pinned fixtures at real sizes on the HEADLINE parameter set: the size-switched paths of the library (the
# four-kernel PCG step from 2 M rows, the non-temporal cache policy from 4 M rows) engage here by size, unforced.  Same digest form;
# consumed by the GPU test only (the oracle needs minutes per case: the CPU suite does not re-run them — `make_golden_large.py huge`).
HUGE = {
    "cavity128_t16p2_jacobi": lambda: scenes.cavity(128, tile=16, pad=2, precond=abi.PRE_DIAGONAL),      # 5.9 M system DOFs, BASELINE config 3 (Jacobi-PCG) at half resolution
    "coil128": lambda: scenes.coil(128, tile=16, pad=2),                        # BASELINE config 2 at its stated size
    "spheres128": lambda: scenes.spheres(128, tile=16, pad=2),                  # BASELINE config 5's geometry (moving solids, mu = 1e4) at 128^3
}
INT_ARRAYS = [s + k for s in abi.SAMPLE_NAMES for k in ("Labels", "ActiveIndices", "ReducedIndices")] + ["validX", "validY", "validZ"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def probe_vector(n):
    return np.random.RandomState(20261004).standard_normal(n)


def digest(get, stats, apply):
    """get(name) -> array, stats -> ps_stats-like, apply(w) -> A w: the same digest from the oracle or from the HIP path"""
    out = {"dimData": np.array(stats.dimData), "iterations": np.int32(stats.solveData[1]), "solveError": np.float64(stats.solveData[0])}
    for nm in INT_ARRAYS:
        a = get(nm)
        out["sha_" + nm] = np.array(sha(a.astype(np.int32) if nm.startswith("valid") else a))
    for s in abi.SAMPLE_NAMES:
        out["sha_" + s + "Weights8"] = np.array(sha(np.round(np.concatenate([get(s + "LiquidWeights"), get(s + "FluidWeights")]) * 8).astype(np.int8)))
    b, x = get("b"), get("solutionVector")
    w = probe_vector(b.size)
    out["x32_stride5"] = x[::5].astype(np.float32)
    out["b_norm"], out["b_dot_w"] = np.float64(np.linalg.norm(b)), np.float64(b @ w)
    out["x_norm"], out["x_dot_w"] = np.float64(np.linalg.norm(x)), np.float64(x @ w)
    y = apply(w)
    out["Aw_norm"], out["wAw"] = np.float64(np.linalg.norm(y)), np.float64(w @ y)
    return out


def build(name, exact_diagonal=False):
    """exact_diagonal: the oracle's Jacobi on 1 / A_jj in fp64 (ps_oracle.Oracle.set_exact_diagonal) instead of the product's 16-bit
    storage form — the fixture large_<name>_exactdiag.npz pins 'Jacobi-PCG on the stored diagonal == Jacobi-PCG' (VERDICT r05 item 2)."""
    from oracle import ps_oracle
    sc, p = (CASES.get(name) or HUGE[name])()
    o = ps_oracle.Oracle()
    if exact_diagonal:
        o.set_exact_diagonal(True)
    rc = o.run(sc, p)
    d = digest(o.array, o.stats, o.apply)
    d["result"] = np.int32(rc)
    return d


VEL_TOL = 1e-8


def velocity_digest(get, stats):
    out = {"iterations": np.int32(stats.solveData[1]), "solveError": np.float64(stats.solveData[0])}
    for a in "XYZ":
        v = np.asarray(get("vel" + a)).ravel()
        out["vel" + a + "_stride5"] = v[::5].astype(np.float32)
        out["vel" + a + "_max"] = np.float64(np.abs(v).max())
    return out


def tight(sp):
    sc, p = sp
    p.tolerance = VEL_TOL
    p.maxSolverIterations = 100000
    return sc, p


def build_velocity(name):
    from oracle import ps_oracle
    sc, p = tight(CASES[name]())
    o = ps_oracle.Oracle()
    rc = o.run(sc, p)
    d = velocity_digest(o.array, o.stats)
    d["result"] = np.int32(rc)
    return d


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "vel":     # python tests/golden/make_golden_large.py vel [case]
        here = os.path.dirname(os.path.abspath(__file__))
        for name in (sys.argv[2:] or list(CASES)):
            np.savez_compressed(os.path.join(here, "large_vel_" + name + ".npz"), **build_velocity(name))
            print("wrote velocities of", name)
        sys.exit(0)
    here = os.path.dirname(os.path.abspath(__file__))
    if len(sys.argv) > 1 and sys.argv[1] == "huge":    # python tests/golden/make_golden_large.py huge [case]
        import time
        for name in (sys.argv[2:] or list(HUGE)):
            t0 = time.time()
            np.savez_compressed(os.path.join(here, "large_" + name + ".npz"), **build(name))
            print("wrote", name, "in %.0f s" % (time.time() - t0), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "exactdiag":    # python tests/golden/make_golden_large.py exactdiag [case]: the Jacobi cases once more on the fp64 diagonal
        import time
        for name in (sys.argv[2:] or [n for n in HUGE if n.endswith("_jacobi")]):
            t0 = time.time()
            d = build(name, exact_diagonal=True)
            keep = {k: d[k] for k in ("result", "iterations", "solveError", "x32_stride5", "x_norm", "x_dot_w", "dimData")}
            np.savez_compressed(os.path.join(here, "large_" + name + "_exactdiag.npz"), **keep)
            print("wrote", name, "(exact diagonal) in %.0f s, %d iterations" % (time.time() - t0, int(d["iterations"])), flush=True)
        sys.exit(0)
    for name in CASES:
        np.savez_compressed(os.path.join(here, "large_" + name + ".npz"), **build(name))
        print("wrote", name)
