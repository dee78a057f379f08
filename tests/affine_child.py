#!/usr/bin/env python3
"""Child process of tests/test_affine.py: runs with PS_LIB / PS_ORACLE_LIB pointing at the AFFINE_REGIONS (11-DOF) builds
(the reduced model is a compile-time choice, as in the reference: lib/include/units.h:9-18).
    affine_child.py cpu           -> oracle-only known answers
    affine_child.py gpu <scene>   -> HIP path against the oracle"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ps_oracle  # noqa: E402
from polystokes_amd import _abi as abi, scenes  # noqa: E402

RD = 11


def scene(name):
    if name == "cavity24":
        return scenes.cavity(24, tile=12)
    if name == "blob":
        return scenes.blob(seed=4)
    return scenes.spheres(32, tile=8)


def cpu():
    assert ps_oracle.reduced_dof() == RD
    rng = np.random.RandomState(1)
    h = 0.37
    for _ in range(20):          # the affine basis is discretely divergence-free too (Solver.cpp:2160-2181)
        c, x = rng.randn(RD), rng.randn(3)
        div = 0.0
        for a in range(3):
            e = np.zeros(3); e[a] = 0.5 * h
            div += (ps_oracle.basis(x + e, a) @ c - ps_oracle.basis(x - e, a) @ c) / h
        assert abs(div) < 1e-12 * (1 + np.abs(c).max() * 10)
    x, y, z = 0.3, -0.7, 1.1
    assert np.array_equal(ps_oracle.basis([x, y, z], 0), [1, 0, 0, x, y, z, 0, 0, 0, 0, 0])
    assert np.array_equal(ps_oracle.basis([x, y, z], 1), [0, 1, 0, 0, 0, 0, x, y, z, 0, 0])
    assert np.array_equal(ps_oracle.basis([x, y, z], 2), [0, 0, 1, -z, 0, 0, 0, -z, 0, x, y])
    sc, p = scenes.cavity(32)
    o = ps_oracle.Oracle(); o.run(sc, p)
    assert o.result == abi.SUCCESS and o.stats.dimData[11] == RD * o.nRegions
    Mr = o.array("reducedMassMatrices").reshape(-1, RD, RD)
    for a in range(3):
        assert Mr[0, a, a] == 14 * 14 * 15       # rho s^2 (s+1), s = 14
    Bi, K = o.array("Inv_Mr_plus_2JDtuDJ").reshape(-1, RD, RD), o.array("reducedViscosityMatrices").reshape(-1, RD, RD)
    assert np.abs(Bi[0] @ (Mr[0] / sc.dt + 2 * K[0]) - np.eye(RD)).max() < 1e-8
    xr = np.random.RandomState(2).standard_normal(o.nP + o.nT)
    yr = np.random.RandomState(3).standard_normal(o.nP + o.nT)
    assert abs(xr @ o.apply(yr) - yr @ o.apply(xr)) <= 1e-9 * abs(xr @ o.apply(yr)) and xr @ o.apply(xr) < 0
    print("affine cpu ok")


def gpu(name):
    import polystokes_amd
    assert polystokes_amd.lib().ps_reduced_dof() == RD and ps_oracle.reduced_dof() == RD
    sc, p = scene(name)
    p.tolerance = 1e-4 if name == "spheres32" else 1e-6   # spheres (mu = 1e4) is ill-conditioned: x agrees to ~cond * tol
    o = ps_oracle.Oracle(); o.run(sc, p)
    g = polystokes_amd.Solver(0)
    rc = g.step(sc, p)
    assert rc == o.result == abi.SUCCESS
    assert list(g.stats.dimData) == list(o.stats.dimData) and g.stats.dimData[11] == RD * o.nRegions
    for s in abi.SAMPLE_NAMES:
        for kind in ("Labels", "ActiveIndices", "ReducedIndices"):
            assert np.array_equal(g.array(s + kind), o.array(s + kind)), s + kind
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    if o.nRegions:
        assert np.array_equal(g.array("reducedRegionCOM"), o.array("reducedRegionCOM"))
        assert rel(g.array("reducedMassMatrices"), o.array("reducedMassMatrices")) < 1e-12
        assert rel(g.array("reducedViscosityMatrices"), o.array("reducedViscosityMatrices")) < 1e-12
        assert rel(g.array("reducedRHSVector"), o.array("reducedRHSVector")) < 1e-8
    assert rel(g.array("b"), o.array("b")) < 1e-9
    x = np.random.RandomState(5).standard_normal(g.nP + g.nT)
    assert rel(g.apply(x), o.apply(x)) < 1e-10
    ito, itg = int(o.stats.solveData[1]), int(g.stats.solveData[1])
    assert abs(ito - itg) <= max(2, 0.02 * ito), (ito, itg)
    xo, xg = o.array("solutionVector"), g.array("solutionVector")
    assert np.linalg.norm(xg - xo) <= 10 * p.tolerance * np.linalg.norm(xo)
    for a in range(3):
        assert np.array_equal(g.valid[a].ravel(), o.array("valid" + "XYZ"[a]))
    print("affine gpu ok", name, "iterations", itg, "regions", o.nRegions)
    g.close()


if __name__ == "__main__":
    cpu() if sys.argv[1] == "cpu" else gpu(sys.argv[2])
