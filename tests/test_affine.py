"""AFFINE_REGIONS (11-DOF reduced model): the reference's other compile-time variant (lib/include/units.h:9-18,
exec/HDK_PolyStokesSolver.cpp:2153-2184).  Both the library and the oracle are built a second time with -DPS_AFFINE_REGIONS
(libpolystokes_hip_affine.so, libps_oracle_affine.so); a process can hold one variant, so the checks run in a child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, PS_LIB=os.path.join(ROOT, "polystokes_amd", "libpolystokes_hip_affine.so"),
           PS_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "libps_oracle_affine.so"))


def _child(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "affine_child.py"), *args], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


def test_affine_libraries_are_built_and_export_the_abi():
    import ctypes
    import polystokes_amd
    polystokes_amd.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    L = ctypes.CDLL(ENV["PS_LIB"])
    for s in polystokes_amd.EXPORTED_SYMBOLS:
        assert hasattr(L, s), s
    L.ps_reduced_dof.restype = ctypes.c_int32
    assert L.ps_reduced_dof() == 11 and polystokes_amd.lib().ps_reduced_dof() == 26


def test_affine_oracle_known_answers():
    assert "affine cpu ok" in _child("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cavity24", "blob", "spheres32"])
def test_affine_hip_path_matches_affine_oracle(scene):
    assert "affine gpu ok" in _child("gpu", scene)
