"""The library the Houdini shim links — libpolystokes_hip_release.so (-DPS_RELEASE: no PS_* environment switch is read, the 38 switches of the
lab build collapse to their defaults at preprocessing time) — run ON THE GPU (it had only been string-checked on
the CPU box).  A wrong default would be invisible in the lab build; here the release binary must
  * reproduce the committed small goldens (tests/test_golden.py::test_hip_path_reproduces_golden, in a child pytest with PS_LIB set), and
  * give a bit-identical solution vector to the lab build on a Jacobi, a Chebyshev and an identity solve (and the same iteration count)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RELEASE = os.path.join(ROOT, "polystokes_amd", "libpolystokes_hip_release.so")

_SOLVES = (
    "import sys, numpy as np\n"
    f"sys.path.insert(0, {ROOT!r})\n"
    "import polystokes_amd\nfrom polystokes_amd import scenes, _abi as abi\n"
    "assert polystokes_amd.LIB_PATH.endswith(sys.argv[2]), polystokes_amd.LIB_PATH\n"
    "cases = {'cavity32_jacobi': scenes.cavity(32, precond=abi.PRE_DIAGONAL), 'coil48_chebyshev': scenes.coil(48), 'blob6_identity': scenes.blob(seed=6),\n"
    "         'cavity96_jacobi_4k': scenes.cavity(96, precond=abi.PRE_DIAGONAL)}   # 2.6 M rows: the four-kernel step engages by size\n"
    "cases['coil48_chebyshev'][1].preconditioner = abi.PRE_CHEBYSHEV\n"
    "out = {}\n"
    "s = polystokes_amd.Solver(0)\n"
    "for name, (sc, p) in cases.items():\n"
    "    p.tolerance = 1e-3 if name.endswith('_4k') else 1e-6\n"
    "    rc = s.step(sc, p)\n"
    "    assert int(s.array('fusedStep')[0]) == (1 if name.endswith('_4k') else 0), name\n"
    "    out[name + '_rc'] = rc; out[name + '_it'] = int(s.stats.solveData[1]); out[name + '_x'] = s.array('solutionVector')\n"
    "    out[name + '_vx'] = s.vel[0].copy(); out[name + '_labels'] = s.array('centerLabels')\n"
    "s.close()\n"
    "np.savez(sys.argv[1], **out)\n"
)


def _run(path, lib_env, suffix):
    env = dict(os.environ)
    env.pop("PS_LIB", None)
    env.update(lib_env)
    # switches of the lab build that would change results must not leak into the comparison (the release build would ignore them anyway)
    for k in list(env):
        if k.startswith("PS_") and k not in ("PS_LIB", "PS_VERBOSE"):
            env.pop(k)
    subprocess.run([sys.executable, "-c", _SOLVES, path, suffix], check=True, env=env, timeout=600)
    return np.load(path)


def test_release_library_is_bit_identical_to_the_lab_build(tmp_path):
    assert os.path.exists(RELEASE), "build it: make -C polystokes_amd/csrc"
    lab = _run(str(tmp_path / "lab.npz"), {}, "libpolystokes_hip.so")
    rel = _run(str(tmp_path / "rel.npz"), {"PS_LIB": RELEASE}, "libpolystokes_hip_release.so")
    assert set(lab.files) == set(rel.files)
    for k in lab.files:
        assert np.array_equal(lab[k], rel[k]), k
    for name in ("cavity32_jacobi", "coil48_chebyshev", "blob6_identity", "cavity96_jacobi_4k"):
        assert int(rel[name + "_rc"]) == 1 and int(rel[name + "_it"]) > 5, name


def test_release_library_reproduces_the_small_goldens():
    env = dict(os.environ, PS_LIB=RELEASE, PS_TEST_CHILD="1")
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_golden.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                         "-k", "test_hip_path_reproduces_golden"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert pr.returncode == 0, pr.stdout[-4000:]
    assert " passed" in pr.stdout and "failed" not in pr.stdout, pr.stdout[-2000:]
