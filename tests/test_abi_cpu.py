"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol the public
header declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "polystokes.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(ps_[a-z_]+|polystokes_step)\s*\(", txt)
    return sorted(set(n for n in names if n not in ("ps_allreduce_fn", "ps_halo_fn")))


def test_library_exports_every_declared_symbol():
    import polystokes_amd
    L = polystokes_amd.lib()
    syms = _header_symbols()
    assert set(syms) == set(polystokes_amd.EXPORTED_SYMBOLS)
    for s in syms:
        assert hasattr(L, s), s
    assert L.ps_abi_version() == 1


def test_release_build_reads_no_environment_switch():
    """the library the Houdini shim links (-DPS_RELEASE) exports the same ABI and holds none of the lab build's
    PS_* switch names (PS_ENV in ps_common.hpp is a null pointer at preprocessing time there) — PS_VERBOSE, which only prints, stays."""
    import polystokes_amd
    rel = os.path.join(ROOT, "polystokes_amd", "libpolystokes_hip_release.so")
    assert os.path.exists(rel), "build it: make -C polystokes_amd/csrc"
    blob = open(rel, "rb").read()
    names = set(m.decode() for m in re.findall(rb"PS_[A-Z][A-Z0-9_]{2,}", blob))
    assert names <= {"PS_VERBOSE"}, names
    lab = set(m.decode() for m in re.findall(rb"PS_[A-Z][A-Z0-9_]{2,}", open(polystokes_amd.LIB_PATH, "rb").read()))
    assert {"PS_DIST_SKIP_ORDER", "PS_RCCL_LIB", "PS_DEBUG_POISON"} <= lab          # the lab build does read them (and says so on stderr)
    L = ctypes.CDLL(rel)
    for s in polystokes_amd.EXPORTED_SYMBOLS:
        assert hasattr(L, s), s
    L.ps_abi_version.restype = ctypes.c_int32
    assert L.ps_abi_version() == 1
    shim = open(os.path.join(ROOT, "shim", "CMakeLists.txt")).read()
    assert "libpolystokes_hip_release.so" in shim


def test_struct_layout_matches_header():
    import polystokes_amd
    from polystokes_amd import _abi as abi
    p = abi.Params()
    polystokes_amd.lib().ps_params_default(ctypes.byref(p))
    d = abi.default_params()
    for name, _ in abi.Params._fields_:
        assert getattr(p, name) == getattr(d, name), name
    assert p.tolerance == 1e-3 and p.maxSolverIterations == 5000 and p.tileSize == 16 and p.tilePadding == 2


def test_no_gpu_fails_loudly():
    """Without a HIP device the product refuses to construct a solver (there is no CPU fallback).  The device is probed through
    the library itself, not through torch: importing torch after the library has dlopen'ed ROCm's librccl would bring a second
    RCCL into the process (two copies abort at interpreter exit)."""
    import polystokes_amd
    try:
        s = polystokes_amd.Solver(0)
    except polystokes_amd.PolyStokesError as e:
        assert "no HIP device" in str(e) or "no CPU fallback" in str(e)
        return
    s.close()
    pytest.skip("GPU present")


def test_traversal_order_roundtrip_matches_oracle(oracle_mod):
    # host restatement of ps_common.hpp::ijkToOrder against the oracle's numbering on a ragged grid
    from polystokes_amd import scenes
    sc, p = scenes.cavity(20)
    p.doReducedRegions = 0
    sc2 = type(sc)(37, 18, 21, sc.dx, sc.dt, 1.0, [0, 0, 0], -1.0, 1.0, 1.0)
    o = oracle_mod.Oracle()
    o.run(sc2, p, solve=False)
    idx = o.array("centerActiveIndices").reshape(21, 18, 37)

    def order(i, j, k, d=(37, 18, 21), T=16):
        tz, ty, tx = k // T, j // T, i // T
        hz = T if tz * T + T <= d[2] else d[2] - tz * T
        hy = T if ty * T + T <= d[1] else d[1] - ty * T
        wx = T if tx * T + T <= d[0] else d[0] - tx * T
        return (d[0] * d[1] * T * tz + d[0] * T * hz * ty + T * hy * hz * tx + (i - tx * T)
                + wx * ((j - ty * T) + hy * (k - tz * T)))
    rng = np.random.RandomState(0)
    for _ in range(500):
        i, j, k = rng.randint(37), rng.randint(18), rng.randint(21)
        assert idx[k, j, i] == order(i, j, k)


def test_shim_parameter_table_covers_the_params_struct():
    """shim/HDK_PolyStokes_shim.C (the Houdini DSO source, built only where $HFS exists): every node parameter of ps_params has
    a row in its template table with the header's default, and the DOP names of the reference are there."""
    src = open(os.path.join(ROOT, "shim", "HDK_PolyStokes_shim.C")).read()
    rows = dict()
    for m in re.finditer(r"\{'([SFITO])',\s*(\"[^\"]+\"|[A-Z_]+),\s*\"[^\"]*\",\s*(nullptr|\"(?:[^\"\\]|\\.)*\"),\s*([-0-9.e]+)\}", src):
        rows[m.group(2).strip('"')] = (m.group(1), float(m.group(4)))
    hdr = open(os.path.join(ROOT, "include", "polystokes.h")).read()
    body = hdr[hdr.index("typedef struct ps_params {"):hdr.index("} ps_params;")]
    body = body[:body.index("extensions")]
    members = re.findall(r"(?:double|int32_t)\s+(\w+);\s*/\*\s*([-0-9.e]+)?", body)
    assert len(members) >= 20
    alias = {"tolerance": "SIM_NAME_TOLERANCE"}
    for name, default in members:
        key = alias.get(name, name)
        assert key in rows, name
        if default and name not in ("matrixSetup", "solverType"):   # the menus keep the reference's off-by-one template ordinal
            assert rows[key][1] == float(default), (name, rows[key], default)
    for needle in ('"hdk_polystokes"', '"HDK Polynomial Stokes Solver"', "initializeSIM", "IMPLEMENT_DATAFACTORY(HDK_PolyStokes)",
                   "polystokes_step(", "ps_set_interrupt("):
        assert needle in src, needle
    assert '"HDK PolyStokes Solver"' in open(os.path.join(ROOT, "shim", "HDK_PolyStokes_shim.h")).read()


def test_documented_environment_switches_exist_in_the_sources():
    """README's table of PS_* switches against the sources: a switch that is documented must be read somewhere (library, Python
    harness, oracle loader), and every switch the library reads must be documented."""
    import glob
    readme = open(os.path.join(ROOT, "README.md")).read()
    documented = set(re.findall(r"`(PS_[A-Z0-9_]+)", readme))
    src = ""
    for pat in ("polystokes_amd/csrc/*.h*", "polystokes_amd/*.py", "oracle/*.py", "bench.py", "__graft_entry__.py"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            src += open(f).read()
    read_in_lib = set(re.findall(r"getenv\(\"(PS_[A-Z0-9_]+)\"\)", src)) | set(re.findall(r"environ(?:\.get)?[\[(]\"(PS_[A-Z0-9_]+)\"", src))
    compile_time = {"PS_CHEB_INTERVAL_RATIO"}                      # a header constant, not an environment variable
    missing_in_src = {d for d in documented if d not in src} - compile_time
    assert not missing_in_src, "documented but not in the sources: %s" % sorted(missing_in_src)
    undocumented = read_in_lib - documented
    assert not undocumented, "read by the sources but not in README.md: %s" % sorted(undocumented)
