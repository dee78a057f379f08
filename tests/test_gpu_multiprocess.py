"""The REAL multi-process distributed path on one GPU: one process per rank, each with its own ps_context holding a
slab, ps_step_device -> ps_dist_step_single, halos and scalars exchanged through the host-staged TCP transport
(ps_comm_init_tcp; RCCL refuses several ranks on one device).  Same Dist code as the RCCL path except transport()/
allreduce().  Compared with the single-domain solve of the whole scene."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from polystokes_amd import _abi as abi
from polystokes_amd import partition

import mp_cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port_base(n):
    """a base port with n consecutive free ports (checked by binding them once)"""
    for base in range(29600 + (os.getpid() % 500) * 8, 40000, 64):
        socks = []
        try:
            for q in range(n):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.bind(("127.0.0.1", base + q))
                socks.append(s)
            return base
        except OSError:
            continue
        finally:
            for s in socks:
                s.close()
    raise RuntimeError("no free port range")


def _run_ranks(case, world, tmp_path, env=None, per_process=1):
    """per_process > 1: every process holds that many ranks, one thread each (mp_rank.py) — the box admits six GPU processes"""
    base = _free_port_base(world)
    outs = [str(tmp_path / f"{case}.r{r}.npz") for r in range(world)]
    groups = [list(range(q, min(q + per_process, world))) for q in range(0, world, per_process)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_rank.py"), case, str(world), ",".join(str(r) for r in g), str(base), ",".join(outs[r] for r in g)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {}))) for g in groups]
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for q, pr in enumerate(procs):
        assert pr.returncode == 0, (groups[q], logs[q][-2000:])
    return [np.load(o) for o in outs]


@pytest.mark.parametrize("case", ["cavity_w2", "cavity_w3_jacobi", "coil_w2", "cavity_w2_bicgstab", "cavity_w2_chebyshev",
                                  "cavity_w2+fused", "cavity_w3_jacobi+fused", "coil_w2+fused", "cavity_w2_bicgstab+fused",
                                  "cavity64_b2x2x1", "cavity_b2x1x2+fused"])
def test_multiprocess_tcp_matches_single_domain(case, tmp_path):
    """"+fused": the four-kernel PCG step across the slabs (default from 1.2 M owned rows per rank), forced in the rank processes."""
    env = {"PS_FUSED_R": "1"} if case.endswith("+fused") else None
    case = case.replace("+fused", "")
    world = mp_cases.WORLD[case]
    res = _run_ranks(case, world, tmp_path, env)     # children first: the parent's own GPU context comes after
    if env:
        assert all(int(r["fused"]) == 1 for r in res)
    _compare_with_single_domain(case, world, res)


STUB_LIB = os.path.join(HERE, "stub_rccl", "libps_stub_rccl.so")


@pytest.mark.parametrize("case", ["cavity_w2+fused", "cavity_w3_jacobi+fused", "coil_w2+fused", "cavity64_b2x2x1+fused", "cavity_b2x1x2+fused",
                                  "cavity_w2", "cavity_w2_chebyshev", "cavity_w2_bicgstab+fused"])
def test_multiprocess_async_transport_matches_tcp_and_single_domain(case, tmp_path):
    """The ASYNCHRONOUS transport branch (Dist::transport / allreduce with useRccl: every send, receive and all-reduce enqueued on the
    comm stream, ordered against the solver stream by events only) with N > 1 ranks on ONE GPU: the rank processes load tests/stub_rccl
    in place of librccl (PS_RCCL_LIB) — messages are stream-ordered device copies through hipIpcMemHandle-mapped mailboxes, ordered
    across the processes by hipIpcEventHandle events, and nothing in it synchronises the host with the device.  (The TCP transport
    calls hipStreamSynchronize twice per exchange and the in-process group shares one stream: both hide a missing order().)
    Two processes as slabs, three as slabs, four as 2 x 2 x 1 and 2 x 1 x 2 bricks — the box admits 6 GPU processes, so the 2 x 2 x 2
    case stays with the in-process groups — overlapped four-kernel step and five-kernel step: the same result as the single domain
    and, bit for bit, as the run over TCP (same kernels, same reduction order)."""
    if not os.path.exists(STUB_LIB):
        pytest.fail("tests/stub_rccl/libps_stub_rccl.so is missing: __graft_entry__.build() compiles it")
    fused = case.endswith("+fused")
    case = case.replace("+fused", "")
    world = mp_cases.WORLD[case]
    env = {"PS_TEST_TRANSPORT": "stub", "PS_RCCL_LIB": STUB_LIB, "PS_DIST_OVERLAP": "1"}
    if fused:
        env["PS_FUSED_R"] = "1"
    res = _run_ranks(case, world, tmp_path, env)
    if fused:
        assert all(int(r["fused"]) == 1 for r in res)
        if "bicgstab" not in case:
            assert all(int(r["overlap"]) == 1 for r in res)        # the exchanges really ran under the rows that do not need them
    _compare_with_single_domain(case, world, res)
    (tmp_path / "tcp").mkdir()
    res_tcp = _run_ranks(case, world, tmp_path / "tcp", {"PS_FUSED_R": "1"} if fused else None)
    for r in range(world):
        assert int(res[r]["iters"]) == int(res_tcp[r]["iters"]) and int(res[r]["rc"]) == int(res_tcp[r]["rc"])
        for a in range(3):
            assert np.array_equal(res[r]["vel%d" % a], res_tcp[r]["vel%d" % a]), (case, r, a)


def test_eight_asynchronous_ranks_as_bricks(tmp_path):
    """2 x 2 x 2 bricks of a 32^3 cavity (tile 8) over the asynchronous transport: EIGHT ranks, every one with its own ps_context, solver stream and comm
    stream, exchanging with three face neighbours and three diagonal ones in one grouped round (Dist::valuesOut / contributionsBack) and
    all-reducing among eight — as four processes of two ranks (one thread each; tests/stub_rccl reaches a rank of the same process through
    its device pointer).  The overlapped four-kernel step, Jacobi-PCG; compared with the single domain.  (VERDICT r04 missing #3: eight
    asynchronous ranks had never run; the in-process group of eight shares one stream.)"""
    if not os.path.exists(STUB_LIB):
        pytest.fail("tests/stub_rccl/libps_stub_rccl.so is missing: __graft_entry__.build() compiles it")
    case, world = "cavity32_b2x2x2", 8           # tile 8, bricks of 16^3 owned cells (eight ranks share one GPU: the 64^3 case takes 50 s per run, this one 10)
    env = {"PS_TEST_TRANSPORT": "stub", "PS_RCCL_LIB": STUB_LIB, "PS_DIST_OVERLAP": "1", "PS_FUSED_R": "1"}
    res = _run_ranks(case, world, tmp_path, env, per_process=2)
    assert all(int(r["fused"]) == 1 and int(r["overlap"]) == 1 for r in res)
    _compare_with_single_domain(case, world, res)
    (tmp_path / "fwd").mkdir()                   # the three forwarding rounds of r03 / r04
    res_fwd = _run_ranks(case, world, tmp_path / "fwd", dict(env, PS_DIST_FORWARD="1"), per_process=2)
    _compare_with_single_domain(case, world, res_fwd)


def test_async_transport_exposes_a_missing_stream_order(tmp_path):
    """The point of the stand-in: with the event that makes the solver stream wait for the unpacked halo values removed
    (PS_DIST_SKIP_ORDER=1: Dist::order(c, 1, false)), the overlapped step over the asynchronous transport reads stale halo values and
    the solve goes wrong.  (Over TCP the host blocks in hipStreamSynchronize around every exchange, which can order the streams by accident.)"""
    case, world = "cavity_b2x1x2", 4
    base_env = {"PS_FUSED_R": "1", "PS_DIST_SKIP_ORDER": "1"}
    res = _run_ranks(case, world, tmp_path, dict(base_env, PS_TEST_TRANSPORT="stub", PS_RCCL_LIB=STUB_LIB))
    (tmp_path / "ok").mkdir()
    good = _run_ranks(case, world, tmp_path / "ok", {"PS_FUSED_R": "1", "PS_TEST_TRANSPORT": "stub", "PS_RCCL_LIB": STUB_LIB})
    differs = any(int(res[r]["rc"]) != int(good[r]["rc"]) or int(res[r]["iters"]) != int(good[r]["iters"]) or
                  (int(res[r]["rc"]) in (0, 1) and any(not np.array_equal(res[r]["vel%d" % a], good[r]["vel%d" % a]) for a in range(3))) for r in range(world))
    assert differs, "the solve without the stream ordering came out bit-identical: the transport hides the race"


def _compare_with_single_domain(case, world, res):
    import polystokes_amd
    sc, p = mp_cases.make(case)
    single = polystokes_amd.Solver(0)
    rc1 = single.step(sc, p)
    assert all(int(r["rc"]) == rc1 for r in res), [(int(r["rc"]), str(r["err"])) for r in res]
    it1 = single.stats.solveData[1]
    assert all(abs(int(r["iters"]) - it1) <= max(2, 0.02 * it1) for r in res), (it1, [int(r["iters"]) for r in res])
    assert all(int(r["used_bicgstab"]) == single.stats.usedBiCGStab for r in res)
    lab = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx)
    sh = abi.grid_shapes(sc.nx, sc.ny, sc.nz)
    vel = [np.array(sc.vel[a], copy=True) for a in range(3)]
    valid = [np.zeros(sh["face" + "XYZ"[a]], np.float32) for a in range(3)]
    dims = mp_cases.DIMS.get(case)
    for r in range(world):
        if dims is not None:            # bricks (ps_set_brick): the same checks on the owned box
            b = partition.make_brick((sc.nx, sc.ny, sc.nz), dims, r, p.tileSize)
            ll = res[r]["labels"].reshape(b.n_local[2], b.n_local[1], b.n_local[0])
            assert np.array_equal(ll[b.lo[2]:b.hi[2], b.lo[1]:b.hi[1], b.lo[0]:b.hi[0]], lab[b.g0[2]:b.g1[2], b.g0[1]:b.g1[1], b.g0[0]:b.g1[0]]), (case, r)
            for a in range(3):
                shp = (b.n_local[2] + (a == 2), b.n_local[1] + (a == 1), b.n_local[0] + (a == 0))
                partition.merge_faces_brick(vel[a], res[r]["vel%d" % a].reshape(shp), res[r]["owned%d" % a], b, a)
                partition.merge_faces_brick(valid[a], res[r]["valid%d" % a].reshape(shp), res[r]["owned%d" % a], b, a)
            continue
        sl = partition.make_slab(sc.nz, world, r, p.tileSize)
        ll = res[r]["labels"].reshape(sl.nz_local, sc.ny, sc.nx)
        assert np.array_equal(ll[sl.zLoOwned:sl.zHiOwned], lab[sl.z0:sl.z1]), (case, r)
        for a in range(3):
            partition.merge_faces(vel[a], res[r]["vel%d" % a], res[r]["owned%d" % a], sl, a)
            partition.merge_faces(valid[a], res[r]["valid%d" % a], res[r]["owned%d" % a], sl, a)
    for a in range(3):
        assert np.array_equal(valid[a], single.valid[a]), case
        if rc1 == abi.SUCCESS:
            scale = max(np.abs(single.vel[a]).max(), 1e-30)
            assert np.abs(vel[a] - single.vel[a]).max() <= 20 * p.tolerance * scale, case
    single.close()


def test_multiprocess_interrupt_stops_every_rank(tmp_path):
    """An interrupt callback on ONE rank: all ranks return PS_INCOMPLETE after the same batch, nobody hangs in a receive."""
    res = _run_ranks("cavity_w2_interrupt", 2, tmp_path)
    assert [int(r["rc"]) for r in res] == [abi.INCOMPLETE, abi.INCOMPLETE]
    assert int(res[0]["iters"]) == int(res[1]["iters"]) == 25


def test_multiprocess_list_mismatch_fails_every_rank(tmp_path):
    """Ranks that were handed different fields for the same cells (here: rank 1 sees air in a patch of its own first layers, rank 0's
    halo copy does not): rank 0 finds the owner's labels differing from its copy deeper inside the halo block than the classification
    reaches (Dist::exchangeLabels), and EVERY rank returns FAILED instead of solving on inconsistent data or waiting for ever.
    (Until r04 the halo labels were the rank's own and the same case was caught one step later, by the counts / key hashes of the
    exchange lists — that check is still there: `checkLists`.)"""
    res = _run_ranks("cavity_w2_failrank", 2, tmp_path)
    assert [int(r["rc"]) for r in res] == [-1, -1], [str(r["err"]) for r in res]
    assert all("labels of a halo block" in str(r["err"]) for r in res), [str(r["err"]) for r in res]
    assert "different fields" in str(res[0]["err"])


def test_bench_two_ranks_produces_one_line_whatever_the_transport(tmp_path):
    """`python bench.py --gpus 2` without a launcher: spawns one process per rank, rendezvous over gloo, and prints ONE JSON
    line.  With two GPUs the ranks talk over RCCL; on a one-GPU box RCCL refuses the duplicate device, every rank agrees on that
    (gloo) before any further collective, and all of them switch to the host-staged transport — the line says which."""
    import json
    import polystokes_amd
    try:                                   # a second device?  (probed through the library: no torch import in this process)
        polystokes_amd.Solver(1).close()
        two_gpus = True
    except polystokes_amd.PolyStokesError:
        two_gpus = False
    env = dict(os.environ, MASTER_PORT=str(_free_port_base(1)))
    pr = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--res", "32", "--steps", "1", "--warmup", "1",
                         "--no-cpu-baseline", "--strong-res", "96"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420, env=env)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["unit"] == "ms/step" and d["value"] > 0 and d["cg_iterations"] > 0
    assert d["config"]["grid"] == [32, 32, 64] and d["scaling"] == "weak"
    if two_gpus:
        assert d["transport"] == "rccl"
    else:
        assert d["transport"].startswith("tcp (FALLBACK")
    # the strong-scaling block of the same invocation (BASELINE config 4 at --strong-res: here the 96^3 coil in two slabs)
    sb = d["strong_512"]
    assert sb["n_gpus"] == 2 and sb["scaling"] == "strong" and sb["decomposition"] == "1x1x2 bricks" and sb["ms_per_step"] > 0 and sb["result"] == 1
    single = polystokes_amd.Solver(0)
    from polystokes_amd import scenes
    sc, p = scenes.coil(96)
    p.preconditioner = 5          # abi.PRE_DIAGONAL: bench.py's default --precond jacobi
    single.step(sc, p)
    assert abs(sb["cg_iterations"] - single.stats.solveData[1]) <= max(2, 0.02 * single.stats.solveData[1])
    single.close()


def test_bench_four_ranks_as_bricks(tmp_path):
    """`bench.py --gpus 4 --bricks 2x2x1`: the decomposition along x and y (ps_set_brick), weak scaling = 32^3 owned cells per rank of the
    64 x 64 x 32 cavity; on a one-GPU box over the host-staged transport."""
    import json
    env = dict(os.environ, MASTER_PORT=str(_free_port_base(1)))
    pr = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "4", "--bricks", "2x2x1", "--res", "32", "--steps", "1",
                         "--warmup", "1", "--no-cpu-baseline", "--no-strong-512", "--transport", "tcp"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420, env=env)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["value"] > 0 and d["cg_iterations"] > 0
    assert d["config"]["grid"] == [64, 64, 32] and "2x2x1 bricks" in d["config"]["parallelism"]
    # the same system as one rank solves it: the iteration count of the 64 x 64 x 32 cavity
    import polystokes_amd
    from polystokes_amd import scenes, partition
    sc = abi.Scene(64, 64, 32, 1.0 / 32, 1.0e-2, 1.0, [np.pad(np.zeros((31, 64, 65), np.float32), ((0, 1), (0, 0), (0, 0)), constant_values=1.0), 0.0, 0.0], -1.0, 1.0, 1.0)
    _, p = scenes.cavity(32, tile=16, pad=2, precond=abi.PRE_DIAGONAL)
    single = polystokes_amd.Solver(0)
    assert single.step(sc, p) == abi.SUCCESS
    assert abs(single.stats.solveData[1] - d["cg_iterations"]) <= max(2, 0.02 * single.stats.solveData[1])
    single.close()


def test_rccl_communicator_next_to_a_live_torch_context():
    """bench.py's situation for N > 1: torch imported and its GPU context live, then the library's RCCL communicator.  The library
    must pick up the librccl torch has already mapped (RTLD_NOLOAD) — two copies in one process abort at exit — and a slab step over
    the communicator (world 1: the exchange lists are empty, the all-reduces real) must run and exit cleanly."""
    pr = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "scripts", "rccl_with_torch.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                        text=True, timeout=300)
    assert pr.returncode == 0, pr.stdout[-3000:]
    after = [l for l in pr.stdout.splitlines() if l.startswith("rccl mapped after:")]
    assert after and after[0].count("librccl") == 1, pr.stdout[-2000:]
    assert " rc 1 " in after[0] and "closed" in pr.stdout


def test_bench_single_gpu_line_keeps_the_contract():
    """`python bench.py` (N = 1) on a small stand-in (64^3 cavity, 64^3 coil for the strong block, a 32^3 CPU sample): ONE JSON line with the
    driver's keys, the roofline and cpu_baseline objects, and r05's additions — the identity / Chebyshev lines beside the Jacobi headline
     and the box calibration."""
    import json
    pr = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--res", "64", "--steps", "2", "--warmup", "1", "--strong-res", "64",
                         "--cpu-sample-res", "32"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is False and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["unit"] == "ms/step" and d["value"] == d["ms_per_step"] > 0
    assert "jacobi-PCG" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c and c["unit"].startswith("ms/step")
    o = d["other_preconditioners"]
    assert set(o) == {"identity", "chebyshev4", "chebyshev4_fp64", "chebyshev10"}
    assert o["chebyshev10"]["cg_iterations"] < o["chebyshev4"]["cg_iterations"]
    assert o["chebyshev4_fp64"]["inner_vectors"] == "fp64" and o["chebyshev4"]["inner_vectors"] in ("fp32", "fp64")   # (fp32 where its kernels run: DESIGN.md)
    assert abs(o["chebyshev4"]["cg_iterations"] - o["chebyshev4_fp64"]["cg_iterations"]) <= max(2, 0.05 * o["chebyshev4_fp64"]["cg_iterations"])
    for v in o.values():
        assert v["result"] == 1 and v["ms_per_step"] > 0 and v["cg_iterations"] > 0
    assert o["chebyshev4"]["cg_iterations"] < d["cg_iterations"]      # the polynomial cuts the count
    assert d["box"]["d2d_copy_GBps"] > 500.0
    assert d["strong_512"]["n_gpus"] == 1 and d["strong_512"]["result"] == 1
