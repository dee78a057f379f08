#!/usr/bin/env python3
"""bench.py — Stokes-solve wall ms/step (assembly + PCG) on the 256^3 reduced-Stokes grid.

One "step" = ps_step_device(): the whole hot path (weights -> classification -> tile blocks -> CSR
assembly -> PCG -> velocity recovery/write-back) on inputs already resident in HBM.
Prints ONE JSON line (rank 0).  See DESIGN.md §measurement for the definitions used here.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves


def _cpu_affinity():
    """CPUs this process may run on: the scheduler affinity, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) / 1048576.0
                break
        else:
            return 0.0
        for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                v = open(path).read().strip()
                if v.isdigit():
                    avail = min(avail, int(v) / 2.0 ** 30)
            except OSError:
                pass
        return avail
    except OSError:
        return 0.0


def cpu_baseline(n_gpu_cells, gpu_n, gpu_iters, params_kw, sample_res=0, second_res=96, bench_res=256):
    """CPU restatement (oracle, kind "port") timed on this host's cores.  r06: MEASURED AT THE BENCHMARK SIZE — the oracle builds the 256^3 system
    with its setup sweeps threaded (ps_oracle.Oracle.set_setup_threads: bit-identical to its serial setup, tests/test_oracle_kat.py; the reference's own
    setup fans out over all cores, exec/HDK_PolyStokesSolver.cpp:154) and the CG iteration (pcg.h:311-335 around ApplyPressureStressMatrix.h:102-179) is
    timed THERE — no extrapolation.  (sample_res > 0, or a host with < 100 GB available / < 8 CPUs: the bounded sample of r01-r05 at that resolution,
    scaled by the DOF ratio and labelled `extrapolated`.)
      A  "reference-shaped": the reference's own pass structure — the three bodies of applyMatrixVectorProducts under
         `omp parallel sections` (so 3 threads at most, McInv*G and McInv*Dt re-formed on every call), Eigen-style
         single-thread vector updates (baseline A);
      B  "fair": one pass per block, every row loop split over OpenMP threads — at min(affinity, 16 / 32 / 64) threads.
    value = the SOLVE stage of one step: the best variant's time per iteration x the GPU run's iteration count.  The restatement's setup
    (threaded, wall time) is reported apart, not in value."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # before libgomp starts: spinning workers starve a shared host
    from oracle import ps_oracle
    from polystokes_amd import scenes
    from polystokes_amd import _abi as abi
    affinity = _cpu_affinity()
    thread_counts = sorted({min(affinity, t) for t in (16, 32, 64)})
    setup_threads = min(affinity, 64)
    mem_gb = _mem_available_gb()
    at_size = sample_res <= 0 and mem_gb >= 100.0 and affinity >= 8
    if sample_res <= 0:
        sample_res = bench_res if at_size else 128
    kw = dict(params_kw)
    kw["precond"] = abi.PRE_IDENTITY      # (the timed iteration applies no preconditioner: building the Jacobi diagonal would only lengthen the setup)

    def sample(ns, full, iters_b, iters_a):
        sc, p = scenes.cavity(ns, **kw)
        o = ps_oracle.Oracle()
        o.set_setup_threads(setup_threads)
        t0 = time.time()
        o.run(sc, p, solve=False)
        setup_ms = (time.time() - t0) * 1e3
        n_s = o.nP + o.nT
        by = {}
        for t in thread_counts:
            ms, used = o.time_cg_mt(iters_b, t)
            by[str(used)] = ms
        out = {"res": ns, "dofs": n_s, "setup_ms": setup_ms, "B_ms_per_cg_iter_by_threads": by}
        if full:
            ms_a, used_a = o.time_cg_sections(iters_a)
            out["A"] = (ms_a, used_a)
        del o
        return out

    big = sample(sample_res, True, 10, 3 if sample_res >= 192 else 10)
    small = sample(second_res, False, 10, 0) if second_res and second_res != sample_res else None
    ns, n_s, setup_ms = big["res"], big["dofs"], big["setup_ms"]
    ms_it_a, used_a = big["A"]
    by = big["B_ms_per_cg_iter_by_threads"]
    best_threads = min(by, key=lambda k: by[k])
    ms_it_b, used_b = by[best_threads], int(best_threads)
    scale_cells = n_gpu_cells / float(ns ** 3)
    dof_ratio = gpu_n / float(n_s)
    measured_at_size = abs(dof_ratio - 1.0) < 1e-9
    solve = lambda ms_it: ms_it * dof_ratio * max(gpu_iters, 1)
    best_it = min(ms_it_b, ms_it_a)
    lin = None
    if small is not None:
        per = lambda smp: {k: v * 1e6 / smp["dofs"] for k, v in smp["B_ms_per_cg_iter_by_threads"].items()}   # ns per DOF-iteration
        lin = {"ns_per_dof_iteration_B": {"%d^3" % small["res"]: per(small), "%d^3" % ns: per(big)},
               "dofs": {"%d^3" % small["res"]: small["dofs"], "%d^3" % ns: n_s},
               "ratio_large_over_small_at_best_threads": per(big)[best_threads] / per(small)[best_threads] if per(small).get(best_threads) else None,
               "note": "time per DOF-iteration of baseline B at a small sample and at the measured size: what an extrapolation from the small sample would have missed"}
    out = {
        # value = the SOLVE of one step only (the iteration count of the GPU run x the best measured CPU iteration time).  Compare with the GPU line's stage_ms.solve.
        "value": solve(best_it), "unit": "ms/step (solve stage only)", "cores": used_b if ms_it_b <= ms_it_a else used_a, "kind": "port",
        "solve_ms_per_step": solve(best_it), "setup_ms_per_step": setup_ms * scale_cells, "setup_threads": setup_threads,
        "cpu_model": _cpu_model(), "nproc": os.cpu_count(), "affinity_cpus": affinity, "cpu_share": used_b, "mem_available_gb": mem_gb,
        "sample": ("oracle (C++ restatement of ApplyPressureStressMatrix + pcg_external_matrix_A) on the same cavity scene at %d^3 (n = %d DOFs%s), setup sweeps on %d threads "
                   "(%.1f s wall, not in value); per CG iteration: baseline A (reference-shaped, 3 omp sections, per-call McInv*G) %.1f ms on %d threads; baseline B "
                   "(fair CSR passes, OpenMP rows) %s ms on %s threads (affinity: %d CPUs); value = best of A / B x the GPU run's %d iterations%s"
                   % (ns, n_s, ": the benchmark's own system" if measured_at_size else "", setup_threads, setup_ms * 1e-3, ms_it_a, used_a,
                      " / ".join("%.1f" % by[k] for k in by), " / ".join(by), affinity, gpu_iters,
                      "" if measured_at_size else " x the DOF ratio %.2f — EXTRAPOLATED from %d^3, not measured at the benchmark size" % (dof_ratio, ns))),
        "sample_res": ns, "sample_dofs": n_s, "sample_setup_ms": setup_ms, "sample_iterations_timed": 10,
        "baseline_A_reference_shaped": {"ms_per_cg_iter": ms_it_a, "threads": used_a, "solve_ms_per_step": solve(ms_it_a)},
        "baseline_B_fair_openmp": {"ms_per_cg_iter": ms_it_b, "threads": used_b, "solve_ms_per_step": solve(ms_it_b),
                                   "by_threads": {k: {"ms_per_cg_iter": v, "solve_ms_per_step": solve(v)} for k, v in by.items()}},
        "linearity": lin,
    }
    if not measured_at_size:
        out["extrapolated"] = "measured at %d^3 (%d DOFs), scaled to the benchmark size: x%.2f in DOFs for the solve, x%.1f in cells for the setup" % (ns, n_s, dof_ratio, scale_cells)
    return out


def _latest_traffic(kernel="k_spmv_St"):
    """Newest committed PMC summary (profiles/rNN_pmc_traffic.json, written by scripts/profile_round.sh).  The counters
    cannot be collected inside this process: traffic is a pointer to that measurement, and says which."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d.get(kernel, {}).get("traffic_bytes_per_launch"), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def box_calibration():
    """What THIS box's memory system reaches on a plain device-to-device copy, and which box it is (the
    4-13 % spread of the headline between runs is a spread between BOXES — profiles/r05_cold_warm.md — every kernel of the iteration,
    the pure streaming ones included, moves with this number; one box repeats to 0.1 % over 30 steps, an idle minute and a second process)."""
    import subprocess
    import torch
    out = {}
    try:
        n = 1 << 27                                              # 1 GiB of doubles
        a = torch.empty(n, dtype=torch.float64, device="cuda").fill_(1.0)
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        out["d2d_copy_GBps"] = 2.0 * n * 8 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9      # bytes read + written
        del a, b
        torch.cuda.empty_cache()
    except Exception as e:                                       # noqa: BLE001
        out["d2d_copy_error"] = str(e)[:200]
    try:
        if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
            raise RuntimeError("under a profiler: no child processes")
        txt = subprocess.run(["rocm-smi", "--showuniqueid", "--showclocks", "--showmaxpower", "--showmemvendor", "--showvbios"], capture_output=True, text=True, timeout=30).stdout
        import re
        for key, pat in (("unique_id", r"Unique ID:\s*(\S+)"), ("fclk_MHz", r"fclk clock level: \S+ \((\d+)Mhz\)"), ("mclk_MHz", r"mclk clock level: \S+ \((\d+)Mhz\)"),
                         ("max_power_W", r"Max Graphics Package Power \(W\):\s*([0-9.]+)"), ("vbios", r"VBIOS version:\s*(\S+)"), ("mem_vendor", r"GPU memory vendor:\s*(\S+)")):
            m = re.search(pat, txt)
            if m:
                out[key] = m.group(1)
    except Exception:                                            # noqa: BLE001
        pass
    return out


def _spawn(args):
    """`bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run BEFORE this process touches the
    GPU (no exec of a process that has initialised HIP), wait, return their exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def strong_block(solver, world, rank, dist, pre, transport_used, res, barrier, steps=3):
    """The strong-scaling point of this world size: one res^3 coiling column (config 4) cut `world` ways."""
    import torch
    from polystokes_amd import scenes
    dims = {1: (1, 1, 1), 2: (1, 1, 2), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world, (1, 1, world))
    t_gen = time.perf_counter()
    try:
        if world == 1:
            sc, p = scenes.coil(res, tile=16, pad=2)
            p.preconditioner = pre
            brick = None
        else:
            sc, p, brick = scenes.scene_brick("coil", res, dims, rank, tile=16, pad=2, precond=pre, weak=False)
        ok, why = True, ""
    except Exception as e:                                   # noqa: BLE001  (a world size the 16-cell cuts cannot serve)
        ok, why = False, str(e)
    if dist is not None:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item()) == 1
    if not ok:
        return {"skipped": "no %d-way decomposition of the %d^3 grid: %s" % (world, res, why or "another rank failed")}
    gen_s = time.perf_counter() - t_gen
    try:
        solver.upload(sc, p)
        if brick is not None:
            solver.set_brick(brick)
        ok, why = True, ""
    except Exception as e:                                   # noqa: BLE001  (e.g. out of memory: agree on it BEFORE the first collective of the step)
        ok, why = False, str(e)[:300]
    if dist is not None:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item()) == 1
    if not ok:
        return {"error": "upload failed: %s" % (why or "on another rank")}
    solver.step_device()                                     # warm-up (allocations, first-touch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        rc = solver.step_device()
    barrier()
    el = time.perf_counter() - t0
    nsys = solver.nP + solver.nT
    if dist is not None:
        t = torch.tensor([el, float(nsys)], dtype=torch.float64)
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t)
        el, nsys = float(tm[0].item()), int(t[1].item())
    st = solver.stats
    blk = {"workload": "synthetic coiling column %d^3 (BASELINE config 4), reduced tiles (tile=16, pad=2), tol 1e-3" % res,
           "decomposition": "single domain" if world == 1 else "%dx%dx%d bricks" % dims, "n_gpus": world, "scaling": "strong",
           "steps": steps, "warmup": 1, "ms_per_step": el * 1e3 / steps, "cg_iterations": int(st.solveData[1]), "result": int(rc),
           "system_dofs": nsys, "solve_ms": float(st.stage_ms[8]), "scene_generation_s": gen_s}
    if transport_used:
        blk["transport"] = transport_used
    if dist is not None:
        mine = solver.dist_stats()
        allst = [None] * world
        dist.all_gather_object(allst, mine)
        ex = [d["exchange_ms_per_transport"] for d in allst if d["exchange_ms_per_transport"] is not None]
        ar = [d["allreduce_ms"] for d in allst if d["allreduce_ms"] is not None]
        blk["multi_gpu"] = {"overlap": all(d["overlap"] for d in allst),
                            "halo_bytes_per_iter": {"max_per_rank": max(d["halo_bytes_per_iter"] for d in allst), "sum": sum(d["halo_bytes_per_iter"] for d in allst)},
                            "exchange_ms_per_iter": (2.0 * max(ex)) if ex else None, "allreduce_ms_per_iter": (2.0 * max(ar)) if ar else None,
                            "owned_rows": {"max": max(d["owned_dofs"] for d in allst), "min": min(d["owned_dofs"] for d in allst)}}
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", dest="n", type=int, default=int(os.environ.get("PS_BENCH_RES", "0")), help="grid resolution per axis (default 256; 512 with --scaling strong)")
    ap.add_argument("--scene", choices=["cavity", "coil", "spheres"], default=os.environ.get("PS_BENCH_SCENE") or None, help="default: cavity (config 3); coil with --scaling strong (config 4)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=os.environ.get("PS_BENCH_SCALING", "weak"),
                    help="N > 1: weak = the n x n x (n N) cavity, one n-layer slab per GPU (default); strong = one n^3 scene cut into N slabs "
                         "(BASELINE config 4: --scaling strong --scene coil --res 512; config 5: --scene spheres --res 256).  A driver that can only "
                         "pass --gpus N selects the strong series with the environment: PS_BENCH_SCALING=strong [PS_BENCH_SCENE=coil PS_BENCH_RES=512]")
    ap.add_argument("--precond", choices=["jacobi", "identity", "chebyshev", "chebyshev64"], default="jacobi",
                    help="jacobi (default: the metric's configuration), identity (the reference's default), chebyshev (this library's polynomial preconditioner, degree 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-res", type=int, default=0, help="resolution of the CPU baseline's sample; 0 (default): the benchmark's own resolution — measured, not extrapolated — when the host has the memory")
    ap.add_argument("--bricks", default=os.environ.get("PS_BENCH_BRICKS", ""), help="N > 1: DXxDYxDZ ranks per axis (e.g. 2x2x2) instead of N z-slabs; "
                    "weak: every GPU owns n^3 cells of the (n DX) x (n DY) x (n DZ) cavity; strong: one n^3 scene cut into bricks")
    ap.add_argument("--transport", choices=["rccl", "tcp"], default="rccl",
                    help="N > 1: rccl = one GPU per rank over RCCL/xGMI (the measured configuration); tcp = host-staged sockets, all ranks "
                         "may share GPU 0 — a REHEARSAL of the multi-process path on a single-GPU box, not a performance number")
    ap.add_argument("--no-strong-512", action="store_true", default=os.environ.get("PS_BENCH_NO_STRONG", "") not in ("", "0"),
                    help="skip the strong_512 block (BASELINE config 4: 3 steps of the 512^3 coil cut N ways, appended to the line of every default run)")
    ap.add_argument("--no-other-preconditioners", action="store_true", help="skip the identity / Chebyshev lines that follow the headline (SURVEY 8(d) config 3)")
    ap.add_argument("--strong-res", type=int, default=int(os.environ.get("PS_BENCH_STRONG_RES", "512")), help="resolution of the strong_512 block (tests use a small one)")
    ap.add_argument("--maxit", type=int, default=0, help="cap on solver iterations (profiling runs only; 0 = node default 5000)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # stdout carries exactly ONE line (the JSON): everything else any library writes to fd 1 (gloo announces its connections
    # there) goes to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import polystokes_amd
    from polystokes_amd import _abi as abi
    from polystokes_amd import scenes

    dist = None
    if world > 1:
        # torch.distributed is the launcher's rendezvous only: gloo (CPU) carries the RCCL unique id, the barrier and the
        # max-over-ranks of the wall time.  The data path is the library's own RCCL communicator (dlopen'ed librccl, the copy
        # torch has already mapped); torch's NCCL backend is never initialised, so one RCCL instance owns the device.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # fewer GPUs than ranks (a rehearsal on a small box): ranks share the GPUs there are — RCCL refuses duplicate devices, so
    # such a run ends up on the host-staged transport (explicitly with --transport tcp, or through the fallback below)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)

    strong = args.scaling == "strong"
    scene_name = args.scene or ("coil" if strong else "cavity")
    n = args.n or (512 if strong else 256)
    pre = {"jacobi": abi.PRE_DIAGONAL, "identity": abi.PRE_IDENTITY, "chebyshev": abi.PRE_CHEBYSHEV_F32, "chebyshev64": abi.PRE_CHEBYSHEV}[args.precond]
    kw = dict(tile=16, pad=2, precond=pre)
    solver = polystokes_amd.Solver(local_rank)
    slab = None
    dims = None
    if args.bricks and world > 1:
        dims = tuple(int(v) for v in args.bricks.lower().split("x"))
        if len(dims) != 3 or dims[0] * dims[1] * dims[2] != world:
            raise SystemExit("bench.py: --bricks %s does not multiply to %d ranks" % (args.bricks, world))
    if world == 1:
        if scene_name == "cavity":
            sc, p = scenes.cavity(n, **kw)
        else:
            sc, p = getattr(scenes, scene_name)(n, tile=16, pad=2)
            p.preconditioner = pre
        grid = [n, n, n]
    elif dims is not None:
        # bricks (ps_set_brick): the decomposition along all three axes, every rank generating its own box (+ halo blocks)
        if not strong and scene_name != "cavity":
            raise SystemExit("weak scaling is defined for the cavity scene; use --scaling strong for %s" % scene_name)
        sc, p, slab = scenes.scene_brick(scene_name, n, dims, rank, weak=not strong, **kw)
        grid = [n, n, n] if strong else [n * dims[0], n * dims[1], n * dims[2]]
    elif strong:
        # strong scaling: ONE n^3 scene cut into `world` z-slabs at multiples of 16 (tile-aligned); every rank generates
        # only its own layers (+ one halo block per interior side)
        sc, p, slab = scenes.scene_slab(scene_name, n, world, rank, **kw)
        grid = [n, n, n]
    else:
        if scene_name != "cavity":
            raise SystemExit("weak scaling is defined for the cavity scene; use --scaling strong for %s" % scene_name)
        # weak scaling: the n x n x (n*world) cavity cut into z-slabs of n layers, one per GPU, coupled through the
        # one-layer halo exchange + scalar all-reduces over RCCL
        sc, p, slab = scenes.cavity_slab(n, world, rank, **kw)
        grid = [n, n, n * world]
    if args.maxit > 0:
        p.maxSolverIterations = args.maxit   # the BiCGStab fallback then runs too: use for kernel profiling only
    solver.upload(sc, p)           # host -> HBM, outside the timed region
    transport_used = None
    if world > 1:
        if dims is not None:
            solver.set_brick(slab)
        else:
            solver.set_slab(slab)

        def tcp_init():
            port = torch.zeros(1, dtype=torch.int64)
            if rank == 0:
                import socket
                with socket.socket() as so:      # a free base port; the ranks listen on port + rank
                    so.bind(("127.0.0.1", 0))
                    port[0] = 20000 + so.getsockname()[1] % 20000
            dist.broadcast(port, 0)
            solver.comm_init_tcp(rank, world, "127.0.0.1", int(port.item()))

        def all_ok(ok):                          # every rank learns whether EVERY rank succeeded (gloo, CPU)
            t = torch.tensor([1 if ok else 0], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        if args.transport == "tcp":
            tcp_init()
            transport_used = "tcp (host-staged rehearsal)"
        else:
            # RCCL, with a way out that still produces a line: if librccl cannot be loaded, the communicator cannot be built or its
            # self-test (all-reduce + grouped send/recv) fails on ANY rank, ALL ranks switch to the host-staged transport and the
            # line says so — slower, but the multi-GPU solve is measured instead of lost.  Each step is agreed on over gloo before
            # the next collective call, so no rank waits inside RCCL for a rank that has already given up.
            why = ""
            uid = torch.zeros(128, dtype=torch.uint8)
            try:
                mine = polystokes_amd.comm_unique_id()       # loads librccl on every rank; only rank 0's id is used
                if rank == 0:
                    uid.copy_(torch.tensor(list(mine), dtype=torch.uint8))
                ok = True
            except Exception as e:                           # noqa: BLE001
                ok, why = False, "load: %s" % e
            if all_ok(ok):
                dist.broadcast(uid, 0)
                try:
                    solver.comm_init(bytes(uid.tolist()), rank, world)
                    solver.comm_selftest()
                    ok = True
                except Exception as e:                       # noqa: BLE001
                    ok, why = False, "init/self-test: %s" % e
                ok = all_ok(ok)
            else:
                ok = False
            if ok:
                transport_used = "rccl"
            else:
                sys.stderr.write("[bench rank %d] RCCL transport unavailable (%s): falling back to the host-staged transport\n" % (rank, why or "another rank failed"))
                tcp_init()
                transport_used = "tcp (FALLBACK: RCCL unavailable%s)" % ((": " + why[:160]) if why else " on another rank")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    box = box_calibration() if rank == 0 else None
    for _ in range(args.warmup):
        solver.step_device()
    barrier()
    t0 = time.perf_counter()
    results = []
    for _ in range(args.steps):
        results.append(solver.step_device())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)

    st = solver.stats
    iters = int(st.solveData[1])
    solve_ms = float(st.stage_ms[8])
    nsys = solver.nP + solver.nT
    if dist is not None:   # local systems include the halo DOFs: an upper bound of the whole-job DOF count
        t = torch.tensor([float(nsys)], dtype=torch.float64)
        dist.all_reduce(t)
        nsys_total = int(t.item())
    else:
        nsys_total = nsys

    # roofline of the dominant kernels, measured live with HIP events on the solver stream
    coded = bool(solver.array("valuesCoded")[0])
    c16 = int(solver.array("columns16")[0]) == 3
    kern = {}
    # the PCG step as the solve ran it: four kernels (S, tiles, St with the residual update in its epilogue, x/p update) on
    # the coded stream, five (S, tiles, St, r update, x/p update) otherwise — the five-kernel ones stay listed: they are what
    # the distributed, Chebyshev and fallback-stream solves launch
    fused = int(solver.array("fusedStep")[0]) == 1
    rpl = solver.array("rowPerLane")
    ell = int(rpl[0]) == 3                                  # both products on the row-per-lane kernels
    names = (["spmv_St_r", "cg_update_xp_u"] if fused else []) + ["spmv_St", "spmv_S", "apply", "tiles", "cg_update_r", "cg_update_xp"]
    if coded and c16:
        names += ["spmv_St_fp64", "spmv_S_fp64"]     # the pipelined kernels on fp64 values (10 B/nnz): the non-dyadic-weights fallback
    names += ["spmv_St_csr", "spmv_S_csr"]            # the one-shot kernels on the plain CSR (12 B/nnz): the last-resort fallback
    # the five kernels of a CG iteration are timed IN SEQUENCE ("seq:": the loop's predecessor kernel runs, untimed, before
    # every timed launch: what rocprof sees in a real solve); "replayed_ms" is the same kernel launched back to back
    in_loop = ("spmv_St_r", "cg_update_xp_u", "spmv_St", "spmv_S", "tiles", "cg_update_r", "cg_update_xp")
    for name in names:
        ms, by = solver.bench_kernel(("seq:" + name) if name in in_loop else name, 20)
        kern[name] = {"ms": ms, "algorithmic_bytes": by, "GBps": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                      "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0}
        if name in in_loop:
            kern[name]["replayed_ms"] = solver.bench_kernel(name, 20)[0]
    dom = "spmv_St_r" if fused else "spmv_St"
    mode = 3 if fused else 0
    traffic, traffic_source = _latest_traffic("k_spmv_St_r" if fused else "k_spmv_St") if (n == 256 and world == 1 and scene_name == "cavity") else (None, None)
    # achieved = the bytes the production kernel has to move per launch (its stored matrix format + the vectors, each once)
    # / its launch duration: the HBM utilisation of the kernel as it runs.  The production kernel streams a lossless 3 B/nnz
    # encoding of the matrix, so this is FEWER bytes than SURVEY section 8(d)'s CSR figure (12 B/nnz + row pointers +
    # vectors); that figure over the same launch duration is reported beside it ("csr_equivalent": it can exceed the
    # HBM peak, which only says the kernel beats a CSR SpMV running at the roofline).  Every fraction in "other_kernels" is
    # that kernel's OWN stored bytes over its OWN measured duration (the fallbacks are timed as themselves).
    # must-move bytes: the stored bytes minus the part of the 3 B/nnz matrix stream that chunks sharing a run read from cache —
    # the vectors (each once), the per-row streams, the chunk records and the DISTINCT stream entries: what has to cross the HBM interface
    # per launch.  `traffic / must_move_bytes` is the re-fetch factor of the gathers (S 2.08x, St 1.34x at the end of r03).
    runs = [int(v) for v in solver.array("streamRuns")]          # S distinct, S entries, St distinct, St entries
    if coded and c16:
        for nm, (dist_e, all_e) in (("spmv_S", (runs[0], runs[1])), ("spmv_St", (runs[2], runs[3])), ("spmv_St_r", (runs[2], runs[3]))):
            if nm in kern:
                kern[nm]["must_move_bytes"] = kern[nm]["algorithmic_bytes"] - 3.0 * (all_e - dist_e)
    ms = kern[dom]["ms"]
    csr = kern["spmv_St_csr"]["algorithmic_bytes"]
    csr_gbps = csr / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": ("k_spmv_St_ell<%d,POL,FX> (one lane per row on the kind-major numbering; POL = cache policy of its streams%s)" % (mode, "; MODE 3: r -= alpha A p in the epilogue, FX = 3: the plain single-domain step (coded uInv, no halo rows) — run as k_spmv_St_ell2<POL>, two 64-row units in flight per wave, unless PS_ST_DUAL=0" if fused else "")) if ell else
                                  ("k_spmv_St_pipe<%d,NV,%s,POL> (NV = 1 or 2 four-entry groups per lane, by the fullest chunk; POL = cache policy of its streams%s)" % (mode, "false" if coded else "true", "; MODE 3: r -= alpha A p in the epilogue" if fused else "") if c16 else "k_spmv_St<0,6,%s>" % ("true" if coded else "false")),
        "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["frac"],
        "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": kern[dom]["algorithmic_bytes"], "avg_launch_ms": ms,
        "must_move_bytes": kern[dom].get("must_move_bytes"),
        "numbering": {0: "voxel-major (kinds of DOF / faces of a voxel adjacent)", 3: "kind-major inside every k-plane of a 16^3 lattice block (DOFs and face rows)"}.get(int(rpl[1]), "mixed (%d)" % int(rpl[1])),
        "algorithmic_bytes_definition": ("stored format: 3*nnz (16-bit windowed column + int8 value code; padded slots of the row-per-lane layout NOT counted) + %s80 B per chunk "
                                         "+ 8*cols (t once) + 8*rows (p) + 1*rows (coded uInv) + %s") % ("" if ell else "1*rows (row length) + ", "16*rows (r read + written) + 2*rows (Jacobi diagonal, stored in 16 bits: the upper half of its fp32 value); A p is not stored" if fused else "8*rows (y)") if (coded and c16) else
                                        "CSR: (12 | 10 | 5)*nnz + 4*(rows+1) + 8*rows (y) + 8*cols (x once) + 16*rows (fused epilogue)",
        "stream_runs": (lambda r: {"S_distinct_entries": int(r[0]), "S_entries": int(r[1]), "St_distinct_entries": int(r[2]), "St_entries": int(r[3]),
                                   "note": "chunks with byte-identical (col16, code, row length) runs share one run, so most of the matrix stream is served from cache: algorithmic bytes still count every entry once per launch (the loads are issued), the HBM bytes are in `traffic`"})(solver.array("streamRuns")),
        "value_format": ("16-bit windowed col + int8 value code (3 B/nnz, lossless)" if c16 else "int32 col + int8 value code (5 B/nnz, lossless)") if coded
                        else ("16-bit windowed col + fp64 value (10 B/nnz)" if c16 else "int32 col + fp64 value (12 B/nnz)"),
        "csr_equivalent": {"bytes_per_launch": csr, "GBps": csr_gbps, "frac_of_peak_if_it_moved_them": csr_gbps / HBM_PEAK_GBS,
                           "definition": "SURVEY 8(d): 12*nnz + 4*(rows+1) + 8*rows + 8*cols + 16*rows, over the production kernel's launch duration (not bytes it moves)"},
        "other_kernels": {k: v for k, v in kern.items() if k != dom},
    }

    link = "RCCL halo exchange + all-reduce" if transport_used == "rccl" else "host-staged TCP transport (%s; not the RCCL number)" % transport_used
    if world == 1:
        par = "1 GPU"
    elif dims is not None:
        par = ("%dx%dx%d bricks of one %d^3 scene (strong), %s" % (dims + (n, link)) if strong
               else "%dx%dx%d bricks, %d^3 owned cells per GPU (weak), %s" % (dims + (n, link)))
    else:
        par = ("%d z-slabs of one %d^3 scene (strong), %s" % (world, n, link) if strong
               else "%d z-slabs, one %d-layer slab per GPU (weak), %s" % (world, n, link))
    workload = {"cavity": "synthetic lid-driven cavity", "coil": "synthetic coiling column (honey_coil stand-in)", "spheres": "pool with 8 moving solid spheres (armadillos stand-in)"}[scene_name]
    out = {
        "metric": "Stokes-solve wall ms/step (assembly+PCG) on 256^3 grid; CG iters/sec",
        "value": ms_per_step, "unit": "ms/step", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        # every vector, sum and recurrence of the solve is fp64; the ONE array not read in fp64 is the Jacobi preconditioner's diagonal
        # (an extension: the reference's Jacobi is a stub) — any fixed positive diagonal preconditions, its rounding moves the count by one
        "preconditioner_storage": ("diagonal of the Jacobi / Chebyshev extensions read as 16 bits per DOF (upper half of its fp32 value)" +
                                   "; equivalence with the exact fp64 diagonal pinned by tests/test_gpu_parity.py::test_stored_diagonal_jacobi_is_equivalent_to_exact_jacobi and, at 5.9 M DOFs (468 iterations either way), "
                                   "tests/test_golden.py::test_hip_jacobi_matches_the_exact_diagonal_oracle_at_real_size"
                                   if args.precond in ("jacobi", "chebyshev", "chebyshev64") else None),
        "config": {"workload": "%s %dx%dx%d, reduced tiles (tile=16, pad=2), %s-PCG, tol 1e-3" % (workload, grid[0], grid[1], grid[2], args.precond),
                   "grid": grid, "parallelism": par},
        "cg_iterations": iters, "cg_iters_per_s": iters / (solve_ms * 1e-3) if solve_ms > 0 else 0.0,
        "system_dofs": nsys_total, "dof_iterations_per_s": nsys_total * iters / (solve_ms * 1e-3) if solve_ms > 0 else 0.0, "active_faces": solver.nA, "regions": solver.nRegions, "result": int(st.result),
        "stage_ms": {abi.STAGE_NAMES[i]: float(st.stage_ms[i]) for i in range(len(abi.STAGE_NAMES))},
        "roofline": roofline,
        "box": box,
    }
    if transport_used:
        out["transport"] = transport_used
    if dist is not None:
        # what explains the scaling curve: every rank's cut traffic, the sampled transport and all-reduce times, the owned rows
        mine = solver.dist_stats()
        mine["rank"] = rank
        allst = [None] * world
        dist.all_gather_object(allst, mine)
        if rank == 0:
            ex = [d["exchange_ms_per_transport"] for d in allst if d["exchange_ms_per_transport"] is not None]
            ar = [d["allreduce_ms"] for d in allst if d["allreduce_ms"] is not None]
            out["multi_gpu"] = {
                "overlap": all(d["overlap"] for d in allst),
                "halo_bytes_per_iter": {"max_per_rank": max(d["halo_bytes_per_iter"] for d in allst), "sum": sum(d["halo_bytes_per_iter"] for d in allst)},
                "exchange_ms_per_iter": (2.0 * max(ex)) if ex else None,        # two exchanges per iteration (x layers out, A p contributions back), slowest rank
                "allreduce_ms_per_iter": (2.0 * max(ar)) if ar else None,       # two scalar all-reduces per iteration, incl. their synchronisation
                "owned_rows": {"max": max(d["owned_dofs"] for d in allst), "min": min(d["owned_dofs"] for d in allst)},
                "samples": {"exchange": min(d["exchange_samples"] for d in allst), "allreduce": min(d["allreduce_samples"] for d in allst)},
                "note": "exchange = one transport timed with events on the rank's comm stream at the end of every 25-iteration batch; it runs UNDER the S / St chunks that need no halo value when overlap is true",
            }
    if world == 1:
        # the boundary as the Houdini shim uses it: host fp32 fields in, velocity / valid fields out (polystokes_step)
        t0 = time.perf_counter()
        solver.step(sc, p)
        out["pcie_inclusive_ms"] = (time.perf_counter() - t0) * 1e3
    # BASELINE config 4 / north_star's "1 -> 8-GPU scaling curve at 512^3" from the SAME invocation: the 512^3 coiling column cut
    # N ways (1: single domain; 2: z-slabs; 4: 2x2x1 bricks; 8: 2x2x2 bricks), 1 warm-up + 3 timed steps, after the headline
    # measurement and on the same communicator — a driver that only passes --gpus N still records the strong series.
    # "Jacobi-PCG and identity-PCG both reported" — the headline above is the metric's Jacobi; the reference's
    # live default (identity: exec/HDK_PolyStokesSolver_Preconditioners.cpp:4-9,31-35) and this library's polynomial preconditioner
    # follow from the same invocation, 1 warm-up + 3 steps each, same scene, same resident fields
    if world == 1 and scene_name == "cavity" and args.maxit == 0 and not args.no_other_preconditioners:
        others = {}
        # chebyshev4: the polynomial with its inner vectors stored as fp32 (PS_PRE_CHEBYSHEV_F32, r06; every sum, r, the outer PCG and its stop
        # rule fp64); chebyshev4_fp64: the all-fp64 polynomial of r04 / r05 under the same key those rounds reported it
        # chebyshev10: the same with ten terms (profiles/r06_cheb32_degree.txt: the degree is flat on the cavity and the coil — 625 ... 657 ms at 3 ... 12 terms —
        # and worth 30 % on the stiff spheres scene; ten terms also mean 2.3x fewer outer iterations, i.e. all-reduces, for a decomposition)
        for nm, code, deg in (("jacobi", abi.PRE_DIAGONAL, 0), ("identity", abi.PRE_IDENTITY, 0), ("chebyshev4", abi.PRE_CHEBYSHEV_F32, 0), ("chebyshev4_fp64", abi.PRE_CHEBYSHEV, 0),
                              ("chebyshev10", abi.PRE_CHEBYSHEV_F32, 10)):
            if code == pre and deg == 0:
                continue
            try:
                p.preconditioner = code
                p.preconditionerDegree = deg
                solver.upload(sc, p)
                solver.step_device()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    rc2 = solver.step_device()
                torch.cuda.synchronize()
                st2 = solver.stats
                others[nm] = {"ms_per_step": (time.perf_counter() - t0) * 1e3 / 3, "cg_iterations": int(st2.solveData[1]), "result": int(rc2),
                              "solve_ms": float(st2.stage_ms[8]), "steps": 3, "warmup": 1}
                if nm.startswith("chebyshev"):
                    others[nm]["inner_vectors"] = "fp32" if int(solver.array("chebInner32")[0]) else "fp64"
            except Exception as e:                               # noqa: BLE001  (the headline is already measured: never lose it)
                others[nm] = {"error": str(e)[:300]}
        p.preconditioner = pre
        p.preconditionerDegree = 0
        out["other_preconditioners"] = others
    redundant = strong and scene_name == "coil" and n == args.strong_res
    if not args.no_strong_512 and not redundant and args.maxit == 0:
        # the headline is measured: whatever happens in this block (out of memory at 78 M DOFs, a solver exception) must not lose the line,
        # and with N > 1 no rank may be left waiting in a barrier — the ranks agree on success before the next collective
        try:
            blk = strong_block(solver, world, rank, dist, pre, transport_used, args.strong_res, barrier)
            ok, why = True, ""
        except Exception as e:                                   # noqa: BLE001
            blk, ok, why = None, False, str(e)[:300]
        if dist is not None:
            t = torch.tensor([1 if ok else 0], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) != 1 and ok:
                blk, ok, why = None, False, "another rank failed"
        out["strong_512"] = blk if ok else {"error": why}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.precond.startswith("chebyshev") and scene_name == "cavity":   # the CPU leg times the reference's own (Jacobi / identity) PCG iteration on the headline scene
        out["cpu_baseline"] = cpu_baseline(n ** 3, nsys, iters, dict(tile=16, pad=2, precond=kw["precond"]), args.cpu_sample_res, second_res=(96 if n > 96 else 0), bench_res=n)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
