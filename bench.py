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


def _cpu_share():
    """Threads this process may keep busy: the scheduler affinity, capped by the cgroup CPU quota and by 16 (a 1-GPU
    box's share of its host; a worker pool larger than the share only thrashes)."""
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
    return max(1, min(n, 16))


def cpu_baseline(n_gpu_cells, gpu_n, gpu_iters, params_kw):
    """CPU restatement (oracle, kind "port") timed on this host's cores on a bounded sample: the same scene at 64^3.
    value = setup (single thread, the restatement is literal) + CG iterations with every row loop of the operator and
    of the vector updates split over OpenMP threads (oracle/ps_oracle_mt.cpp), scaled to the metric's unit (ms/step at
    the benchmark size) by cell count (setup) and by DOFs x the GPU run's iteration count (solve).  The single-thread
    timings of the reference-shaped and of the fused operator are kept alongside."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # before libgomp starts: spinning workers starve a shared host
    from oracle import ps_oracle
    from polystokes_amd import scenes
    ns = 64
    sc, p = scenes.cavity(ns, **params_kw)
    o = ps_oracle.Oracle()
    t0 = time.time()
    o.run(sc, p, solve=False)
    setup_ms = (time.time() - t0) * 1e3
    n_s = o.nP + o.nT
    cores = _cpu_share()
    ms_it = o.time_cg(6, fair=False)
    ms_it_fair = o.time_cg(6, fair=True)
    ms_it_mt, used = o.time_cg_mt(20, cores)
    scale_cells = n_gpu_cells / float(ns ** 3)
    est = setup_ms * scale_cells + ms_it_mt * (gpu_n / float(n_s)) * max(gpu_iters, 1)
    est_1t = setup_ms * scale_cells + ms_it * (gpu_n / float(n_s)) * max(gpu_iters, 1)
    return {
        "value": est, "unit": "ms/step", "cores": used, "kind": "port",
        "sample": ("oracle (C++ restatement of ApplyPressureStressMatrix + pcg_external_matrix_A) on the same cavity scene at 64^3, n=%d: "
                   "setup %.0f ms (1 thread); per CG iteration %.2f ms with %d OpenMP threads (fused operator), %.1f ms single-thread "
                   "reference-shaped, %.1f ms single-thread fused; scaled to 256^3 by cell count (setup) and by DOFs x the GPU run's "
                   "iteration count (solve)" % (n_s, setup_ms, ms_it_mt, used, ms_it, ms_it_fair)),
        "sample_setup_ms": setup_ms, "sample_ms_per_cg_iter_mt": ms_it_mt, "sample_ms_per_cg_iter": ms_it,
        "sample_ms_per_cg_iter_fused": ms_it_fair, "sample_dofs": n_s, "value_single_thread_reference_shaped": est_1t,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", dest="n", type=int, default=256, help="grid resolution per axis (default 256)")
    ap.add_argument("--precond", choices=["jacobi", "identity"], default="jacobi")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--maxit", type=int, default=0, help="cap on solver iterations (profiling runs only; 0 = node default 5000)")
    args = ap.parse_args()

    import torch
    import polystokes_amd
    from polystokes_amd import _abi as abi
    from polystokes_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)

    n = args.n
    kw = dict(tile=16, pad=2, precond=abi.PRE_DIAGONAL if args.precond == "jacobi" else abi.PRE_IDENTITY)
    solver = polystokes_amd.Solver(local_rank)
    if world == 1:
        sc, p = scenes.cavity(n, **kw)
        slab = None
    else:
        # weak scaling: the n x n x (n*world) cavity cut into z-slabs of n layers, one per GPU, coupled through the
        # one-layer halo exchange + scalar all-reduces over RCCL
        sc, p, slab = scenes.cavity_slab(n, world, rank, **kw)
    if args.maxit > 0:
        p.maxSolverIterations = args.maxit   # the BiCGStab fallback then runs too: use for kernel profiling only
    solver.upload(sc, p)           # host -> HBM, outside the timed region
    if world > 1:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.tensor(list(polystokes_amd.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        solver.set_slab(slab)
        solver.comm_init(bytes(uid.cpu().tolist()), rank, world)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)

    st = solver.stats
    iters = int(st.solveData[1])
    solve_ms = float(st.stage_ms[8])
    nsys = solver.nP + solver.nT
    if dist is not None:   # whole-job DOF count (owned DOFs only would need the owned range; local systems include the halo)
        t = torch.tensor([float(nsys)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        nsys_total = int(t.item())
    else:
        nsys_total = nsys

    # roofline of the dominant kernels, measured live with HIP events on the solver stream
    coded = bool(solver.array("valuesCoded")[0])
    kern = {}
    names = ["spmv_St", "spmv_S", "apply", "tiles", "cg_update_r", "cg_update_xp"] + (["spmv_St_fp64", "spmv_S_fp64"] if coded else [])
    for name in names:
        ms, by = solver.bench_kernel(name, 20)
        kern[name] = {"ms": ms, "algorithmic_bytes": by, "GBps": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                      "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0}
    dom = "spmv_St"
    traffic = None
    tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tp) and n == 256 and world == 1:
        try:
            traffic = json.load(open(tp)).get("k_spmv_St", {}).get("traffic_bytes_per_launch")
        except Exception:
            traffic = None
    # achieved = the bytes the production kernel has to move per launch (its stored matrix format + the vectors, each once)
    # / its launch duration: the HBM utilisation of the kernel as it runs.  The production kernel streams a lossless 3 B/nnz
    # encoding of the matrix, so this is FEWER bytes than SURVEY section 8(d)'s CSR figure (12 B/nnz + row pointers +
    # vectors); that figure over the same launch duration is reported beside it ("csr_equivalent": it can exceed the
    # HBM peak, which only says the kernel beats a CSR SpMV running at the roofline), and so is the kernel variant that
    # really streams the fp64 CSR values ("spmv_St_fp64").
    ms = kern[dom]["ms"]
    csr = kern[dom + "_fp64"]["algorithmic_bytes"] if coded else kern[dom]["algorithmic_bytes"]
    csr_gbps = csr / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    c16 = int(solver.array("columns16")[0]) == 3
    roofline = {
        "bound": "hbm", "kernel": "k_spmv_St_pipe<0,%s>" % ("compressed stream" if (coded and c16) else ("int8-coded values" if coded else "fp64 values")),
        "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": traffic,
        "algorithmic_bytes_per_launch": kern[dom]["algorithmic_bytes"], "avg_launch_ms": ms,
        "algorithmic_bytes_definition": ("stored format: 3*nnz (16-bit windowed column + int8 value code) + 1*rows (row length) + 72 B per 256-row chunk "
                                         "+ 8*rows (y) + 8*cols (x once) + 16*rows (fused -1/2 uInv x epilogue)") if (coded and c16) else
                                        "CSR: (12 | 5)*nnz + 4*(rows+1) + 8*rows (y) + 8*cols (x once) + 16*rows (fused epilogue)",
        "value_format": ("16-bit windowed col + int8 value code (3 B/nnz, lossless)" if c16 else "int32 col + int8 value code (5 B/nnz, lossless)") if coded
                        else "int32 col + fp64 value (12 B/nnz)",
        "csr_equivalent": {"bytes_per_launch": csr, "GBps": csr_gbps, "frac": csr_gbps / HBM_PEAK_GBS,
                           "definition": "SURVEY 8(d): 12*nnz + 4*(rows+1) + 8*rows + 8*cols + 16*rows, over the production kernel's launch duration"},
        "other_kernels": {k: v for k, v in kern.items() if k != dom},
    }

    out = {
        "metric": "Stokes-solve wall ms/step (assembly+PCG) on 256^3 grid; CG iters/sec",
        "value": ms_per_step, "unit": "ms/step", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "synthetic lid-driven cavity %d^3, reduced tiles (tile=16, pad=2), %s-PCG, tol 1e-3" % (n, args.precond),
                   "grid": [n, n, n * world], "parallelism": "1 GPU" if world == 1 else "%d z-slabs, RCCL halo exchange + all-reduce" % world},
        "cg_iterations": iters, "cg_iters_per_s": iters / (solve_ms * 1e-3) if solve_ms > 0 else 0.0,
        "system_dofs": nsys_total, "dof_iterations_per_s": nsys_total * iters / (solve_ms * 1e-3) if solve_ms > 0 else 0.0, "active_faces": solver.nA, "regions": solver.nRegions, "result": int(st.result),
        "stage_ms": {abi.STAGE_NAMES[i]: float(st.stage_ms[i]) for i in range(len(abi.STAGE_NAMES))},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n ** 3, nsys, iters, dict(tile=16, pad=2, precond=kw["precond"]))
    if rank == 0:
        print(json.dumps(out))
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
