#!/usr/bin/env python3
"""One rank of a real multi-process distributed step (one process per rank, ps_step_device on a slab context) with the
host-staged TCP transport — several ranks may share one GPU.  Started by tests/test_gpu_multiprocess.py:
    mp_rank.py <case> <world> <rank> <base_port> <out.npz> [device]
Writes the rank's owned faces (local layout), masks, result code and iteration count.
<rank> may be a list "r0,r1" (with <out.npz> a matching list): the process then holds SEVERAL ranks, one thread and one ps_context each
(the box admits six GPU processes: eight ranks run as four processes of two; the library calls release the GIL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    case, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[4])
    ranks, outs = [int(r) for r in sys.argv[3].split(",")], sys.argv[5].split(",")
    device = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    if len(ranks) == 1:
        run_rank(case, world, ranks[0], port, outs[0], device, None)
        return
    import threading
    if os.environ.get("PS_DBG_DUMP"):           # debugging aid: where every thread stands after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PS_DBG_DUMP"]), exit=False)
        if os.environ.get("PS_DBG_BT"):         # native stacks too (scripts/dbg/btdump.c)
            import ctypes
            bt = ctypes.CDLL(os.environ["PS_DBG_BT"])
            threading.Timer(int(os.environ["PS_DBG_DUMP"]) + 2, bt.bt_dump_all).start()
    import polystokes_amd  # noqa: F401  (loaded once, before the threads)
    done = threading.Barrier(len(ranks))        # no rank of this process frees its buffers while another still enqueues work
    errs = []

    def body(r, o):
        try:
            run_rank(case, world, r, port, o, device, done)
        except BaseException as e:              # noqa: BLE001 — reported by the process' exit code
            errs.append((r, repr(e)))
            done.abort()

    th = [threading.Thread(target=body, args=(r, o)) for r, o in zip(ranks, outs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise SystemExit("ranks failed: %s" % errs)


def run_rank(case, world, rank, port, out, device, done):
    import polystokes_amd
    from polystokes_amd import partition
    import mp_cases
    sc, p = mp_cases.make(case)
    dims = mp_cases.DIMS.get(case.replace("_interrupt", "").replace("_failrank", ""))
    s = polystokes_amd.Solver(device)
    if dims is not None:
        sl = partition.make_brick((sc.nx, sc.ny, sc.nz), dims, rank, p.tileSize)
        loc = partition.local_scene_brick(sc, sl)
    else:
        sl = partition.make_slab(sc.nz, world, rank, p.tileSize)
        loc = partition.local_scene(sc, sl)
    if case.endswith("_failrank") and rank == 1:
        # this rank sees air in a patch of its own first layers: its labels on the cut differ from what rank 0 computes
        # from its halo copy: the owners' label exchange finds the two views of the same cells in disagreement
        loc.surface[sl.zLoOwned:sl.zLoOwned + 3, 5:12, 5:12] = 1.0
    s.upload(loc, p)
    if dims is not None:
        s.set_brick(sl)
    else:
        s.set_slab(sl)
    if os.environ.get("PS_TEST_TRANSPORT") == "stub":
        # the asynchronous (RCCL) branch of the transport on the stand-in library of tests/stub_rccl (PS_RCCL_LIB): the unique id
        # travels through a file next to the outputs
        import time
        uid_path = os.path.join(os.path.dirname(out), "%s.w%d.p%d.uid" % (case, world, port))
        if rank == 0:
            with open(uid_path + ".tmp", "wb") as f:
                f.write(polystokes_amd.comm_unique_id())
            os.replace(uid_path + ".tmp", uid_path)
        t0 = time.time()
        while not os.path.exists(uid_path):
            time.sleep(0.01)
            if time.time() - t0 > 120:
                raise SystemExit("rank 0 never wrote the unique id")
        s.comm_init(open(uid_path, "rb").read(), rank, world)
        s.comm_selftest()
    else:
        s.comm_init_tcp(rank, world, "127.0.0.1", port)
    mem_after_upload = s.memory_stats()["live_bytes"]      # (process-wide: all ranks of this process)
    if case.endswith("_interrupt") and rank == world - 1:
        s.set_interrupt(lambda: True)     # ONE rank asks to stop: every rank must return PS_INCOMPLETE at the same batch
    try:
        rc = s.step_device()
        err = ""
    except polystokes_amd.PolyStokesError as e:
        rc, err = -1, str(e)
    res = dict(rc=rc, err=err, iters=int(s.stats.solveData[1]), used_bicgstab=int(s.stats.usedBiCGStab), mem_after_upload=mem_after_upload,
               mem_peak=s.memory_stats()["peak_bytes"], mem_deferred=s.memory_stats()["deferred_bytes"])
    if rc in (0, 1):
        lv, lval = s.download()
        for a in range(3):
            res["vel%d" % a], res["valid%d" % a] = lv[a], lval[a]
            res["owned%d" % a] = s.array("owned" + "XYZ"[a])
        res["labels"] = s.array("centerLabels")
        res["fused"] = int(s.array("fusedStep")[0])
        res["overlap"] = int(bool(s.dist_stats()["overlap"]))
    np.savez(out, **res)
    if done is not None:
        done.wait()
    s.close()


if __name__ == "__main__":
    main()
