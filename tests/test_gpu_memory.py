"""Device memory across steps.  A step never calls hipFree in its middle (a device-wide synchronisation; with
threaded ranks a deadlock): buffers it drops or grows wait on the CONTEXT's list (ps_common.hpp: DeferredFrees) and are released where the context's
stream has just been synchronised — at the end of every setup / solve / step, failed ones included.  These tests pin that
  * setup-only loops and alternating scenes do not grow the memory held, and nothing waits for release after any call;
  * a failing step (forced out-of-memory: PS_DEBUG_ALLOC_LIMIT) fails with a message, leaves nothing waiting, and the context goes on working;
  * with several asynchronous ranks in ONE process (the test arrangement of tests/mp_rank.py "r0,r1") running out of memory FAILS the ranks through
    ps_last_error instead of releasing inside the step (which could hang two ranks)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from polystokes_amd import _abi as abi
from polystokes_amd import scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def test_setup_loops_and_scene_changes_do_not_grow_device_memory():
    import polystokes_amd
    s = polystokes_amd.Solver(0)
    big, small = scenes.cavity(48, precond=abi.PRE_DIAGONAL), scenes.coil(32)
    seen = []
    for rnd in range(4):
        for sc, p in (big, small):
            s.upload(sc, p)
            for _ in range(3):                      # setup-only loop (ps_setup_device never reached a release point before r06)
                s.setup()
                assert s.memory_stats()["deferred_bytes"] == 0
            s.solve()
            assert s.memory_stats()["deferred_bytes"] == 0
            assert s.step_device() == abi.SUCCESS
            m = s.memory_stats()
            assert m["deferred_bytes"] == 0
            seen.append(m["live_bytes"])
    # after the first round every buffer has reached the size of the larger scene: the bytes held repeat exactly
    assert len(set(seen[2:])) <= 2 and max(seen[2:]) == max(seen[:2]), seen
    assert seen[-1] == seen[-3] and seen[-2] == seen[-4], seen
    s.close()
    assert polystokes_amd.process_memory_stats()["live_bytes"] == 0, "closing the only context must return all its memory"


_CHILD = (
    "import sys, json, numpy as np\n"
    f"sys.path.insert(0, {ROOT!r})\n"
    "import polystokes_amd\nfrom polystokes_amd import scenes, _abi as abi\n"
    "s = polystokes_amd.Solver(0)\n"
    "out = {}\n"
    "sc, p = scenes.cavity(64, precond=abi.PRE_DIAGONAL)\n"
    "try:\n"
    "    s.step(sc, p)\n"
    "    out['big'] = 'ok'\n"
    "except polystokes_amd.PolyStokesError as e:\n"
    "    out['big'] = str(e)\n"
    "out['after_big'] = s.memory_stats()\n"
    "sc2, p2 = scenes.cavity(16, tile=8)\n"
    "out['small_rc'] = s.step(sc2, p2)\n"
    "out['small_it'] = int(s.stats.solveData[1])\n"
    "out['after_small'] = s.memory_stats()\n"
    "s.close()\n"
    "print('RESULT ' + json.dumps(out))\n"
)


def _child(env_extra):
    import json
    env = dict(os.environ, **env_extra)
    pr = subprocess.run([sys.executable, "-c", _CHILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert pr.returncode == 0, pr.stderr[-3000:]
    line = [l for l in pr.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_forced_out_of_memory_fails_the_step_and_the_context_recovers():
    free = _child({})
    assert free["big"] == "ok" and free["small_rc"] == abi.SUCCESS
    peak = free["after_big"]["peak_bytes"]
    limit = peak // 2                                # enough for the fields and the first setup stages, not for the system
    lim = _child({"PS_DEBUG_ALLOC_LIMIT": str(limit)})
    assert "out of device memory" in lim["big"], lim["big"]
    assert lim["after_big"]["deferred_bytes"] == 0            # the failed step's dropped buffers were released on the error path
    assert lim["after_big"]["live_bytes"] <= limit
    assert lim["small_rc"] == abi.SUCCESS and lim["small_it"] == free["small_it"]   # the same context goes on working
    assert lim["after_small"]["deferred_bytes"] == 0


def test_threaded_ranks_fail_instead_of_releasing_inside_a_step(tmp_path):
    """Two asynchronous ranks (stand-in transport) as threads of ONE process, a limit between what their uploads hold and what their steps need:
    both return through ps_last_error — no hipFree inside the step, no hang."""
    stub = os.path.join(HERE, "stub_rccl", "libps_stub_rccl.so")
    assert os.path.exists(stub), "build it: make -C tests/stub_rccl"
    outs = [str(tmp_path / ("r%d.npz" % r)) for r in range(2)]

    def run(extra, port):
        env = dict(os.environ, PS_TEST_TRANSPORT="stub", PS_RCCL_LIB=stub, **extra)
        pr = subprocess.run([sys.executable, os.path.join(HERE, "mp_rank.py"), "cavity_w2", "2", "0,1", str(port), ",".join(outs)],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
        return pr, [np.load(o) for o in outs]

    pr, res = run({}, 29710)
    assert pr.returncode == 0, pr.stdout[-3000:]
    assert all(int(r["rc"]) == abi.SUCCESS for r in res)
    after_upload = max(int(r["mem_after_upload"]) for r in res)
    peak = max(int(r["mem_peak"]) for r in res)
    assert peak > after_upload
    limit = after_upload + (peak - after_upload) // 3
    for o in outs:
        os.remove(o)
    pr, res = run({"PS_DEBUG_ALLOC_LIMIT": str(limit)}, 29720)
    assert pr.returncode == 0, pr.stdout[-3000:]
    errs = [str(r["err"]) for r in res]
    assert all(int(r["rc"]) == -1 for r in res), errs
    assert any("out of device memory" in e for e in errs), errs
    assert all(("out of device memory" in e) or ("another rank failed" in e) for e in errs), errs
    assert any("several ranks of one communicator share this process" in e for e in errs) or all("out of device memory" in e for e in errs), errs
