"""PCIe-inclusive step: polystokes_step (host fields in -> host velocity out) against ps_step_device on resident inputs."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0)
s.step(sc, p)
t = []
for i in range(3):
    t0 = time.perf_counter(); s.step(sc, p); t.append(time.perf_counter() - t0)
s.upload(sc, p); s.step_device()
d = []
for i in range(3):
    t0 = time.perf_counter(); s.step_device(); d.append(time.perf_counter() - t0)
print("n", n, "polystokes_step (host in/out) ms", [round(x * 1e3, 1) for x in t], " ps_step_device ms", [round(x * 1e3, 1) for x in d])
