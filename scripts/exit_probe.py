"""Process-exit probe: RCCL communicator (world 1) + optional oracle library (libgomp) + optional torch; prints before exiting."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if "t" in mode:
    import torch
import polystokes_amd
s = polystokes_amd.Solver(0)
s.comm_init(polystokes_amd.comm_unique_id(), 0, 1)
s.comm_selftest()
if "c" in mode:
    s.close()
if "o" in mode:
    from oracle import ps_oracle
    ps_oracle.lib()
print("probe", mode, "reached exit", flush=True)
