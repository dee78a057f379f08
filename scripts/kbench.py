import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n=int(sys.argv[1]) if len(sys.argv)>1 else 256
sc,p=scenes.cavity(n, precond=abi.PRE_DIAGONAL)
s=polystokes_amd.Solver(0); s.upload(sc,p); s.setup()
for k in (sys.argv[2].split(",") if len(sys.argv)>2 else ("spmv_S","spmv_St","apply")):
    ms,by=s.bench_kernel(k,30)
    print(k, "ms %.4f"%ms, "GB/s %.0f"%(by/ms/1e6))
