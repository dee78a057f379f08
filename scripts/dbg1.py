import sys; sys.path.insert(0,'.')
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
from oracle.ps_oracle import Oracle
sc,p = scenes.cavity(32)
o=Oracle(); o.run(sc,p,solve=False)
g=polystokes_amd.Solver(0); g.upload(sc,p); g.setup()
for nm in ["reducedMassMatrices","reducedViscosityMatrices","reducedRegionBestFitVectors"]:
    a=g.array(nm); b=o.array(nm)
    R=o.nRegions
    a=a.reshape(R,-1); b=b.reshape(R,-1)
    for r in range(R):
        d=np.abs(a[r]-b[r])
        idx=np.argsort(-d)[:5]
        print(nm, r, "maxdiff", d.max(), [(int(i//26),int(i%26),float(a[r][i]),float(b[r][i])) for i in idx if d[i]>0][:4])
bb=g.array
