import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
from helpers import basis_rows
sc,p = scenes.cavity(32)
g=polystokes_amd.Solver(0); g.upload(sc,p); g.setup()
Mr=g.array("reducedMassMatrices").reshape(-1,26,26)[0]
com=g.array("reducedRegionCOM").reshape(-1,3)[0]
dx=sc.dx
parts=[]
for a in range(3):
    rng=[range(2,16)]*3
    rng=list(rng); rng[a]=range(2,17)
    pts=np.array([(i,j,k) for k in rng[2] for j in rng[1] for i in rng[0]],float)
    pts[:,a]-=0.5
    C=basis_rows(pts*dx-com, np.full(len(pts),a))
    parts.append(C.T@C)
tot=sum(parts)
print("full diff", np.abs(Mr-tot).max())
for a in range(3):
    rest=Mr-sum(parts[b] for b in range(3) if b!=a)
    print("axis",a,"gpu part vs expected", np.abs(rest-parts[a]).max())
    if np.abs(rest-parts[a]).max()>1e-6:
        d=np.abs(rest-parts[a]); idx=np.argwhere(d>1e-6)
        for (m,n) in idx[:30]: print("   ",m,n,rest[m,n],parts[a][m,n])
