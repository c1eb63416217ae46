import os, sys, numpy as np
REPO="/root/repo"
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import torch, oracle as orc
from caelo import synth
from caelo.engine import Engine
eng=Engine(); dev=eng.device
pc=synth.make_scan(0)
o_ring,o_cnt=orc.ProjectPC2SphericalRing(pc)
ring,cnt,st=eng.project(torch.from_numpy(pc).to(dev))
r,c=ring.cpu().numpy(),cnt.cpu().numpy()
print("sum cnt", c.sum(), o_cnt.sum(), "nnz", (c>0).sum(), (o_cnt>0).sum(), "max", c.max(), o_cnt.max())
d=(c!=o_cnt)
print("rows with diffs", np.nonzero(d.any(axis=1))[0])
print("cols with diffs (first 20)", np.nonzero(d.any(axis=0))[0][:20], d.any(axis=0).sum())
ys,xs=np.nonzero(d); print("examples", [(y,x,c[y,x],o_cnt[y,x]) for y,x in zip(ys[:10],xs[:10])])
# compare per-point indices with numpy formulas
x,y,z=pc[:,0].astype(np.float64),pc[:,1].astype(np.float64),pc[:,2]
rr=np.sqrt((pc[:,0]*pc[:,0]+pc[:,1]*pc[:,1])+pc[:,2]*pc[:,2]).astype(np.float32)
d2r=np.pi/180; az=0.2*d2r; vd=-24.8*d2r; vu=2.0*d2r; vres=(vu-vd)/63; voff=-vd/vres
col=((np.pi-np.arctan2(y,x))/az).astype(np.int64)
val=np.arcsin((z/rr).astype(np.float64))/vres+voff
print("frac dist to integer: min", np.abs(val-np.round(val)).min(), "median", np.median(np.abs(val-np.round(val))))
t=torch.from_numpy(pc).to(dev).double()
tv=(torch.asin((torch.from_numpy(pc).to(dev)[:,2]/torch.from_numpy(rr).to(dev)).double())/vres+voff).cpu().numpy()
print("torch gpu asin vs numpy: max abs diff", np.abs(tv-val).max(), "int mismatch", (tv.astype(np.int64)!=val.astype(np.int64)).sum())
