"""The key point rule on NaN inputs: the oracle against the REFERENCE itself (GetKeyPtsByAE imported from /root/reference under
/opt/conda/bin/python3.9 with the stubs of SURVEY appendix C).  NaN in the response of any window pixel -> the pixel is no candidate
(cp.min over all 25 norms, SphericalRing.py:159,179); NaN intensity -> the five-channel range test of :197-198 is false.
    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tools/check_nan_semantics_vs_reference.py"""
import sys, types, numpy as np
sys.dont_write_bytecode = True
if not hasattr(np, "bool"): np.bool = bool
for n in ("mayavi", "mayavi.mlab"): sys.modules[n] = types.ModuleType(n)
sys.modules["mayavi"].mlab = sys.modules["mayavi.mlab"]
cp = types.ModuleType("cupy")
for k in ("array", "zeros", "min", "sum", "squeeze", "int32", "float32"): setattr(cp, k, getattr(np, k))
cp.bool = bool; cp.asnumpy = np.asarray
cp.argsort = lambda a: np.argsort(a, kind="stable")
sys.modules["cupy"] = cp
sys.path.insert(0, "/root/reference"); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/cae-lo_amd'); sys.path.insert(0,'/root/repo/oracle')
import SphericalRing as SR
import oracle as orc
from caelo import synth
models=orc.load_models('/root/repo/weights/SphericalRingPCRespondLayer.h5','/root/repo/weights/EncoderModel4VoxelPatch.h5')
pc=synth.make_scan(0)
ring,cnt=orc.ProjectPC2SphericalRing(pc)
resp=models[0].predict(ring[None,0:64,0:1792,0:3])[0]
k0=orc.GetKeyPtsByAE(ring,cnt,resp)[1]
r2=resp.copy()
for (r,c) in k0[::97][:8]:
    r2[r,c+1,3]=np.nan; r2[r-2,c,0]=np.nan
import warnings; warnings.simplefilter("ignore")
ref=SR.GetKeyPtsByAE(ring.copy(),cnt.copy(),r2.copy())
o=orc.GetKeyPtsByAE(ring,cnt,r2)
print("NaN in response: oracle == reference key pixels:", np.array_equal(np.asarray(ref[1]),o[1]), len(o[1]))
rg=ring.copy()
for (r,c) in k0[5::101][:8]: rg[r,c,3]=np.nan
ref=SR.GetKeyPtsByAE(rg.copy(),cnt.copy(),resp.copy()); o=orc.GetKeyPtsByAE(rg,cnt,resp)
print("NaN intensity: oracle == reference key pixels:", np.array_equal(np.asarray(ref[1]),o[1]), len(o[1]))
