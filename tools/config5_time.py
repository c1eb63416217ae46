"""Times the BASELINE.json configs[4] stress case (128-beam x 4000-az scan, 32^3 patches) stage by stage on the GPU."""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cae-lo_amd"))
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import default_engine

e = default_engine()
pc = torch.from_numpy(synth.make_scan(0, n_beams=128, n_az=4000)).to(e.device)
ff = e.extract(pc)                       # 16^3 path for the key points
key_pts = ff.key_pts.contiguous()
vmap, st = e.voxelize(pc)
wd1, bd1 = e.seeded_dense1_32()
e.set_encoder32_dense(wd1, bd1)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

bits = e.patches32(vmap, key_pts)
out = {"points": int(pc.shape[0]), "patches32_ms": timed(lambda: e.patches32(vmap, key_pts)),
       "encode32_ms": timed(lambda: e.encode32(bits, group=3)), "voxelize_ms": timed(lambda: e.voxelize(pc, vmap)),
       "extract16_ms": timed(lambda: e.extract(pc))}
out["encode32_tflops"] = 194.1e9 / (out["encode32_ms"] * 1e-3) / 1e12
print(json.dumps(out))
