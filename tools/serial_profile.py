"""One stream, one frame at a time (Engine.extract + Engine.match_pose): undisturbed per-kernel times for
rocprofv3 --kernel-trace --stats."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, ransac_draws
eng = Engine()
pcs = [torch.from_numpy(synth.make_scan(i)).to(eng.device) for i in range(6)]
rnd = torch.from_numpy(ransac_draws(1)).to(eng.device)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
prev = eng.extract(pcs[5])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    cur = eng.extract(pcs[i % 6])
    eng.match_pose(prev, cur, rnd)
    prev = cur
torch.cuda.synchronize()
print("serial: %.1f us/frame (%d frames, includes Python issue time)" % ((time.perf_counter() - t0) / n * 1e6, n))
