"""Time caelo_match on a set of 8 pairs (the pipeline's launch shape) with HIP events."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth, _ffi
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, ransac_draws
eng = Engine()
# 17 consecutive scans walked back and forth, 8 per batch: no batch holds a scan twice (equal patches are looked for across
# the frames of a batch; like bench.py)
pcs = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(17)]
rnd = [torch.from_numpy(ransac_draws(i)).to(eng.device) for i in range(17)]
pipe = eng.pipeline(8)
order = ([i for i in range(1, 17)] + [i for i in range(15, -1, -1)]) * 2
scans = [pcs[i] for i in order[:64]]; draws = [rnd[i] for i in order[:64]]
assert all(len(set(order[i:i + 8])) == 8 for i in range(0, 64, 8))
prev = eng.extract(pcs[0])
for _ in range(3):
    pipe.run(scans, draws, prev=prev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    pipe.run(scans, draws, prev=prev)
e1.record(); torch.cuda.synchronize()
print("pipeline: %.1f us/frame" % (e0.elapsed_time(e1) / 5 / 64 * 1e3))
