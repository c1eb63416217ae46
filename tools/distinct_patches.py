"""Distinct bit-packed patches per batch of 8 consecutive scans of tools/match_time.py's pool (what the de-duplicated encoder launches
work on): prints the mean over the batches."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
from caelo.engine import Engine
eng = Engine()
pcs = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(17)]
bits = [eng.patches(eng.voxelize(p)[0], eng.extract(p).key_pts.contiguous())[0].reshape(-1, 64) for p in pcs]
order = ([i for i in range(1, 17)] + [i for i in range(15, -1, -1)]) * 2
n = [len(torch.unique(torch.cat([bits[j] for j in order[b:b + 8]]), dim=0)) for b in range(0, 64, 8)]
print("%.0f" % np.mean(n))
