"""Encoder kernel times when all 3072 patches come from one scale (fixed vs data-dependent cost)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
pc = torch.from_numpy(synth.make_scan(0)).to(eng.device)
ff = eng.extract(pc)
bits, _ = eng.patches(eng.voxelize(pc)[0], ff.key_pts.contiguous())
bt = bits.reshape(-1, 3, 64)
def t(b, label):
    b = b.contiguous()
    for _ in range(3): eng.encode_profile(b, group=1)
    ms = np.mean([eng.encode_profile(b, group=1)[1] for _ in range(10)], axis=0)
    print("%-10s stage1 %.1f  conv3 %.1f  dense1 %.1f  head %.1f us" % ((label,) + tuple(ms * 1e3)))
t(bits.reshape(-1, 64), "mixed")
for s in range(3):
    t(bt[:, s].repeat(3, 1), "scale %d" % s)
t(torch.zeros_like(bits.reshape(-1, 64)), "empty")
