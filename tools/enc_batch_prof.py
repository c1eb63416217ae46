"""Encoder kernel times vs batch size (patches of 1, 2, 3 frames in one launch): amortisation of the fixed costs."""
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
for k in (1, 2, 4, 8):
    b = bits.repeat(k, 1, 1).contiguous()
    for _ in range(3): eng.encode_profile(b, group=3)
    ms = np.mean([eng.encode_profile(b, group=3)[1] for _ in range(10)], axis=0) * 1e3
    print("frames/launch %d: stage1 %.1f conv3 %.1f dense1 %.1f head %.1f  total %.1f us = %.1f us/frame" % (
        k, ms[0], ms[1], ms[2], ms[3], ms.sum(), ms.sum() / k))
