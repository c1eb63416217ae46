"""Launch time of the four encoder kernels against the number of patches per launch (caelo_encode_profile, HIP events): the intercept
is what a launch costs before it has encoded anything (prologues, weight fetches, tails), the slope the per-patch cost.
    python tools/enc_fixed_cost_probe.py [set share=0.01]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
rs = np.random.RandomState(3)
print("%8s %10s %10s %10s %10s" % ("patches", "stage1 us", "conv3 us", "dense1 us", "head us"))
for n in (192, 768, 1536, 3072, 6144, 12288, 24576):
    dense = rs.random_sample((n, 4096)) < p
    bits = np.packbits(dense.reshape(n, 512, 8), axis=2, bitorder="little").reshape(n, 512).view(np.uint64)
    b = torch.from_numpy(np.ascontiguousarray(bits).view(np.int64)).to(eng.device)
    for _ in range(3):
        eng.encode_profile(b, group=3)
    prof = np.array([eng.encode_profile(b, group=3)[1] for _ in range(8)])
    us = np.median(prof[:, :4], axis=0) * 1e3
    print("%8d %10.1f %10.1f %10.1f %10.1f" % (n, us[0], us[1], us[2], us[3]))
