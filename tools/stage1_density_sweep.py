"""Stage 1 (k_enc_stage1x) against the DENSITY of the patches: 24 576 random patches per launch with a given share of set voxels,
from empty to full -- launch time (HIP events), MFMA instructions executed per patch, share of the dense conv2 tap rows.  The two
synthetic scenes sit at 0.05 % / 1.3 % / 1.6 % (boxes) and 0.06 % / 1.0 % / 3.6 % (clutter) set voxels at the three scales.
    python tools/stage1_density_sweep.py [share ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
if os.environ.get("CAELO_ENC_S1") == "f32":   # (read HERE, by the tool: the library has no environment switch for arithmetic)
    eng.set_encoder_reference(True)
n = int(os.environ.get("S1X_N", "24576"))   # patches per launch
rs = np.random.RandomState(3)
print("%-10s %10s %14s" % ("set share", "launch us", "MFMAs / patch"))
shares = [float(a) for a in sys.argv[1:]] or [0.0, 0.0005, 0.002, 0.01, 0.03, 0.1, 0.3, 1.0]
for p in shares:
    if p >= 1.0:
        bits = np.full((n, 64), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    else:
        dense = rs.random_sample((n, 4096)) < p
        bits = np.packbits(dense.reshape(n, 512, 8), axis=2, bitorder="little").reshape(n, 512).view(np.uint64)
    b = torch.from_numpy(np.ascontiguousarray(bits).view(np.int64)).to(eng.device)
    for _ in range(2):
        eng.encode_profile(b, group=3)
    prof = np.array([eng.encode_profile(b, group=3)[1] for _ in range(6)])
    us = prof[:, 0].mean() * 1e3
    mfma = prof[:, 4].mean() * 1e6 / n
    # cells with a set voxel in their 4^3 field -> conv1 tiles of 16; the rest of the MFMAs are conv2 tap rows (3 each, 864 dense)
    print("%-10.4f %10.1f %14.1f" % (p, us, mfma))
