"""Encoder kernels of one launch set timed with HIP events (the numbers bench.py's roofline.encoder_kernels quotes), for 1 and 8
frames per launch.  `CAELO_LIB=... python tools/enc_table.py` times an experiment build (results are not checked here)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine

eng = Engine()
if os.environ.get("CAELO_ENC_S1") == "f32":   # (read HERE, by the tool: the library has no environment switch for arithmetic)
    eng.set_encoder_reference(True)
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(8)]
bits = [eng.patches(eng.voxelize(p)[0], eng.extract(p).key_pts.contiguous())[0] for p in pool]
for frames in (1, 8):
    b = torch.cat([bits[i].reshape(-1, 64) for i in range(frames)], dim=0).contiguous()
    for _ in range(3):
        eng.encode_profile(b, group=3)
    prof = np.array([eng.encode_profile(b, group=3)[1] for _ in range(20)])
    ms = prof[:, 0:4].mean(axis=0)
    print("%d frame(s): stage1 %.1f  conv3 %.1f  dense1 %.1f  head %.1f us   total %.1f" % (frames, *(1e3 * ms), 1e3 * ms.sum()))
