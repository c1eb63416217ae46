"""stage1_sparse_sweep.py -- stage 1 of the encoder (k_enc_stage1s + k_enc_stage1x) against the sparse threshold
(caelo_set_encoder_sparse: patches with at most that many non-background cells go to the wavefront-per-patch kernel), HIP events
around the two launches (caelo_encode_profile), 8 frames = 24 576 patches per launch, every patch (no de-duplication), and the
de-duplicated pipeline rate at 60 batches per threshold.  The thresholds change scheduling only (tests: same bits)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, Pipeline, ransac_draws
import time

eng = Engine()
traj = os.environ.get("CAELO_SWEEP_TRAJECTORY", "circuit")
pool = [torch.from_numpy(synth.make_scan(300 + i, quantum=1e-3, trajectory=traj)).to(eng.device) for i in range(8)]
bits = [eng.patches(eng.voxelize(p)[0], eng.extract(p).key_pts.contiguous())[0] for p in pool]
b = torch.cat([x.reshape(-1, 64) for x in bits], dim=0).contiguous()
ths = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 4, 8, 16, 24, 32, 48, 64]
pipe_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 60
scans = [pool[i % 8] for i in range(8 * pipe_batches)]
rnd = [torch.from_numpy(ransac_draws(i)).to(eng.device) for i in range(8)]
rnds = [rnd[i % 8] for i in range(len(scans))]
pipe = Pipeline(eng, 8, 3)
print("threshold  stage1 us (24576 patches)  executed MFMA M  conv3 dense1 head us | pipeline frames/s (%d batches, de-duplicated, no host half)" % pipe_batches)
for th in ths:
    eng.set_encoder_sparse(th)
    for _ in range(3):
        eng.encode_profile(b, group=3)
    prof = np.array([eng.encode_profile(b, group=3)[1] for _ in range(20)])
    ms = prof[:, 0:4].mean(axis=0)
    rate = 0.0
    if pipe_batches > 0:
        pipe.run(scans[:64], rnds[:64]); torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); pipe.run(scans, rnds); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = max(best, len(scans) / dt)
        rate = best
    print("%9d  %8.1f (min %.1f)  %10.2f  %6.1f %6.1f %6.1f | %8.0f" % (th, 1e3 * ms[0], 1e3 * prof[:, 0].min(), prof[:, 4].mean(), 1e3 * ms[1], 1e3 * ms[2], 1e3 * ms[3], rate), flush=True)
eng.set_encoder_sparse(32)
