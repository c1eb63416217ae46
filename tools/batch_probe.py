"""Pipeline rate by batch size (frames per launch), exact RANSAC included: `CAELO_LIB=... python tools/batch_probe.py 8 10 [steps]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, Pipeline, FrameBatch, ransac_draws
eng = Engine()
batches = [int(a) for a in sys.argv[1:3]] if len(sys.argv) > 2 else [8]
frames_total = int(sys.argv[3]) if len(sys.argv) > 3 else 960
for B in batches:
    P = 2 * B + 1
    pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(P)]
    rh = [ransac_draws(i) for i in range(P)]
    rd = [torch.from_numpy(r).to(eng.device) for r in rh]
    pipe = Pipeline(eng, batch=B)
    def walk(i):
        i %= 2 * (P - 1)
        return i if i < P else 2 * (P - 1) - i
    for n in (frames_total // B * B, 20 * B):
        order = [walk(i + 1) for i in range(n)]
        assert all(len(set(order[i:i + B])) == B for i in range(0, n, B))
        scans, rands, rands_h = [pool[j] for j in order], [rd[j] for j in order], [rh[j] for j in order]
        prev = eng.extract(pool[0]); out = FrameBatch(eng, n)
        fps = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.run(scans, rands, prev=prev, out=out, certify=True, rands_host=rands_h, publish=False)
            torch.cuda.synchronize(); fps.append(n / (time.perf_counter() - t0))
        print("batch %2d, %4d frames (%d steps): %s frames/s" % (B, n, n // B, " ".join("%.0f" % f for f in fps[1:])))
