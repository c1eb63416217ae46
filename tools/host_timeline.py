"""Where the wall time of a short pipeline run goes on the host side (the driver's 20-step bench line): output allocation,
submission, waiting for the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, Pipeline, FrameBatch, ransac_draws

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine()
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(6)]
rnd = [torch.from_numpy(ransac_draws(i)).to(eng.device) for i in range(6)]
pipe = Pipeline(eng, 8, 3)
prev = eng.extract(pool[0])
walk = lambda i: (i % 10) if (i % 10) < 6 else 10 - (i % 10)   # 0 1 .. 5 4 .. 1 0: every pair is a pair of neighbouring scans
scans = [pool[walk(i + 1)] for i in range(K)]
draws = [rnd[walk(i + 1)] for i in range(K)]
pipe.run([pool[walk(i + 1)] for i in range(24)], [rnd[walk(i + 1)] for i in range(24)], prev=prev)
torch.cuda.synchronize()
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = FrameBatch(eng, K)
    t1 = time.perf_counter()
    pipe.run(scans, draws, prev=prev, out=out)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("K=%d  alloc %.0f us  submit %.0f us  wait %.0f us  total %.0f us  -> %.0f frames/s" % (
        K, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6, K / (t3 - t0)))
