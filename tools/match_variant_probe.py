"""The NN match's launch shape (8 pairs per launch) timed alone and the pipeline's rate, for one build of the library (CAELO_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, FrameBatch, ransac_draws
eng = Engine()
B = 8
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(2 * B + 1)]
ff = [eng.extract(p) for p in pool[:B + 1]]
ms = [eng.match_profile(ff, repeats=50)[:2] for _ in range(3)]
ref = [eng.match(ff[i].features, ff[i + 1].features, ff[i].n_key, ff[i + 1].n_key) for i in range(B)]
_, _, idx = eng.match_profile(ff, repeats=1)
same = all(torch.equal(idx[i], ref[i]) for i in range(B))
print("match: both %.1f us, prep %.1f us, screen %.1f us per 8 pairs; batched == single calls: %s" % (
    min(m[0] for m in ms) * 1e3, min(m[1] for m in ms) * 1e3, min(m[0] - m[1] for m in ms) * 1e3, same))
rand = [torch.from_numpy(ransac_draws(i)).to(eng.device) for i in range(len(pool))]
pipe = eng.pipeline(B)
n = 120 * B
def walk(i):
    i %= 2 * (len(pool) - 1)
    return i if i < len(pool) else 2 * (len(pool) - 1) - i
order = [walk(i) for i in range(n)]
scans, rands = [pool[j] for j in order], [rand[j] for j in order]
prev = eng.extract(pool[1]); out = FrameBatch(eng, n)
fps = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.run(scans, rands, prev=prev, out=out)
    torch.cuda.synchronize(); fps.append(n / (time.perf_counter() - t0))
print("pipeline (no certify): %s frames/s" % " ".join("%.0f" % f for f in fps[1:]))
