"""Time caelo_match / caelo_ransac alone (one pair per call) with HIP events."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, ransac_draws
eng = Engine()
pcs = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(2)]
rnd = torch.from_numpy(ransac_draws(1)).to(eng.device)
fa, fb = eng.extract(pcs[0]), eng.extract(pcs[1])
idx = eng.match(fa.features, fb.features, fa.n_key, fb.n_key)
res, mask = eng.ransac(fa.key_pts, fb.key_pts, idx, rnd, fb.n_key)
r = eng.pose_result(res)
print("result: success %d thr %.2f iters %d inliers %d best %d" % (r.success, r.threshold, r.iterations, r.n_inliers, r.best_trial))
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("match  : %.1f us/call" % timeit(lambda: eng.match(fa.features, fb.features, fa.n_key, fb.n_key)))
print("ransac : %.1f us/call" % timeit(lambda: eng.ransac(fa.key_pts, fb.key_pts, idx, rnd, fb.n_key)))
