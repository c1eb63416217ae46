"""What the exact RANSAC costs a run of the pipeline: no certificates / certificates only (kernels) / certificates + host half.
`python tools/cert_cost_probe.py [steps=120]` -> frames/s of three repetitions each, and the phases of a certified run."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np
import torch
import caelo; caelo.configure_runtime()
from caelo import synth, _ffi
from caelo.engine import Engine, Pipeline, FrameBatch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = 8
eng = Engine()
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(2 * B + 1)]
rng = np.random.RandomState(7)
rand_h = [rng.random_sample(6000) for _ in pool]
rand = [torch.from_numpy(r).to(eng.device) for r in rand_h]
pipe = Pipeline(eng, batch=B)
n = steps * B
def walk(i):
    i %= 2 * (len(pool) - 1)
    return i if i < len(pool) else 2 * (len(pool) - 1) - i
order = [walk(i) for i in range(n)]
scans, rands, rands_h = [pool[j] for j in order], [rand[j] for j in order], [rand_h[j] for j in order]
prev = eng.extract(pool[1])
out = FrameBatch(eng, n)
modes = [("off", False), ("device", "device"), ("host", True)]
if len(sys.argv) > 2:
    modes = [m for m in modes if m[0] in sys.argv[2].split(",")]
for name, mode in modes:
    for _ in range(2):
        pipe.run(scans, rands, prev=prev, out=out, certify=mode, rands_host=rands_h)
        torch.cuda.synchronize()
    fps = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run(scans, rands, prev=prev, out=out, certify=mode, rands_host=rands_h)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        fps.append(n / (t2 - t0))
    st = pipe.stats()
    print("%-7s %s frames/s   (last run: run() %.2f ms, sync %.2f ms; issue %.1f us/frame) %s" % (
        name, " ".join("%.0f" % f for f in fps), 1e3 * (t1 - t0), 1e3 * (t2 - t1), st["issue_us_per_frame"], (pipe.cert_stats() if mode is True else ""), ) + " " + str({k: round(v, 2) for k, v in pipe.last_times.items()}))
