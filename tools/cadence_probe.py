"""Cadence of the pipeline inside ONE run: the time at which each batch's rows are encoded (an event on a side stream behind
caelo_pipeline_wait_encoded), averaged over groups of batches.   python tools/cadence_probe.py [steps=120] [group=10]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, FrameBatch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
group = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = 8
eng = Engine()
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(2 * B + 1)]
rng = np.random.RandomState(7)
rand = [torch.from_numpy(rng.random_sample((1500, 4))).to(eng.device) for _ in pool]
pipe = eng.pipeline(B, 3)
n = steps * B


def walk(i):
    i %= 2 * (len(pool) - 1)
    return i if i < len(pool) else 2 * (len(pool) - 1) - i


order = [walk(i) for i in range(n)]
scans, rands = [pool[j] for j in order], [rand[j] for j in order]
out = FrameBatch(eng, n)
side = torch.cuda.Stream(device=eng.device)
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record(side)
    k = [0]

    mode = os.environ.get("PROBE_MODE", "timing")   # none: per-batch submits only; wait: + wait_encoded; plain: + an untimed event

    def encoded(lo, hi):
        k[0] += 1
        if mode == "none":
            return
        if mode.startswith("host"):   # host1 / host2: host-paced, `lag` batches behind; then what a chunk's collective would enqueue
            pipe.sync_encoded(int(mode[4:]))
            torch.cuda.Event().record(side)
            return
        pipe.wait_encoded(side)
        if mode == "timing":
            ev[k[0]].record(side)
        elif mode == "plain":
            torch.cuda.Event().record(side)

    pipe.run(scans, rands, out=out, on_batch=encoded)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if mode != "timing":
        print("%d steps, mode %s: %.0f frames/s" % (steps, mode, n / dt))
        continue
    t = [ev[0].elapsed_time(e) * 1e3 for e in ev[1:]]
    line = " ".join("%.0f" % ((t[min(i + group, steps) - 1] - (t[i - 1] if i else 0.0)) / (min(i + group, steps) - i)) for i in range(0, steps, group))
    print("%d steps, %.0f frames/s; first batch encoded at %.0f us; us per batch by groups of %d: %s" % (steps, n / dt, t[0], group, line))
