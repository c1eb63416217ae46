"""Where the host's time goes in one Pipeline.run: preparing the job records, issuing (caelo_pipeline_submit_many), the flush, and
the wait for the GPU.  `python tools/host_prep_probe.py [steps=20]`"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np
import torch
from caelo import synth, _ffi
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, Pipeline, FrameBatch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 8
eng = Engine()
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(2 * B + 1)]
rng = np.random.RandomState(7)
rand = [torch.from_numpy(rng.random_sample((1500, 4))).to(eng.device) for _ in pool]
pipe = Pipeline(eng, batch=B)
n = steps * B
def walk(i):   # 0 1 .. 16 15 .. 1 0 1 ..: consecutive frames stay neighbours (bench.py)
    i %= 2 * (len(pool) - 1)
    return i if i < len(pool) else 2 * (len(pool) - 1) - i
order = [walk(i) if os.environ.get("PROBE_ORDER", "walk") == "walk" else i % len(pool) for i in range(n)]
scans, rands = [pool[j] for j in order], [rand[j] for j in order]
out = FrameBatch(eng, n)
for _ in range(2):
    pipe.run(scans, rands, out=out)
    torch.cuda.synchronize()
lib = eng.lib
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    _ffi.check(lib.caelo_pipeline_expect(pipe.h, n))
    _ffi.check(lib.caelo_pipeline_begin(pipe.h, eng.stream))
    for pc in scans:
        assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] == 4 and pc.is_contiguous()
    jobs = pipe._jobs([pc.data_ptr() for pc in scans], [pc.shape[0] for pc in scans], rands, None, out, True, 5, False, True)
    t1 = time.perf_counter()
    _ffi.check(lib.caelo_pipeline_submit_many(pipe.h, jobs.ctypes.data, n))
    t2 = time.perf_counter()
    _ffi.check(lib.caelo_pipeline_flush(pipe.h, eng.stream))
    e1.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("%d frames: prepare %.0f us, submit %.0f us, flush %.0f us, wait %.0f us, total %.0f us = %.0f frames/s; on the GPU, event to event, %.0f us" %
          (n, 1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2), 1e6 * (t4 - t3), 1e6 * (t4 - t0), n / (t4 - t0), 1e3 * e0.elapsed_time(e1)))
