"""Is there throughput left in running MORE batches concurrently?  Two independent caelo_pipeline objects (each with its own
four streams, voxel maps, workspaces) fed from two host threads, against one pipeline doing the same frames.
    python tools/two_pipelines_probe.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, FrameBatch, ransac_draws
eng = Engine()
pcs = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(17)]
rnd = [torch.from_numpy(ransac_draws(i)).to(eng.device) for i in range(17)]
walk = list(range(1, 17)) + list(range(15, -1, -1))
n = 256
order = [walk[i % 32] for i in range(n)]
scans, draws = [pcs[i] for i in order], [rnd[i] for i in order]
prev = eng.extract(pcs[0])
pipes = [eng.pipeline(8, 3), eng.pipeline(8, 3)] if False else None
p1 = eng.pipeline(8, 3)
out1 = FrameBatch(eng, n)
def one(reps):
    for _ in range(reps):
        p1.run(scans, draws, prev=prev, out=out1)
one(2); torch.cuda.synchronize()
t = time.perf_counter(); one(4); torch.cuda.synchronize()
print("one pipeline : %.0f frames/s" % (4 * n / (time.perf_counter() - t)))
# two pipelines: separate engines' pipeline objects cannot share a cache key -> build the second by hand
from caelo.engine import Pipeline
p2 = Pipeline(eng, 8, 3)
outs = [FrameBatch(eng, n // 2), FrameBatch(eng, n // 2)]
streams = [torch.cuda.Stream(device=eng.device) for _ in range(2)]
def worker(k, reps):
    torch.cuda.set_device(eng.device)
    pipe = (p1, p2)[k]
    lo = k * (n // 2)
    with torch.cuda.stream(streams[k]):
        for _ in range(reps):
            pipe.run(scans[lo:lo + n // 2], draws[lo:lo + n // 2], prev=prev, out=outs[k])
def two(reps):
    th = [threading.Thread(target=worker, args=(k, reps)) for k in range(2)]
    [t_.start() for t_ in th]; [t_.join() for t_ in th]
two(2); torch.cuda.synchronize()
t = time.perf_counter(); two(4); torch.cuda.synchronize()
print("two pipelines: %.0f frames/s" % (4 * n / (time.perf_counter() - t)))
