"""Frame rate of the batched pipeline with every batch of scans uploaded from pinned host memory on a copy stream
(Pipeline.run_uploading) beside the same frames resident in HBM (Pipeline.run).  Written for profiles/r03_upload_overlap.txt."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, FrameBatch, ransac_draws
eng = Engine()
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)) for i in range(17)]   # 2 x 8 + 1: no batch holds a scan twice (like bench.py)
host = [p.pin_memory() for p in pool]
dev = [p.to(eng.device) for p in pool]
rnd = [torch.from_numpy(ransac_draws(i)).to(eng.device) for i in range(17)]
pipe = eng.pipeline(8, 3)
n = 256
walk = list(range(1, 17)) + list(range(15, -1, -1))
order = [walk[i % 32] for i in range(n)]
assert all(len(set(order[i:i + 8])) == 8 for i in range(0, n, 8))
prev = eng.extract(dev[0])
out = FrameBatch(eng, n)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t0)
print("resident      %.0f frames/s" % t(lambda: pipe.run([dev[j] for j in order], [rnd[j] for j in order], prev=prev, out=out)))
print("run_uploading %.0f frames/s" % t(lambda: pipe.run_uploading([host[j] for j in order], [rnd[j] for j in order], prev=prev, out=out)))
# the same protocol with device-to-device copies instead of PCIe ones: what the batch-by-batch submission itself costs
print("run_uploading, sources in HBM %.0f frames/s" % t(lambda: pipe.run_uploading([dev[j] for j in order], [rnd[j] for j in order], prev=prev, out=out)))
# uploads alone
copy = torch.cuda.Stream()
bufs = [torch.empty((130000, 4), device=eng.device) for _ in range(8)]
def up():
    with torch.cuda.stream(copy):
        for i, j in enumerate(order):
            bufs[i % 8][:host[j].shape[0]].copy_(host[j], non_blocking=True)
    torch.cuda.current_stream().wait_stream(copy)
print("uploads alone %.0f frames/s (%.1f GB/s)" % (t(up), t(up) * host[0].numel() * 4 / 1e9))
# the two together with NO dependency between them (copies into buffers nobody reads): what the hardware overlaps
def both():
    with torch.cuda.stream(copy):
        for i, j in enumerate(order):
            bufs[i % 8][:host[j].shape[0]].copy_(host[j], non_blocking=True)
    pipe.run([dev[j] for j in order], [rnd[j] for j in order], prev=prev, out=out)
    torch.cuda.current_stream().wait_stream(copy)
print("resident run + unrelated uploads beside it %.0f frames/s" % t(both))
# host side: how long the calling thread needs to ISSUE a run (returns before the GPU is done)
for name, f in (("run", lambda: pipe.run([dev[j] for j in order], [rnd[j] for j in order], prev=prev, out=out)),
                ("run_uploading", lambda: pipe.run_uploading([host[j] for j in order], [rnd[j] for j in order], prev=prev, out=out))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-14s host issue %.0f us per batch of 8, GPU done after %.0f us per batch" % (name, (t1 - t0) / (n / 8) * 1e6, (t2 - t0) / (n / 8) * 1e6))
# how long the calling thread waits for a batch's scans to arrive (Event.synchronize inside run_uploading)
_orig_sync = torch.cuda.Event.synchronize
waits = []
def _timed_sync(self):
    t0 = time.perf_counter(); _orig_sync(self); waits.append(time.perf_counter() - t0)
torch.cuda.Event.synchronize = _timed_sync
try:
    torch.cuda.synchronize()
    pipe.run_uploading([host[j] for j in order], [rnd[j] for j in order], prev=prev, out=out)
    torch.cuda.synchronize()
finally:
    torch.cuda.Event.synchronize = _orig_sync
print("run_uploading: the calling thread waited for arrivals %.0f us per batch (max %.0f)" % (1e6 * sum(waits) / max(len(waits), 1), 1e6 * max(waits)))
