"""Does PCIe traffic itself slow the resident pipeline?  The pipeline over resident scans (120 batches) alone, and with an unrelated
stream copying 2 MB pinned buffers to the device back to back for the whole run (no dependency between the two).  If the second figure
stays at the first, what the upload mode loses (15.5 k against 20 k frames/s) is the hand-over protocol, not the copies."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, FrameBatch, ransac_draws
eng = Engine(); eng.host_blas()
B, NB = 8, 120
pool = [torch.from_numpy(synth.make_scan(f, quantum=1e-3)).to(eng.device) for f in range(17)]
draws = [ransac_draws(f) for f in range(17)]; rnd = [torch.from_numpy(d).to(eng.device) for d in draws]
def walk(i):          # back and forth: consecutive frames stay neighbours (a 16 -> 0 jump escalates RANSAC on the host)
    i %= 32
    return i if i < 17 else 32 - i
order = [walk(i) for i in range(B * NB)]
pcs = [pool[i] for i in order]; rn = [rnd[i] for i in order]; rh = [draws[i] for i in order]
pipe = eng.pipeline(B)
out = FrameBatch(eng, B * NB)
prev = eng.extract(pool[1])
host = [torch.empty((131072, 4), dtype=torch.float32).pin_memory() for _ in range(8)]
dst = [torch.empty((131072, 4), dtype=torch.float32, device=eng.device) for _ in range(8)]
side = torch.cuda.Stream(eng.device)
def run():
    torch.cuda.synchronize(); t = time.perf_counter()
    pipe.run(pcs, rn, prev=prev, out=out, certify=True, rands_host=rh, publish=False)
    torch.cuda.synchronize(); return B * NB / (time.perf_counter() - t)
def measure(label, body):
    """4 runs of the pipeline while `body()` loops in another thread"""
    global stop, n_copies
    stop, n_copies = False, 0
    th = threading.Thread(target=body); th.start()
    time.sleep(0.05)
    t0 = time.perf_counter(); c0 = n_copies
    res = [round(run()) for _ in range(4)]
    dt = time.perf_counter() - t0; c1 = n_copies
    stop = True; th.join()
    print("%-44s: %s frames/s   (%.1f GB/s of copies meanwhile)" % (label, res, (c1 - c0) * 2.097152e6 / dt / 1e9))

def copier(srcs, pause=0.0):
    def body():
        global n_copies
        with torch.cuda.stream(side):
            while not stop:
                for i in range(8):
                    dst[i].copy_(srcs[i], non_blocking=True)
                n_copies += 8
                side.synchronize()          # keep the queue short: 8 copies (16 MB) in flight at most
                if pause:
                    time.sleep(pause)
    return body

def spinner():
    while not stop:
        side.synchronize()
        time.sleep(0.0002)

stop = False
n_copies = 0
run(); run()
print("resident alone                              : %s frames/s" % [round(run()) for _ in range(4)])
measure("a thread that only synchronises a stream", spinner)
measure("unrelated H2D copies, back to back", copier(host))
measure("unrelated H2D copies, 16 MB every ~1 ms", copier(host, 0.0006))
big_h = torch.empty((8 * 131072, 4), dtype=torch.float32).pin_memory()
big_d = torch.empty((8 * 131072, 4), dtype=torch.float32, device=eng.device)
def one_big():
    global n_copies
    with torch.cuda.stream(side):
        while not stop:
            big_d.copy_(big_h, non_blocking=True)
            n_copies += 8
            side.synchronize()
measure("the same bytes as ONE 16 MB copy per step", one_big)
def one_big_paced():
    global n_copies
    with torch.cuda.stream(side):
        while not stop:
            big_d.copy_(big_h, non_blocking=True)
            n_copies += 8
            side.synchronize()
            time.sleep(0.0001)
measure("ONE 16 MB copy per step, 0.1 ms pause", one_big_paced)
dsrc = [torch.empty((131072, 4), dtype=torch.float32, device=eng.device) for _ in range(8)]
measure("unrelated device-to-device copies", copier(dsrc))
