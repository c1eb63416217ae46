"""Would two encoder launch sets overlap usefully?  Stage 1 is latency bound (matrix pipe 19 % busy), conv3 / Dense(200) are pipe
bound: two streams, each running the four encoder kernels on its own 24 576 patches, against the same work on one stream.
CAELO_S1X_SLOTS (grid of the persistent stage 1) decides whether two stage-1 grids can be resident at once.
    CAELO_S1X_SLOTS=256 python tools/enc_overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "frame_q0.npz"))["patch_bits"].reshape(-1, 64)
bits = torch.from_numpy(np.ascontiguousarray(np.tile(g, (8, 1))).view(np.int64)).to(eng.device)
s = [torch.cuda.Stream(device=eng.device) for _ in range(2)]
def run(two, reps=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        for k in range(2):
            with torch.cuda.stream(s[k if two else 0]):
                eng.encode(bits, group=3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6
for two in (False, True):
    run(two, 3)
    print("CAELO_S1X_SLOTS=%s  %s: %.1f us per two launch sets (2 x 24576 patches)" % (os.environ.get("CAELO_S1X_SLOTS", "default"),
          "two streams" if two else "one stream ", run(two)))
