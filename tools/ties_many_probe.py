"""Engine.resolve_ties_many / match_pose_exact_many on the clutter scene: time per call for several lane counts (first call of a lane
count creates the lanes' voxel maps), against the frame-by-frame Engine.resolve_ties."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, ransac_draws
eng = Engine(); eng.host_blas()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 96      # (scans 96.. of the circuit: about half of them split a tie class)
pcs = [torch.from_numpy(synth.make_scan(FIRST + f, quantum=1e-3, scene_kind="clutter", trajectory="circuit")).to(eng.device) for f in range(N)]
draws = [ransac_draws(f) for f in range(N)]; rnd = [torch.from_numpy(d).to(eng.device) for d in draws]
pipe = eng.pipeline(8)
def fresh():
    o = pipe.run(pcs, rnd, certify=True, rands_host=draws); torch.cuda.synchronize(); return o
def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t)
o = fresh()
_, ms = T(lambda: [eng.resolve_ties(o.frame(j), pcs[j]) for j in range(N)])
o = fresh()
r, ms = T(lambda: [eng.resolve_ties(o.frame(j), pcs[j]) for j in range(N)])
print("%d frames, %d tied: frame by frame %.1f ms" % (N, sum(1 for x in r if x), ms))
for lanes in [int(x) for x in os.environ.get("PROBE_LANES", "1,2,4,8,16,32").split(",")]:
    for rep in range(3):
        o = fresh()
        (tied, cnt), ms = T(lambda: eng.resolve_ties_many([(o.frame(j), pcs[j]) for j in range(N)], lanes=lanes))
        print("lanes %2d run %d: %d tied, %.1f ms  %s" % (lanes, rep, len(tied), ms, {k: round(v, 1) for k, v in eng.last_tie_times.items()}))
redo = sorted({t for u in tied for t in (u, u + 1) if 0 < t < N})
for rep in range(3):
    _, ms = T(lambda: eng.match_pose_exact_many([(o.frame(j - 1), o.frame(j)) for j in redo], [rnd[j] for j in redo], [draws[j] for j in redo]))
    print("match_pose_exact_many over %d pairs: %.1f ms" % (len(redo), ms))
_, ms = T(lambda: [eng.match_pose_exact(o.frame(j - 1), o.frame(j), rnd[j], draws[j]) for j in redo])
print("match_pose_exact pair by pair: %.1f ms" % ms)
