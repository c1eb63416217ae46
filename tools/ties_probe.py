"""Where Engine.resolve_ties spends its time on the clutter scene (frames whose 496-nearest cut splits a tie class)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine, raise_status
eng = Engine()
def T(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("   %-58s %8.1f us" % (label, 1e6 * (time.perf_counter() - t))); return r
n_done = 0
for f in range(40):
    pc = torch.from_numpy(synth.make_scan(f, quantum=1e-3, scene_kind="clutter")).to(eng.device)
    ff = eng.extract(pc)
    fl = ff.flags.cpu().numpy()
    if not (fl & 2).any():
        continue
    print("frame %d: tie-split patches per scale %s" % (f, [(fl[:, s] & 2).astype(bool).sum() for s in range(3)]))
    for rep in range(2):
        cap = max(eng.max_points, pc.shape[0])
        vm, st = T("voxelize (exact, first touch)", lambda: eng.voxelize(pc, eng.voxmap(cap, slot=2)))
        mask = sum(1 << s for s in range(3) if (fl[:, s] & 2).any())
        T("voxmap_order (scales %s: compaction + padded sort)" % [s for s in range(3) if mask >> s & 1], lambda: eng.voxmap_order(vm, mask))
        vm2 = vm
        k = int(ff.n_key.item())
        kp = ff.key_pts[:k].contiguous()
        bits, flags = T("patches (+ kd collect/build/query)", lambda: eng.patches(vm2, kp))
        bits, flags = T("patches again (tree built)", lambda: eng.patches(vm2, kp))
        T("encode", lambda: eng.encode(bits.reshape(-1, 64), group=3))
    n_done += 1
    if n_done >= 3:
        break
