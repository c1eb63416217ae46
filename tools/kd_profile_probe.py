"""Per-level times of k_kd_build (a -DKD_PROFILE build prints them: CAELO_LIB=variants/libcaelo_kdprof.so) on clutter frames with tie-split
patches (one at the 64 cm scale, then ones at the finer scales)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine
eng = Engine()
done = 0
for f in range(64):
    pc = torch.from_numpy(synth.make_scan(f, quantum=1e-3, scene_kind="clutter")).to(eng.device)
    ff = eng.extract(pc)
    fl = ff.flags.cpu().numpy()
    per = [(fl[:, s] & 2).astype(bool).sum() for s in range(3)]
    if sum(per) == 0 or (per[0] == 0 and per[1] == 0 and done >= 1):
        continue
    print("frame %d: tie-split patches per scale %s" % (f, per), flush=True)
    eng.resolve_ties(ff, pc); torch.cuda.synchronize()
    done += 1
    if done >= 4:
        break
