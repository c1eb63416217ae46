"""How many columns of a descriptor match the f16 screen certifies alone, decides between two rows exactly, re-scans exactly
(caelo_match's statistics words), and the measured error of the screen would need a debug build -- see tests.  Frames 0..3."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine
eng = Engine()
ff = [eng.extract(torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device)) for i in range(4)]
for a, b in ((0, 1), (1, 2), (2, 3), (0, 0)):
    ws = eng._ws("match1024", int(eng.lib.caelo_match_ws_bytes(1024)))
    ws[:256].zero_()
    idx = eng.match(ff[a].features, ff[b].features, ff[a].n_key, ff[b].n_key)
    torch.cuda.synchronize()
    st = ws[:8].view(torch.int32).cpu().numpy()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.match(ff[a].features, ff[b].features, ff[a].n_key, ff[b].n_key)
    e1.record(); torch.cuda.synchronize()
    print("frames %d -> %d: columns re-scanned exactly %d, decided between two rows %d, of %d; %.1f us per single-pair call" % (a, b, st[0], st[1], int(ff[b].n_key.item()), e0.elapsed_time(e1) / 20 * 1e3))
