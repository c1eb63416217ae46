"""How many columns of the NN match the f16 screen certifies alone, decides among 2..8 candidate rows exactly, or re-scans exactly
(caelo_match's statistics words), per pair of the 17-scan pool bench.py walks, and the single-pair call time of each."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth
from caelo.engine import Engine
eng = Engine()
scene = sys.argv[1] if len(sys.argv) > 1 else "boxes"
ff = [eng.extract(torch.from_numpy(synth.make_scan(i, quantum=1e-3, scene_kind=scene)).to(eng.device)) for i in range(17)]
tot = np.zeros(2, np.int64)
for a in range(16):
    b = a + 1
    ws = eng._ws("match1024", int(eng.lib.caelo_match_ws_bytes(1024)))
    ws[:256].zero_()
    idx = eng.match(ff[a].features, ff[b].features, ff[a].n_key, ff[b].n_key)
    torch.cuda.synchronize()
    st = ws[:8].view(torch.int32).cpu().numpy().copy()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.match(ff[a].features, ff[b].features, ff[a].n_key, ff[b].n_key)
    e1.record(); torch.cuda.synchronize()
    d = len(torch.unique(ff[a].features[:int(ff[a].n_key.item())], dim=0))
    tot += st[:2]
    print("frames %2d -> %2d: re-scanned exactly %3d, decided among 2..8 rows %3d, of %d columns; distinct descriptors in frame %d: %d; %.1f us per single-pair call" % (
        a, b, st[0], st[1], int(ff[b].n_key.item()), a, d, e0.elapsed_time(e1) / 20 * 1e3))
print("total: re-scanned %d, 2..8 candidates %d of %d columns" % (tot[0], tot[1], 16 * 1024))
