"""Phase profile of k_enc_stage1 (needs a library built with `make -C cae-lo_amd/csrc PROF=1`):
shader-clock cycles per phase summed over all patches of one frame."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
pc = torch.from_numpy(synth.make_scan(0)).to(eng.device)
ff = eng.extract(pc)
bits, _ = eng.patches(eng.voxelize(pc)[0], ff.key_pts.contiguous())
if len(sys.argv) > 1:      # one scale only (or "empty")
    bits = (torch.zeros_like(bits) if sys.argv[1] == "empty" else
            bits.reshape(-1, 3, 64)[:, int(sys.argv[1])].repeat(1, 3).reshape(-1, 3, 64).contiguous())
torch.cuda.synchronize()
buf = (C.c_ulonglong * 40)()
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
before = np.array(buf[16:32], dtype=np.int64)
for _ in range(3):
    eng.encode_profile(bits, group=3)
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
before = np.array(buf[16:32], dtype=np.int64)
_, ms = eng.encode_profile(bits, group=3)
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
d = np.array(buf[16:32], dtype=np.int64) - before
names = ["loop", "B1 masks (gather)", "B1 queue", "B2 conv1+pool", "conv2 mfma", "cleanup"]
tot = d[0:6].sum()
print("stage1 %.1f us; patches %d, queued cells/patch %.1f" % (ms[0] * 1e3, d[6], d[7] / max(d[6], 1)))
for i in range(0, 6):
    print("  %-14s %8.0f cycles/patch  %5.1f%%" % (names[i], d[i] / max(d[6], 1), 100.0 * d[i] / tot))
bt = bits.reshape(-1, 3, 64)
for s in range(3):
    b = bt[:, s].cpu().numpy().view(np.uint8)
    print("scale %d: set voxels/patch mean %.1f" % (s, np.unpackbits(b, axis=1).sum(1).mean()))
