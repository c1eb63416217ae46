"""Phase profile of k_enc_stage1w (library built with `make -C cae-lo_amd/csrc PROF=1`): 100 MHz ticks per phase summed over
all wavefronts and patches of one frame."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth, _ffi
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
pc = torch.from_numpy(synth.make_scan(0, quantum=1e-3)).to(eng.device)
ff = eng.extract(pc)
bits, _ = eng.patches(eng.voxelize(pc)[0], ff.key_pts.contiguous())
if len(sys.argv) > 1:      # one scale only (or "empty")
    bits = (torch.zeros_like(bits) if sys.argv[1] == "empty" else
            bits.reshape(-1, 3, 64)[:, int(sys.argv[1])].repeat(1, 3).reshape(-1, 3, 64).contiguous())
torch.cuda.synchronize()
buf = (C.c_ulonglong * 40)()
for _ in range(3):
    eng.encode_profile(bits, group=3)
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
before = np.array(buf[16:32], dtype=np.int64)
_, ms = eng.encode_profile(bits, group=3)
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
d = np.array(buf[16:32], dtype=np.int64) - before
names = ["scatter+fetch", "queue", "conv1", "conv2", "wipe/cleanup", None, None, "loop head"]
print("stage1 %.1f us; patches %d, queued cells/patch %.1f" % (ms[0] * 1e3, d[5], d[6] / max(d[5], 1)))
tot = sum(d[i] for i in (0, 1, 2, 3, 4, 7))
for i in (0, 1, 2, 3, 4, 7):
    print("  %-14s %8.2f us/patch  %5.1f%%" % (names[i], d[i] / max(d[5], 1) / 100.0, 100.0 * d[i] / tot))
print("  total          %8.2f us/patch" % (tot / max(d[5], 1) / 100.0))
