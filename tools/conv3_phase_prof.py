"""Phase profile of k_enc_conv3 (needs a library built with `make -C cae-lo_amd/csrc PROF=1`): shader-clock cycles of thread 0 of
every workgroup, summed: split + LDS store / first barrier / MFMA phase (fragment loads + MFMAs + F3 stores) / second barrier."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pool = [torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device) for i in range(frames)]
bits = torch.cat([eng.patches(eng.voxelize(p)[0], eng.extract(p).key_pts.contiguous())[0].reshape(-1, 64) for p in pool], dim=0).contiguous()
buf = (C.c_ulonglong * 40)()
for _ in range(3):
    eng.encode_profile(bits, group=3)
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
before = np.array(buf[16:32], dtype=np.int64)
_, ms = eng.encode_profile(bits, group=3)
eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
d = np.array(buf[16:32], dtype=np.int64) - before
wgs = max(d[12], 1)
names = ["split + store", "barrier 1", "MFMA phase", "barrier 2 + loop"]
tot = d[8:12].sum()
print("conv3 %.1f us; %d workgroups, %d patches" % (ms[1] * 1e3, wgs, bits.numel() // 64))
for i in range(4):
    print("  %-18s %9.0f cycles/workgroup  %5.1f%%" % (names[i], d[8 + i] / wgs, 100.0 * d[8 + i] / tot))
