"""Phase profile of k_enc_conv3 (library built with `make -C cae-lo_amd/csrc C3PROF=1 BUILD=... OUT=tools/_variant_c3prof.so`):
shader-clock cycles wave 0 of every workgroup spends per phase, per pair of patches, for an 8-frame launch.
    CAELO_LIB=tools/_variant_c3prof.so python tools/enc_phase_prof_c3.py"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
parts = []
for i in range(6):
    pc = torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device)
    ff = eng.extract(pc)
    parts.append(eng.patches(eng.voxelize(pc)[0], ff.key_pts.contiguous())[0].reshape(-1, 64))
buf = (C.c_ulonglong * 40)()
names = ["split the staged rows into LDS", "barrier 1", "matrix work + F3 stores", "barrier 2 + loop"]
for frames in (1, 8):
    bits = torch.cat([parts[i % 6] for i in range(frames)], dim=0).contiguous()
    for _ in range(3):
        eng.encode_profile(bits, group=3)
    eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
    before = np.array(buf[16:32], dtype=np.int64)
    _, ms = eng.encode_profile(bits, group=3)
    eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
    d = np.array(buf[16:32], dtype=np.int64) - before
    wgs, pairs = int(d[12]), bits.shape[0] // 2
    tot = d[8:12].sum()
    print("%d frame(s): conv3 %.1f us; %d workgroups, %d pairs of patches, cycles per pair (one workgroup) %.0f" % (frames, ms[1] * 1e3, wgs, pairs, tot / pairs))
    for i in range(4):
        print("  %-34s %8.0f cycles/pair  %5.1f%%" % (names[i], d[8 + i] / pairs, 100.0 * d[8 + i] / tot))
