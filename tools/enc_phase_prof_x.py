"""Phase profile of k_enc_stage1x (library built with `make -C cae-lo_amd/csrc PROF=1 BUILD=... OUT=tools/_variant_prof.so`):
shader-clock cycles wave 0 of every workgroup spends per phase, per patch, for a 1-frame and an 8-frame launch.
    CAELO_LIB=tools/_variant_prof.so python tools/enc_phase_prof_x.py"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
if os.environ.get("CAELO_ENC_S1") == "f32":   # (read HERE, by the tool: the library has no environment switch for arithmetic)
    eng.set_encoder_reference(True)
parts = []
DENS = os.environ.get("DENSITY")     # DENSITY=0 / 0.0005 ...: random patches of that share of set voxels instead of the scans' patches
if DENS is not None:
    rs = np.random.RandomState(3)
    for i in range(6):
        dense = rs.random_sample((3072, 4096)) < float(DENS)
        b = np.packbits(dense.reshape(3072, 512, 8), axis=2, bitorder="little").reshape(3072, 512).view(np.uint64)
        parts.append(torch.from_numpy(np.ascontiguousarray(b).view(np.int64)).to(eng.device))
for i in range(0 if DENS is not None else 6):
    pc = torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device)
    ff = eng.extract(pc)
    parts.append(eng.patches(eng.voxelize(pc)[0], ff.key_pts.contiguous())[0].reshape(-1, 64))
buf = (C.c_ulonglong * 40)()
names = ["wipe+masks+queue", "barrier 1", "conv1 (mfma)", "barrier 2", "end of patch (rest)", "barrier 3"]
for frames in (1, 8):
    bits = torch.cat([parts[i % 6] for i in range(frames)], dim=0).contiguous()
    for _ in range(3):
        eng.encode_profile(bits, group=3)
    eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
    before = np.array(buf[16:32], dtype=np.int64)
    _, ms = eng.encode_profile(bits, group=3)
    eng.lib.caelo_debug_read(C.cast(buf, C.c_void_p))
    d = np.array(buf[16:32], dtype=np.int64) - before
    tot = d[0:6].sum() + d[8:12].sum()
    print("%d frame(s): stage1 %.1f us; patches %d, queued cells/patch %.1f, cycles/patch (one workgroup) %.0f" % (frames, ms[0] * 1e3, d[6], d[7] / max(d[6], 1), tot / max(d[6], 1)))
    for i in range(6):
        print("  %-18s %8.0f cycles/patch  %5.1f%%" % (names[i], d[i] / max(d[6], 1), 100.0 * d[i] / tot))
    for nm, k in (("conv2 (pairs)", 8), ("wait: all memory back", 11), ("next rows to LDS", 9), ("P2 stores", 10)):
        print("  %-26s %8.0f cycles/patch  %5.1f%%" % (nm, d[k] / max(d[6], 1), 100.0 * d[k] / tot))
