"""One group of eight tie-split clutter frames through the stages of Engine.resolve_ties_many (round 6: grouped), HIP events between them."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import synth, _ffi
from caelo.engine import Engine, MAX_K
eng = Engine()
FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 96
pcs, ffs = [], []
f = FIRST
while len(pcs) < 8:
    pc = torch.from_numpy(synth.make_scan(f, quantum=1e-3, scene_kind="clutter", trajectory="circuit")).to(eng.device)
    ff = eng.extract(pc)
    if bool((ff.flags & 2).any().item()):
        pcs.append(pc); ffs.append(ff)
    f += 1
print("frames", FIRST, "..", f - 1, ": 8 with tie-split patches:", [int(((ff.flags & 2) != 0).sum().item()) for ff in ffs])
n = 8
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(3):
    gbits = eng.empty((n, MAX_K, 3, 64), torch.int64); gflags = eng.empty((n, MAX_K, 3), torch.uint8); gpts = eng.empty((n, MAX_K, 3), torch.float32)
    torch.cuda.synchronize()
    e0 = ev()
    maps, sts = [], []
    for q in range(n):
        vm, st = eng.voxelize(pcs[q], eng.voxmap(max(eng.max_points, pcs[q].shape[0]), slot=10 + q)); maps.append(vm); sts.append(st)
    e1 = ev()
    for q in range(n):
        eng.voxmap_order(maps[q], 7)
        gpts[q].copy_(ffs[q].key_pts)
    e2 = ev()
    arr = lambda xs: (C.c_void_p * n)(*xs)
    _ffi.check(eng.lib.caelo_patches_many(eng.ctx, n, arr([m.h for m in maps]), arr([gpts[q].data_ptr() for q in range(n)]), MAX_K,
                                          arr([ffs[q].n_key.data_ptr() for q in range(n)]), arr([gbits[q].data_ptr() for q in range(n)]),
                                          arr([gflags[q].data_ptr() for q in range(n)]), arr([st.data_ptr() for st in sts]), eng.stream))
    e3 = ev()
    feats = eng.encode(gbits.reshape(-1, 64), group=3)
    e4 = ev()
    torch.cuda.synchronize()
    print("rep %d: voxelize x8 %.2f ms, order x8 (all scales) %.2f ms, patches_many (8 gathers + kd of 8 maps) %.2f ms, encode 24576 patches %.2f ms" % (
        rep, e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4)))
