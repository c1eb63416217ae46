"""Dense(200) on 192-row against 128-row tiles (CAELO_D1_TILE3_FROM=100000000 keeps every launch on 128): time and output hash.  tools/d1_tile_ab.sh"""
import os, sys, hashlib
sys.path.insert(0, "/root/repo/cae-lo_amd")
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
rs = np.random.RandomState(3)
for n in (12288, 24576):
    dense = rs.random_sample((n, 4096)) < 0.01
    bits = np.packbits(dense.reshape(n, 512, 8), axis=2, bitorder="little").reshape(n, 512).view(np.uint64)
    b = torch.from_numpy(np.ascontiguousarray(bits).view(np.int64)).to(eng.device)
    for _ in range(3): out, _ = eng.encode_profile(b, group=3)
    prof = np.array([eng.encode_profile(b, group=3)[1] for _ in range(10)])
    us = np.median(prof[:, :4], axis=0) * 1e3
    print("tiles=%s n=%d dense1 %.1f us head %.1f us  out sha %s" % (("128" if os.environ.get("CAELO_D1_TILE3_FROM") else "192 above 16384 rows"), n, us[2], us[3], hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]))
