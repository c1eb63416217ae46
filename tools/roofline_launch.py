"""The launch bench.py's `roofline` object times -- caelo_encode_profile on the 3072 patches of one frame -- repeated, for
rocprofv3 --kernel-trace / --pmc passes (profiles/r02_*): same input, same kernels, one frame per launch."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import torch
from caelo import synth
from caelo.engine import Engine
eng = Engine()
pc = torch.from_numpy(synth.make_scan(0, quantum=1e-3)).to(eng.device)
bits, _ = eng.patches(eng.voxelize(pc)[0], eng.extract(pc).key_pts.contiguous())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for _ in range(n):
    eng.encode_profile(bits, group=3)
torch.cuda.synchronize()
print("done", n)
