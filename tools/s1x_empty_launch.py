"""N launches of the encoder on 24 576 EMPTY patches (for rocprofv3 --pmc: what an item costs stage 1 when there is nothing to compute)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
b = torch.zeros((24576, 64), dtype=torch.int64, device=eng.device)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.encode_profile(b, group=3)
torch.cuda.synchronize()
