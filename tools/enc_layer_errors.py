"""Per-layer error of the HIP encoder against the f32 CPU oracle (and of the oracle against an f64 evaluation of the same
network): max |difference| after pool2 (P2), after conv3 (F3), of the Dense(200) pre-activations and of the descriptors,
on every patch of the quantised golden frame.  `CAELO_ENC_S1=f32 python tools/enc_layer_errors.py` measures round 2's
f32-input stage 1 for comparison.  The budget table of tests/test_gpu_parity.py::test_encoder_layer_error_budget is
3 x what this prints for the default kernels."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import caelo; caelo.configure_runtime()
import oracle as orc
from caelo.engine import Engine

eng = Engine(device=0)
if os.environ.get("CAELO_ENC_S1") == "f32":   # (read HERE, by the tool: the library has no environment switch for arithmetic)
    eng.set_encoder_reference(True)
_, enc_m = orc.load_models(os.path.join(REPO, "weights", "SphericalRingPCRespondLayer.h5"), os.path.join(REPO, "weights", "EncoderModel4VoxelPatch.h5"))
bits = np.ascontiguousarray(np.load(os.path.join(REPO, "tests", "golden", "frame_q0.npz"))["patch_bits"].reshape(-1, 64))
o_p2, o_f3, o_h, o_out = enc_m.predict_layers(bits)
p2, f3, pre, out = eng.encode_layers(torch.from_numpy(bits.view(np.int64)).to(eng.device))
torch.cuda.synchronize()
h = np.tanh((pre.cpu().numpy().astype(np.float64) + enc_m.w[7].astype(np.float64)))
for name, a, b in (("P2", p2.cpu().numpy(), o_p2), ("F3", f3.cpu().numpy(), o_f3), ("tanh(Dense(200))", h, o_h), ("descriptors", out.cpu().numpy(), o_out)):
    d = np.abs(a.astype(np.float64) - b)
    print("%-18s max abs %.3e   mean abs %.3e   element-wise rel (floor 0.1) %.3e" % (name, d.max(), d.mean(), (d / np.maximum(np.abs(b), 0.1)).max()))
