"""The launches bench.py's `roofline` object times -- caelo_encode_profile on the patches of `frames` frames (1: round 1's figure,
8: the pipeline's launch shape and bench.py's headline) -- repeated, for rocprofv3 --kernel-trace / --pmc passes (profiles/r02_*).
    python tools/roofline_launch.py [repeats=12] [frames=1] [match] [plain]      `match`: also the NN match's launch shape
    (caelo_match_profile: `frames` pairs behind one k_match_prep + one k_match_screen launch; bench.py's second roofline object);
    `plain`: the encoder launches through caelo_encode, as the product issues them"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
import torch
from caelo import synth
import caelo; caelo.configure_runtime()
from caelo.engine import Engine
eng = Engine()
if os.environ.get("CAELO_ENC_S1") == "f32":   # (read HERE, by the tool: the library has no environment switch for arithmetic)
    eng.set_encoder_reference(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # frames per launch: 8 = the pipeline's launch shape (bench.py's headline roofline)
parts = []
pcs = []
for i in range(min(frames, 6)):
    pc = torch.from_numpy(synth.make_scan(i, quantum=1e-3)).to(eng.device)
    pcs.append(pc)
    ring, counter, _ = eng.project(pc)          # staged calls, no encoder launch: the trace then holds the profiled launches only
    kpts = eng.keypoints(ring, counter, eng.respond(ring))[0]
    parts.append(eng.patches(eng.voxelize(pc)[0], kpts.contiguous())[0].reshape(-1, 64))
bits = torch.cat([parts[i % len(parts)] for i in range(frames)], dim=0).contiguous()
plain = "plain" in sys.argv[3:]     # the PRODUCTION launch set (caelo_encode: k_enc_stage1x<false>, no events, no MFMA count) instead of
for _ in range(n):                  # the profiled one (k_enc_stage1x<true>): the two must take the same time in a kernel trace
    eng.encode(bits, group=3) if plain else eng.encode_profile(bits, group=3)
torch.cuda.synchronize()
if "match" in sys.argv[3:]:
    ff = [eng.extract(pc) for pc in pcs]
    chain = [ff[i % len(ff)] for i in range(min(frames, 8) + 1)]     # (neighbouring scans; a pair of a scan with itself at the wrap)
    ms = eng.match_profile(chain, repeats=n)
    print("match: %.1f us per %d pairs (prep %.1f us)" % (ms[0] * 1e3, len(chain) - 1, ms[1] * 1e3))
print("done", n)
