#!/usr/bin/env python
"""tie_redo_check.py [--scene clutter] [--frames 600] [--first 0] -- the tie redo (Engine.resolve_ties: ordered voxel lists, scikit-learn's
kd-tree order, csrc/kdorder.hip) frame by frame against the oracle, on the frames whose 496-nearest cut splits a tie class.

For every frame of the run: the fused front half (key points, flags).  For every frame with a tie-split patch: the redo's bits
(voxelize -> voxmap_order -> patches, what resolve_ties_many issues) against oracle.patches_bits on the oracle's ordered voxel lists
(Voxel.py:177-216 restated, pinned to scikit-learn by tools/make_goldens.py), with the HIP key points (bit-identical to the oracle's
in every soak); and the HIP encoder on those bits against the oracle's.  Prints every patch that differs with its flags, the
number of voxels that differ and whether the oracle calls the patch tie-split.  (Written for the one clutter patch of
profiles/r06_parity_soak_600.txt whose descriptor was 5.6e-3 off.)"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cae-lo_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="clutter")
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--encode", type=int, default=1)
    a = ap.parse_args()
    import torch
    import oracle as orc
    import parity_soak as ps
    from caelo.engine import Engine
    eng = Engine()
    wdir = os.path.join(REPO, "weights")
    models = orc.load_models(os.path.join(wdir, "SphericalRingPCRespondLayer.h5"), os.path.join(wdir, "EncoderModel4VoxelPatch.h5"))
    t0 = time.time()
    scans = ps.make_scans(a.scene, a.first + a.frames)[a.first:]
    print("%d scans in %.0f s" % (len(scans), time.time() - t0), flush=True)
    dev = eng.device
    n_tied = n_bad = n_patches = n_desc_bad = 0
    for i, pc in enumerate(scans):
        d = torch.from_numpy(pc).to(dev)
        ff = eng.extract(d)
        k = int(ff.n_key)
        fl = ff.flags[:k].cpu().numpy()
        if not (fl & 2).any():
            continue
        n_tied += 1
        mask = sum(1 << s for s in range(3) if (fl[:, s] & 2).any())
        vm, st = eng.voxelize(d, eng.voxmap(max(eng.max_points, pc.shape[0]), slot=2))
        eng.voxmap_order(vm, mask)
        kp = ff.key_pts[:k].contiguous()
        bits, flags = eng.patches(vm, kp)
        torch.cuda.synchronize()
        gb = bits.cpu().numpy().view(np.uint64)
        gf = flags.cpu().numpy()
        v = orc.Voxelization(pc[:, 0:3])
        kph = kp.cpu().numpy()
        for s in range(3):
            ob, of = orc.patches_bits(kph, v[6 + s], s)
            bad = np.flatnonzero((gb[:, s] != ob).any(axis=1))
            n_patches += int(((of & 4) != 0).sum())
            for j in bad:
                x = gb[j, s] ^ ob[j]
                nd = int(sum(bin(int(w)).count("1") for w in x))
                n_bad += 1
                print("frame %d key point %d scale %d: %d voxels differ; device flags 0x%x (before the redo 0x%x), oracle flags 0x%x; list length %d"
                      % (a.first + i, j, s, nd, int(gf[j, s]), int(fl[j, s]), int(of[j]), len(v[6 + s])), flush=True)
            if a.encode and not len(bad):
                want = models[1].predict_bits(ob)
                got = eng.encode(torch.from_numpy(ob.view(np.int64)).to(dev).reshape(-1, 64), group=1).cpu().numpy()
                rel = np.abs(got - want) / np.maximum(np.abs(want), 0.1)
                if (rel > 1e-4).any():
                    n_desc_bad += 1
                    j = int(np.argmax(rel.max(axis=1)))
                    print("frame %d scale %d: ENCODER differs on equal bits, patch %d rel %.3g" % (a.first + i, s, j, rel.max()), flush=True)
        if n_tied % 25 == 0:
            print("  %d frames, %d tied, %d tie-split patches, %d differ   %.0f s" % (i + 1, n_tied, n_patches, n_bad, time.time() - t0), flush=True)
    print("%s: %d frames, %d with a tie-split patch (%d such patches by the oracle's flags); redo bits differing from the oracle's: %d patches; encoder mismatches on equal bits: %d"
          % (a.scene, len(scans), n_tied, n_patches, n_bad, n_desc_bad))


if __name__ == "__main__":
    main()
