#!/usr/bin/env python
"""demo_match.py -- the counterpart of the reference's `Match.py __main__` (Match.py:294-356, SURVEY.md 8c harness
row) with the same function names, argument order and return tuples, served by libcaelo through caelo.api.

    python demo_match.py [scan0.bin scan1.bin] [--seed 0]

Without file arguments two synthetic 64x2000 scans (caelo.synth frames 0 and 1) are used.  Prints what the
reference prints at the end of its demo: R, T, success, the inlier counts and the residual threshold.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from caelo import stageio, synth  # noqa: E402
from caelo.api import (GetFeaturesFromPatches, GetKeyPtsByAE, GetPatchesList, ProjectPC2SphericalRing,  # noqa: E402
                       SolveRelativePose, Voxelization, load_model)
from caelo.engine import ENCODER_H5, RESPOND_H5  # noqa: E402


def frame(PC, RespondLayer, PatchEncoder):
    SphericalRing, GridCounter = ProjectPC2SphericalRing(PC)                                    # SphericalRing.py:72
    x = np.ascontiguousarray(SphericalRing[0:64, 0:1792, 0:3])
    RespondImg = np.squeeze(RespondLayer.predict(x.reshape((1,) + x.shape)))                    # SphericalRing.py:405-408
    KeyPts, KeyPixels, PlanarPts = GetKeyPtsByAE(SphericalRing, GridCounter, RespondImg)        # :414 (demo mode, 5 channels)
    out = Voxelization(PC[:, 0:3])                                                              # Voxel.py:100
    KeyPts, PatchesList = GetPatchesList(KeyPts, out[6], out[7], out[8])                        # Match.py:330
    Features = GetFeaturesFromPatches(PatchEncoder, PatchesList)                                # Match.py:336
    return KeyPts, Features


def run(PC0, PC1, seed=0):
    """Match.py:312-353 on two scans -> dict of everything the reference's demo computes."""
    PatchEncoder = load_model(ENCODER_H5)                                                       # Match.py:313
    RespondLayer = load_model(RESPOND_H5)                                                       # Match.py:324
    KeyPts0, Features0 = frame(PC0, RespondLayer, PatchEncoder)
    KeyPts1, Features1 = frame(PC1, RespondLayer, PatchEncoder)
    Weights0 = np.ones((KeyPts0.shape[0], 1), np.float32); Weights1 = np.ones((KeyPts1.shape[0], 1), np.float32)
    R, T, isSuccess, inliersIdx0, inliersIdx1, residualThreshold = SolveRelativePose(
        KeyPts0, Features0, Weights0, KeyPts1, Features1, Weights1, rng=np.random.RandomState(seed))   # Match.py:349
    return dict(KeyPts0=KeyPts0, Features0=Features0, KeyPts1=KeyPts1, Features1=Features1, R=R, T=T, isSuccess=isSuccess,
                inliersIdx0=inliersIdx0, inliersIdx1=inliersIdx1, residualThreshold=residualThreshold)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    files = [a for a in argv if not a.startswith("--") and not a.lstrip("-").isdigit()]
    seed = int(argv[argv.index("--seed") + 1]) if "--seed" in argv else 0
    PC0, PC1 = (stageio.read_scan(files[0]), stageio.read_scan(files[1])) if len(files) >= 2 else (synth.make_scan(0), synth.make_scan(1))
    o = run(PC0, PC1, seed)
    R, T = o["R"], o["T"]
    print("nKeyPts =", o["KeyPts0"].shape[0], o["KeyPts1"].shape[0])
    print("R =\n", np.round(R, 5)); print("T =", np.round(np.asarray(T).ravel(), 4))
    print("isSuccess =", o["isSuccess"], " nInliers =", len(o["inliersIdx0"]), " residualThreshold =", o["residualThreshold"])
    if len(files) < 2:
        Rg, Tg = synth.relative_pose_gt(0, 1)
        print("synthetic ground truth T =", np.round(Tg.ravel(), 4), " |dT| = %.4f m" % float(np.linalg.norm(np.asarray(T).ravel() - Tg.ravel())))


if __name__ == "__main__":
    main()
