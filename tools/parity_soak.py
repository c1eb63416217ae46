#!/usr/bin/env python
"""parity_soak.py -- HIP path vs the CPU oracle over LONG runs of consecutive frames.

    python tools/parity_soak.py [--frames 200] [--scenes boxes,clutter,boxes_mm,shuffled] [--out profiles/r04_parity_soak.txt]

Per scene: F consecutive synthetic scans (caelo.synth; `boxes` = continuous coordinates, `clutter` and `boxes_mm` = coordinates in
whole millimetres, `shuffled` = boxes_mm with 1 % repeated points in a randomly permuted file order) go through

  * the batched pipeline (caelo_pipeline: what bench.py times) -- key pixels, descriptors, pair_idx, inlier masks, poses;
  * staged calls (caelo_voxelize_fast + caelo_voxmap_dump, caelo_patches) -- voxel sets of the three scales, patch bits;
  * the oracle (oracle/: the reference restated, pinned to the reference by tools/make_goldens.py), frame by frame and pair
    by pair with the same per-pair RANSAC draws (RandomState(seed_base + i), Match.py:182).

What must hold (the bars of BASELINE.json's north_star):

  key pixels, key points, voxel sets, patch bits        bit-exact, every frame
  descriptors                                           |got - want| <= 1e-4 max(|want|, 0.1), every element
  NN match on the ORACLE's descriptors (caelo_match)    pair_idx bit-exact, every column    (the match kernel on its own)
  RANSAC on the ORACLE's pairs (caelo_ransac +     inlier set, threshold, success, R_star / T_star and the refit BIT-EXACT, every pair, no
  the host half, Engine.certify)                        exception: the kernels score the 500 hypotheses and bound what the reference's float32 /
                                                        BLAS arithmetic can give each, the host half re-evaluates the deciding ones through
                                                        NumPy's own BLAS / LAPACK calls (csrc/certify.hip)
  the pipeline end to end                               pair_idx equal to the oracle's EXCEPT where the float64 margin between
                                                        the two candidates (oracle descriptors) is below what the descriptor
                                                        error of that pair can move a distance by (triangle inequality:
                                                        |d'(a,b) - d(a,b)| <= |da| + |db|); every exception is listed with both
                                                        numbers.  Pairs without exception: inlier sets AND poses bit-exact (same pairs in, same bits out);
                                                        pairs with one: inlier-set difference and pose difference listed.

The descriptor -> argmin -> inlier-set chain is float -> integer: a descriptor that differs in the 7th digit may flip an argmin
whose two candidates are closer than that.  Such a flip is not an error of either side (the f32 oracle is itself 1.3e-6 away
from a float64 evaluation of the network); what the soak shows is how often it happens and that NOTHING else differs.

Importable: tests/test_gpu_parity.py::test_parity_soak_short runs `soak()` on a few dozen frames per scene."""
import argparse
import concurrent.futures as cf
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cae-lo_amd"), os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

# every family on the closed "circuit" trajectory (caelo.synth.sensor_pose): structure at EVERY frame index -- on the "line" law of
# rounds 1-5 the sensor left the scene by frame ~150 and every later scan was the same bare ground plane in every family
TRAJECTORY = os.environ.get("CAELO_SOAK_TRAJECTORY", "circuit")
SCENES = {"boxes": dict(scene_kind="boxes", quantum=None), "clutter": dict(scene_kind="clutter", quantum=1e-3),
          "boxes_mm": dict(scene_kind="boxes", quantum=1e-3), "shuffled": dict(scene_kind="boxes", quantum=1e-3, shuffle=True)}
REL_TOL, FLOOR = 1e-4, 0.1


def _make(args):
    from caelo import synth
    frame, kw = args
    kw = dict(kw)
    shuffle = kw.pop("shuffle", False)
    pc = synth.make_scan(frame, trajectory=TRAJECTORY, **kw)
    return synth.shuffle_scan(pc, 1000 + frame) if shuffle else pc


def make_scans(scene, n, workers=None, first=0):
    """n consecutive scans of a scene; ray casting is ~0.5 s of one core per scan, so in worker processes (spawn: the parent may
    hold a HIP context)."""
    jobs = [(f, SCENES[scene]) for f in range(first, first + n)]   # (first: another stretch of the circuit -- other scans, not other seeds)
    workers = workers or min(n, max(1, (os.cpu_count() or 2) // 2), 48)
    if workers <= 1 or n <= 2:
        return [_make(j) for j in jobs]
    with cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_make, jobs, chunksize=max(1, n // (4 * workers))))


def oracle_frame(orc, models, pc, dist_channels=5):
    ring, cnt = orc.ProjectPC2SphericalRing(pc)
    resp = models[0].predict(ring[None, 0:64, 0:1792, 0:3])[0]
    if dist_channels == 3:   # batch mode (BatchPreprocess.py:97-98,131-136): the cropped three-channel ring, the full counter
        kp, kpix, _ = orc.GetKeyPtsByAE(np.ascontiguousarray(ring[0:64, 0:1792, 0:3]), cnt, resp)
    else:
        kp, kpix, _ = orc.GetKeyPtsByAE(ring, cnt, resp)
    v = orc.Voxelization(pc[:, 0:3])
    bits, flags = zip(*[orc.patches_bits(kp, v[6 + s], s) for s in range(3)])
    feats = np.concatenate([models[1].predict_bits(b) for b in bits], axis=1)
    return dict(kp=kp, kpix=kpix, vox=[v[6], v[7], v[8]], bits=np.stack(bits, 1), flags=np.stack(flags, 1), feats=feats)


def _sorted_rows(a):
    a = np.asarray(a, np.int32)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def soak(engine, orc, models, scene, n_frames, seed_base=5000, batch=8, log=None, scans=None, workers=None, dist_channels=5):
    """-> report dict (counts; `exceptions` = list of per-column records) for `n_frames` consecutive frames of `scene`."""
    import torch
    from caelo import _ffi
    from caelo.engine import ransac_draws
    say = log or (lambda *_: None)
    t0 = time.time()
    scans = scans if scans is not None else make_scans(scene, n_frames, workers)
    say("%s: %d scans synthesised in %.1f s" % (scene, n_frames, time.time() - t0))
    dev = engine.device
    dpcs = [torch.from_numpy(pc).to(dev) for pc in scans]
    draws = [ransac_draws(seed_base + i) for i in range(n_frames)]
    rnd = [torch.from_numpy(d).to(dev) for d in draws]
    pipe = engine.pipeline(batch)
    out = pipe.run(dpcs, rnd, dist_channels=dist_channels, certify=True, rands_host=draws)   # exact RANSAC: certifier thread of the pipeline
    torch.cuda.synchronize()
    cstat = pipe.cert_stats()
    # frames whose 496-nearest cut splits a tie class: the fused path used its canonical rule there (flag bit 2); redone the
    # reference's way (Engine.resolve_ties: ordered voxel lists, scikit-learn's kd-tree order), then the two pairs they are part of
    fl = out.flags.cpu().numpy()
    tied = [i for i in range(n_frames) if (fl[i] & 2).any()]
    tied2, n_tied_patches = engine.resolve_ties_many([(out.frame(i), dpcs[i]) for i in range(n_frames)], batch=out)
    assert tied2 == tied
    unresolved = int(getattr(engine, "last_tie_unresolved", 0))   # tie-split patches the kd-tree redo LEFT on the canonical rule (a quickselect gave up)
    n_tied_patches = sum(n_tied_patches)
    redo = sorted({j for t in tied for j in (t, t + 1) if 1 <= j < n_frames})
    if redo:
        rs_, ms_, xs_ = engine.match_pose_exact_many([(out.frame(i - 1), out.frame(i)) for i in redo], [rnd[i] for i in redo], [draws[i] for i in redo])
        for k_, i in enumerate(redo):
            out.result[i].copy_(torch.from_numpy(np.frombuffer(rs_[k_].tobytes(), np.uint8).copy())); out.inlier_mask[i].copy_(torch.from_numpy(ms_[k_])); out.pair_idx[i].copy_(xs_[k_])
    torch.cuda.synchronize()
    rows = out.rows.cpu().numpy(); kpix = out.key_pixels.cpu().numpy(); nkey = out.n_key.cpu().numpy()
    pidx = out.pair_idx.cpu().numpy(); mask = out.inlier_mask.cpu().numpy().astype(bool); res = out.result.cpu().numpy()
    status = out.status.cpu().numpy()
    rep = dict(scene=scene, frames=n_frames, pairs=n_frames - 1, frames_with_tie_split=len(tied), tie_split_patches=n_tied_patches, tie_redo_unresolved=unresolved, keypixel_mismatch_frames=0, keypoint_mismatch_frames=0,
               voxel_set_mismatch=[0, 0, 0], patch_mismatch=0, patch_mismatch_canonical_rule=0, patches=0, patches_truncated=0, patches_tie_ambiguous=0,
               desc_max_abs=0.0, desc_max_rel=0.0, desc_over_tol=0, status_or=int(np.bitwise_or.reduce(status[:, 0])),
               match_kernel_mismatch_cols=0, ransac_kernel_mismatch_pairs=0, ransac_kernel_max_rt=0.0,
               columns=0, flips=0, flips_unexplained=0, pairs_with_flip=0, exact_pairs_inlier_mismatch=0, exact_pairs_max_rt=0.0,
               flip_pairs_inlier_diff=[], flip_pairs_max_rt=0.0, success_mismatch=0, threshold_mismatch=0, exceptions=[],
               lane_faults=0, host_hypotheses_per_pair=round(cstat["evals_per_pair"], 2), certifier_us_per_pair=round(cstat["host_us_per_pair"], 1),
               ransac_kernel_evals_max=0, ransac_kernel_bitexact_pairs=0, pairs_compared=0, success_mismatch_exact_pairs=0, exact_pairs_bitexact_pose=0, exact_pairs=0,
               bound_checks=0, bound_violations=0, bound_slack_sum=0, bound_slack_max=0)
    prev = None
    vm = engine.voxmap(max(engine.max_points, max(p.shape[0] for p in scans)), slot=6)
    for i in range(n_frames):
        o = oracle_frame(orc, models, scans[i], dist_channels)
        k = len(o["kp"])
        same_pix = int(nkey[i]) == k and np.array_equal(kpix[i, :k], o["kpix"].astype(np.int64))
        same_pts = same_pix and np.array_equal(rows[i, :k, 60:63], o["kp"])
        rep["keypixel_mismatch_frames"] += 0 if same_pix else 1
        rep["keypoint_mismatch_frames"] += 0 if same_pts else 1
        # voxel sets and patches through the staged entry points (same kernels as the fused path)
        engine.voxelize_fast(dpcs[i], vm)
        for s in range(3):
            if not np.array_equal(engine.voxmap_voxels(vm, s), _sorted_rows(o["vox"][s])):
                rep["voxel_set_mismatch"][s] += 1
        gbits, gflags = engine.patches(vm, torch.from_numpy(np.ascontiguousarray(o["kp"])).to(dev))
        gbits = gbits.cpu().numpy().view(np.uint64)
        bad = (gbits != o["bits"]).any(axis=2)
        # the set-based map knows no list order: its tie-split patches (device flag 2 = oracle flag 4) follow the canonical rule
        split = (gflags.cpu().numpy() & 2) != 0
        assert np.array_equal(split, (o["flags"] & 4) != 0), "tie-split patches: device and oracle disagree on which"
        rep["patch_mismatch_canonical_rule"] += int((bad & split).sum())
        bad &= ~split
        rep["patch_mismatch"] += int(bad.sum()); rep["patches"] += bad.size
        rep["patches_truncated"] += int(((o["flags"] & 1) != 0).sum()); rep["patches_tie_ambiguous"] += int(((o["flags"] & 6) != 0).sum())
        if same_pts:
            d = np.abs(rows[i, :k, 0:60].astype(np.float64) - o["feats"])
            rel = d / np.maximum(np.abs(o["feats"]), FLOOR)
            rep["desc_max_abs"] = max(rep["desc_max_abs"], float(d.max())); rep["desc_max_rel"] = max(rep["desc_max_rel"], float(rel.max()))
            rep["desc_over_tol"] += int((rel > REL_TOL).sum())
        if prev is not None and same_pts and prev["same_pts"]:
            k0 = len(prev["kp"])
            # --- the oracle's pair
            o_idx, _ = orc.match(prev["feats"], o["feats"])
            R, T, ok, i0, i1, thr = orc.SolveRelativePose(prev["kp"], prev["feats"], None, o["kp"], o["feats"], None,
                                                          rng=np.random.RandomState(seed_base + i))
            # --- the kernels on the ORACLE's inputs: these have no excuse
            f0 = torch.from_numpy(np.ascontiguousarray(prev["feats"], np.float32)).to(dev); f1 = torch.from_numpy(np.ascontiguousarray(o["feats"], np.float32)).to(dev)
            g_idx = engine.match(f0, f1)
            rep["match_kernel_mismatch_cols"] += int((g_idx.cpu().numpy() != o_idx).sum())
            p0 = torch.from_numpy(np.ascontiguousarray(prev["kp"])).to(dev); p1 = torch.from_numpy(np.ascontiguousarray(o["kp"])).to(dev)
            rep["pairs_compared"] += 1
            cert = engine.new_cert(1)
            engine.ransac(p0, p1, torch.from_numpy(o_idx).to(dev), rnd[i], cert=cert[0])
            # the certificate itself: every hypothesis the oracle's loop evaluated at the first level has a count <= the kernel's bound
            trace = []
            orc.RANSAC4RT(prev["kp"][o_idx], o["kp"], rng=np.random.RandomState(seed_base + i), trace=trace)
            lvl0 = np.array([t[1] for t in trace if t[2] == 0.4], np.int64)
            hi = cert.cpu().numpy().view(_ffi.CERT_DTYPE).reshape(-1)[0]["hi"][:len(lvl0)].astype(np.int64)
            rep["bound_checks"] += len(lvl0); rep["bound_violations"] += int((hi < lvl0).sum())
            for t_ in np.flatnonzero(hi < lvl0):   # (tools/bound_violation_probe.py isolates and saves such a hypothesis)
                rep.setdefault("bound_violation_list", []).append((i, int(t_), int(hi[t_]), int(lvl0[t_])))
            rep["bound_slack_sum"] += int((hi - lvl0).sum()); rep["bound_slack_max"] = max(rep["bound_slack_max"], int((hi - lvl0).max()) if len(lvl0) else 0)
            cres, cmask, cev, cst = engine.certify(cert, [draws[i]])
            pr = cres[0]; m_ = cmask[0, :k].astype(bool)
            assert cst[0] == 0
            rt = max(np.abs(pr["R"].reshape(3, 3) - R).max(), np.abs(pr["T"] - T.ravel()).max() / max(1.0, np.abs(T).max()))
            same = (np.array_equal(np.flatnonzero(m_), i1) and bool(pr["success"]) == bool(ok) and pr["threshold"] == np.float32(thr))
            rep["ransac_kernel_mismatch_pairs"] += 0 if (same and rt <= REL_TOL) else 1
            rep["ransac_kernel_bitexact_pairs"] += int(same and np.array_equal(pr["R"].reshape(3, 3), R) and np.array_equal(pr["T"].reshape(3, 1), T))
            rep["ransac_kernel_max_rt"] = max(rep["ransac_kernel_max_rt"], float(rt))
            rep["ransac_kernel_evals_max"] = max(rep["ransac_kernel_evals_max"], int(cev[0]))
            # --- the pipeline end to end (its own descriptors)
            rep["columns"] += k
            flips = np.flatnonzero(pidx[i, :k] != o_idx)
            g0 = rows[i - 1, :k0, 0:60].astype(np.float64); g1 = rows[i, :k, 0:60].astype(np.float64)
            e0 = np.linalg.norm(g0 - prev["feats"], axis=1); e1 = np.linalg.norm(g1 - o["feats"], axis=1)
            for j in flips:
                a, b = int(o_idx[j]), int(pidx[i, j])
                fo = o["feats"][j].astype(np.float64)
                d_o = np.sqrt(((prev["feats"][a].astype(np.float64) - fo) ** 2).sum()); d_g = np.sqrt(((prev["feats"][b].astype(np.float64) - fo) ** 2).sum())
                margin = d_g - d_o                       # >= 0: the oracle's row is the float64 argmin of the oracle's descriptors
                reach = e0[a] + e0[b] + 2.0 * e1[j]      # how far the descriptor differences can move d(b) - d(a)
                rep["exceptions"].append(dict(scene=scene, frame=i, col=int(j), oracle_row=a, hip_row=b, margin=float(margin),
                                              reach=float(reach), explained=bool(margin <= reach)))
                rep["flips_unexplained"] += 0 if margin <= reach else 1
            rep["flips"] += len(flips)
            pr = _ffi.PoseResult.from_buffer_copy(res[i].tobytes())
            rt = max(np.abs(np.array(pr.R).reshape(3, 3) - R).max(), np.abs(np.array(pr.T) - T.ravel()).max() / max(1.0, np.abs(T).max()))
            rep["success_mismatch"] += int(bool(pr.success) != bool(ok)); rep["threshold_mismatch"] += int(pr.threshold != np.float32(thr))
            got_in = np.flatnonzero(mask[i, :k])
            if len(flips) == 0:
                rep["exact_pairs"] += 1
                rep["exact_pairs_inlier_mismatch"] += 0 if np.array_equal(got_in, i1) else 1
                rep["exact_pairs_max_rt"] = max(rep["exact_pairs_max_rt"], float(rt))
                rep["success_mismatch_exact_pairs"] += int(bool(pr.success) != bool(ok) or pr.threshold != np.float32(thr))
                rep["exact_pairs_bitexact_pose"] += int(np.array_equal(np.array(pr.R, np.float32).reshape(3, 3), R) and np.array_equal(np.array(pr.T, np.float32).reshape(3, 1), T))
            else:
                rep["pairs_with_flip"] += 1
                rep["flip_pairs_inlier_diff"].append((i, len(flips), len(np.setxor1d(got_in, i1)), len(i1)))
                rep["flip_pairs_max_rt"] = max(rep["flip_pairs_max_rt"], float(rt))
        prev = dict(o, same_pts=same_pts)
        if (i + 1) % 25 == 0:
            say("  %s: %d / %d frames, %.1f s" % (scene, i + 1, n_frames, time.time() - t0))
    rep["lane_faults"] = engine.lane_faults()
    rep["seconds"] = round(time.time() - t0, 1)
    return rep


def clean(rep):
    """True iff everything that must be bit-exact / within tolerance is: integers and RANSAC results exactly, descriptors within 1e-4,
    every argmin flip within the reach of the descriptor error.  No RANSAC difference is ever 'explained'."""
    return (rep["keypixel_mismatch_frames"] == 0 and rep["keypoint_mismatch_frames"] == 0 and sum(rep["voxel_set_mismatch"]) == 0
            and rep["patch_mismatch"] == 0 and rep["desc_over_tol"] == 0 and rep["status_or"] == 0
            and rep["match_kernel_mismatch_cols"] == 0 and rep["ransac_kernel_mismatch_pairs"] == 0
            and rep["ransac_kernel_bitexact_pairs"] == rep["pairs_compared"] and rep["bound_violations"] == 0
            and rep.get("tie_redo_unresolved", 0) == 0
            and rep["flips_unexplained"] == 0 and rep["exact_pairs_inlier_mismatch"] == 0 and rep["exact_pairs_max_rt"] <= REL_TOL
            and rep["success_mismatch_exact_pairs"] == 0 and rep["lane_faults"] == 0)


def render(rep):
    L = ["scene %-9s %d frames, %d pairs, %.0f s  -> %s" % (rep["scene"], rep["frames"], rep["pairs"], rep["seconds"], "CLEAN" if clean(rep) else "NOT CLEAN"),
         "  key pixels: %d frames differ; key points: %d; status OR 0x%x; lane faults %d" % (
             rep["keypixel_mismatch_frames"], rep["keypoint_mismatch_frames"], rep["status_or"], rep["lane_faults"]),
         "  voxel sets differing (frames, scale 0/1/2): %s" % rep["voxel_set_mismatch"],
         "  patches: %d of %d differ (truncated at the 496-NN cut: %d; cut inside a tie class: %d in %d frames -- the set-based fused build "
         "uses a canonical rule there (%d of them differ from the reference's choice) and Engine.resolve_ties redid those frames from ordered lists in scikit-learn's kd-tree order; "
         "%d left on the canonical rule by the redo)" % (
             rep["patch_mismatch"], rep["patches"], rep["patches_truncated"], rep["tie_split_patches"], rep["frames_with_tie_split"],
             rep["patch_mismatch_canonical_rule"], rep.get("tie_redo_unresolved", 0)),
         "  descriptors: max |err| %.3g, max relative (0.1 floor) %.3g, elements over 1e-4: %d" % (rep["desc_max_abs"], rep["desc_max_rel"], rep["desc_over_tol"]),
         "  kernels on the oracle's inputs: caelo_match %d wrong columns; caelo_ransac + host half: %d of %d pairs differ in inlier set / success / threshold / pose beyond 1e-4; "
         "%d of %d bit-exact incl. R_star, T_star and the refit (max R/T err %.2g; at most %d hypotheses re-evaluated on the host for a pair)" % (
             rep["match_kernel_mismatch_cols"], rep["ransac_kernel_mismatch_pairs"], rep["pairs_compared"], rep["ransac_kernel_bitexact_pairs"], rep["pairs_compared"],
             rep["ransac_kernel_max_rt"], rep["ransac_kernel_evals_max"]),
         "    certificates: %d hypothesis counts of the oracle's first-level loops checked against the kernel's upper bounds: %d violations; mean slack %.2f, largest %d" % (
             rep["bound_checks"], rep["bound_violations"], rep["bound_slack_sum"] / max(1, rep["bound_checks"]), rep["bound_slack_max"])
         + ("".join("\n      bound violated: frame %d trial %d: hi %d < reference count %d" % v for v in rep.get("bound_violation_list", []))),
         "  pipeline end to end (exact RANSAC inside the pipeline: %.2f hypotheses per pair on the host, %.1f us of the certifier thread per pair): %d of %d argmin columns differ from the oracle's (%d pairs); unexplained by the descriptor error: %d" % (
             rep["host_hypotheses_per_pair"], rep["certifier_us_per_pair"], rep["flips"], rep["columns"], rep["pairs_with_flip"], rep["flips_unexplained"]),
         "    pairs without a flip (%d): inlier sets differing %d, success / threshold differing %d, poses bit-identical to the oracle's %d, max R/T err %.2g" % (
             rep["exact_pairs"], rep["exact_pairs_inlier_mismatch"], rep["success_mismatch_exact_pairs"], rep["exact_pairs_bitexact_pose"], rep["exact_pairs_max_rt"]),
         "    pairs with a flip: max R/T err %.2g; (frame, flips, inlier-set symmetric difference, oracle inliers): %s" % (
             rep["flip_pairs_max_rt"], rep["flip_pairs_inlier_diff"])]
    for e in rep["exceptions"]:
        L.append("    flip %s frame %4d col %4d: oracle row %4d, HIP row %4d, float64 margin %.3g, descriptor reach %.3g  %s" % (
            e["scene"], e["frame"], e["col"], e["oracle_row"], e["hip_row"], e["margin"], e["reach"], "explained" if e["explained"] else "UNEXPLAINED"))
    return "\n".join(L)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--scenes", default="boxes,clutter,boxes_mm,shuffled")
    ap.add_argument("--shuffled-frames", type=int, default=24, help="frames of the `shuffled` scene (an order check, not a soak)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed-base", type=int, default=5000)
    ap.add_argument("--first-frame", type=int, default=0, help="frame index of the trajectory the soaked stretch starts at")
    ap.add_argument("--dist-channels", type=int, default=5, choices=(3, 5), help="5 = demo mode, 3 = batch mode of the key point rule (SURVEY 8a-3')")
    args = ap.parse_args()
    names = args.scenes.split(",")
    plan = {s: (args.shuffled_frames if s == "shuffled" else args.frames) for s in names}
    all_scans = {s: make_scans(s, n, first=args.first_frame) for s, n in plan.items()}      # before the HIP context exists
    import caelo
    caelo.configure_runtime()
    import oracle as orc
    from caelo.engine import Engine
    orc.build()
    models = orc.load_models(os.path.join(REPO, "weights", "SphericalRingPCRespondLayer.h5"), os.path.join(REPO, "weights", "EncoderModel4VoxelPatch.h5"))
    eng = Engine()
    text = ["parity soak: HIP pipeline vs CPU oracle, %s, oracle on %d threads, key point rule in %s mode (dist_channels %d)" % (
        time.strftime("%Y-%m-%d"), orc.num_threads(), "demo" if args.dist_channels == 5 else "batch", args.dist_channels)
        + ("" if not args.first_frame else "  [frames %d .. of the trajectory, seed base %d]" % (args.first_frame, args.seed_base))]
    ok = True
    for s in names:
        rep = soak(eng, orc, models, s, plan[s], seed_base=args.seed_base, log=lambda m: print(m, file=sys.stderr, flush=True), scans=all_scans[s],
                   dist_channels=args.dist_channels)
        ok &= clean(rep)
        text.append(render(rep))
        print(text[-1], flush=True)
    text.append("ALL CLEAN" if ok else "NOT CLEAN")
    print(text[-1])
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(text) + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
