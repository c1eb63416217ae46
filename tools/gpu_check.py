#!/usr/bin/env python
"""Stage-by-stage parity report (GPU box): HIP engine vs the CPU oracle on one synthetic frame pair.
Prints a line per stage; never stops at the first mismatch (one gpurun call = maximum signal)."""
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))

import torch  # noqa: E402

import oracle as orc  # noqa: E402
from caelo import synth  # noqa: E402
import caelo; caelo.configure_runtime()
from caelo.engine import Engine, ransac_draws  # noqa: E402

results = []


def stage(name):
    def deco(fn):
        t0 = time.time()
        try:
            msg = fn()
            ok = True
        except Exception as e:  # noqa: BLE001
            msg = "EXC %s: %s\n%s" % (type(e).__name__, e, traceback.format_exc())
            ok = False
        print("[%s] %-22s %s  (%.2fs)" % ("OK " if ok and not str(msg).startswith("FAIL") else "BAD", name, msg, time.time() - t0), flush=True)
        results.append((name, ok and not str(msg).startswith("FAIL")))
        return fn
    return deco


eng = Engine()
dev = eng.device
resp_model, enc_model = orc.load_models(os.path.join(REPO, "weights", "SphericalRingPCRespondLayer.h5"),
                                         os.path.join(REPO, "weights", "EncoderModel4VoxelPatch.h5"))
S = {}
pc = synth.make_scan(0)
pcd = torch.from_numpy(pc).to(dev)
o_ring, o_cnt = orc.ProjectPC2SphericalRing(pc)
o_resp = resp_model.predict(o_ring[None, 0:64, 0:1792, 0:3])[0]
o_kp, o_kpix, _ = orc.GetKeyPtsByAE(o_ring, o_cnt, o_resp)
o_vox = orc.Voxelization(pc[:, 0:3])
oA = (o_vox[6], o_vox[7], o_vox[8])


@stage("project")
def _():
    ring, cnt, st = eng.project(pcd)
    S["ring"], S["cnt"] = ring, cnt
    r, c = ring.cpu().numpy(), cnt.cpu().numpy()
    ok = np.array_equal(r, o_ring) and np.array_equal(c, o_cnt)
    return "%s ring_diff=%d cnt_diff=%d status=%d" % ("exact" if ok else "FAIL", int((r != o_ring).sum()), int((c != o_cnt).sum()), int(st.item()))


@stage("respond")
def _():
    resp = eng.respond(torch.from_numpy(o_ring).to(dev))
    S["resp"] = resp
    r = resp.cpu().numpy()
    nd = int((r != o_resp).sum())
    return "%s diff=%d maxabs=%.3g" % ("exact" if nd == 0 else "FAIL", nd, float(np.abs(r - o_resp).max()))


@stage("keypoints")
def _():
    kpts, kpix, nkey, st = eng.keypoints(torch.from_numpy(o_ring).to(dev), torch.from_numpy(o_cnt).to(dev), torch.from_numpy(o_resp).to(dev))
    k = int(nkey.item())
    a, b = kpix[:k].cpu().numpy(), kpts[:k].cpu().numpy()
    ok = k == len(o_kpix) and np.array_equal(a, o_kpix) and np.array_equal(b, o_kp)
    extra = ""
    if not ok and k == len(o_kpix):
        extra = " first_diff=%s" % (np.nonzero((a != o_kpix).any(axis=1))[0][:5],)
    # batch mode
    r3 = np.ascontiguousarray(o_ring[0:64, 0:1792, 0:3]); c3 = np.ascontiguousarray(o_cnt[0:64, 0:1792])
    o2 = orc.GetKeyPtsByAE(r3, c3, o_resp)
    k2 = eng.keypoints(torch.from_numpy(r3).to(dev), torch.from_numpy(c3).to(dev), torch.from_numpy(o_resp).to(dev))
    ok2 = np.array_equal(k2[1][:int(k2[2].item())].cpu().numpy(), o2[1])
    return "%s K=%d/%d batch_mode=%s%s" % ("exact" if ok and ok2 else "FAIL", k, len(o_kpix), ok2, extra)


@stage("voxelize+export")
def _():
    vmap, st = eng.voxelize(pcd[:, 0:3].contiguous())
    S["vmap"] = vmap
    a = eng.voxmap_export(vmap, pcd.shape[0])
    oks = [np.array_equal(x.cpu().numpy(), y) for x, y in zip(a, oA)]
    return "%s counts=%s/%s status=%d order_exact=%s" % ("exact" if all(oks) else "FAIL", [len(x) for x in a], [len(y) for y in oA], int(st.item()), oks)


@stage("patches(map)")
def _():
    kp = torch.from_numpy(o_kp).to(dev)
    bits, flags = eng.patches(S["vmap"], kp)
    S["bits"] = bits
    b = bits.cpu().numpy().view(np.uint64)
    ob = np.stack([orc.patches_bits(o_kp, oA[s], s)[0] for s in range(3)], axis=1)
    S["obits"] = ob
    nd = int((b != ob).any(axis=2).sum())
    return "%s differing_patches=%d flags=%s setbits=%d/%d" % ("exact" if nd == 0 else "FAIL", nd, flags.cpu().numpy().sum(axis=0), int(np.unpackbits(b.view(np.uint8)).sum()), int(np.unpackbits(ob.view(np.uint8)).sum()))


@stage("patches(lists)+trunc")
def _():
    g = np.load(os.path.join(REPO, "tests", "golden", "patch_truncation.npz"))
    msgs = []
    allok = True
    for name in ("sparse", "mid", "dense"):
        vox, pts = g[name + "_vox"], g[name + "_pts"]
        v = torch.from_numpy(vox).to(dev)
        vmap, st = eng.voxmap_from_lists(v, v, v)
        bits, flags = eng.patches(vmap, torch.from_numpy(pts).to(dev))
        b = bits[:, 1, :].cpu().numpy().view(np.uint64)
        ob, of = orc.patches_bits(pts, vox, 1)
        ok = np.array_equal(b, ob) and np.array_equal(flags[:, 1].cpu().numpy(), of)
        ref_ok = np.array_equal(b[(of & 2) == 0], g[name + "_bits"][(of & 2) == 0])
        allok &= ok and ref_ok
        msgs.append("%s:%s/%s" % (name, ok, ref_ok))
    return ("exact " if allok else "FAIL ") + " ".join(msgs)


@stage("unpack/pack")
def _():
    bits = S["bits"][:64, 1, :].contiguous()
    dense = eng.unpack_patches(bits)
    od = orc.unpack_patches(bits.cpu().numpy().view(np.uint64))
    back = eng.pack_patches(dense)
    ok = np.array_equal(dense.cpu().numpy(), od) and torch.equal(back, bits)
    return "exact" if ok else "FAIL"


@stage("encoder")
def _():
    ob = S["obits"]
    bits = torch.from_numpy(ob.view(np.int64)).to(dev)
    feats = eng.encode(bits, group=3)
    S["feats"] = feats
    f = feats.cpu().numpy()
    of = np.concatenate([enc_model.predict_bits(np.ascontiguousarray(ob[:, s])) for s in range(3)], axis=1)
    S["ofeats"] = of
    err = np.abs(f - of)
    rel = err.max() / np.abs(of).max()
    bad = np.argwhere(err > 1e-4)
    return "%s max_abs=%.3g rel=%.3g nan=%d bad=%d first_bad=%s" % ("ok" if rel < 1e-4 else "FAIL", err.max(), rel, int(np.isnan(f).sum()), len(bad), bad[:4].tolist())


@stage("encoder group=1 odd n")
def _():
    ob = S["obits"][:37, 2]
    f = eng.encode(torch.from_numpy(np.ascontiguousarray(ob).view(np.int64)).to(dev), group=1).cpu().numpy()
    of = enc_model.predict_bits(np.ascontiguousarray(ob))
    return "%s max_abs=%.3g" % ("ok" if np.abs(f - of).max() < 1e-4 else "FAIL", np.abs(f - of).max())


pc1 = synth.make_scan(1)


@stage("extract(fused) x2")
def _():
    fa = eng.extract(pcd)
    fb = eng.extract(torch.from_numpy(pc1).to(dev))
    S["fa"], S["fb"] = fa, fb
    k = int(fa.n_key.item())
    okk = np.array_equal(fa.key_pixels[:k].cpu().numpy(), o_kpix)
    err = np.abs(fa.features[:k].cpu().numpy() - S["ofeats"]).max()
    return "%s K=%d kp_exact=%s feat_err=%.3g status=%d/%d" % ("ok" if okk and err < 1e-4 else "FAIL", k, okk, err, int(fa.status[0].item()), int(fb.status[0].item()))


@stage("match")
def _():
    fa, fb = S["fa"], S["fb"]
    idx = eng.match(fa.features, fb.features, fa.n_key, fb.n_key).cpu().numpy()
    oi, od = orc.match(fa.features.cpu().numpy(), fb.features.cpu().numpy())
    S["pair_idx"] = idx
    return "%s diff=%d" % ("exact" if np.array_equal(idx, oi) else "FAIL", int((idx != oi).sum()))


@stage("ransac+pose")
def _():
    fa, fb = S["fa"], S["fb"]
    out = []
    allok = True
    for seed in (0, 1, 2, 3):
        rand = torch.from_numpy(ransac_draws(seed)).to(dev)
        res, mask, idx = eng.match_pose(fa, fb, rand)
        r = eng.pose_result(res)
        trace = []
        oR, oT, ook, oi0, oi1, othr = orc.SolveRelativePose(fa.key_pts.cpu().numpy(), fa.features.cpu().numpy(), None, fb.key_pts.cpu().numpy(), fb.features.cpu().numpy(), None, rng=np.random.RandomState(seed), trace=trace)
        m = mask.cpu().numpy().astype(bool)
        same = np.array_equal(np.nonzero(m)[0], oi1)
        R = np.array(r.R).reshape(3, 3); T = np.array(r.T)
        dR = np.abs(R - oR).max(); dT = np.abs(T - oT.ravel()).max()
        ok = same and bool(r.success) == bool(ook) and abs(r.threshold - othr) < 1e-6 and dR < 1e-4 and dT < 1e-4 * max(1, np.abs(oT).max())
        allok &= ok
        out.append("s%d:%s in=%d/%d it=%d dR=%.1e dT=%.1e" % (seed, "ok" if ok else "BAD", int(m.sum()), len(oi1), r.iterations, dR, dT))
    return ("ok " if allok else "FAIL ") + " | ".join(out)


@stage("ransac esc/fail")
def _():
    g = np.load(os.path.join(REPO, "tests", "golden", "pair_0_1.npz"))
    out = []
    allok = True
    for name in ("esc", "fail"):
        P0, P1 = g[name + "_P0"], g[name + "_P1"]
        rand = torch.from_numpy(ransac_draws(7)).to(dev)
        idx = torch.arange(len(P1), device=dev)
        res, mask = eng.ransac(torch.from_numpy(P0).to(dev), torch.from_numpy(P1).to(dev), idx, rand)
        r = eng.pose_result(res)
        m = mask.cpu().numpy().astype(bool)
        ok = bool(r.success) == bool(g[name + "_ok"]) and abs(r.threshold - float(g[name + "_thr"])) < 1e-6 and np.array_equal(m, g[name + "_mask"])
        allok &= ok
        out.append("%s:%s succ=%d thr=%.1f in=%d/%d" % (name, "ok" if ok else "BAD", r.success, r.threshold, int(m.sum()), int(g[name + "_mask"].sum())))
    return ("ok " if allok else "FAIL ") + " | ".join(out)


@stage("solve_rt")
def _():
    rs = np.random.RandomState(3)
    P1 = rs.uniform(-20, 20, (200, 3)).astype(np.float32)
    Rg, Tg = synth.relative_pose_gt(0, 5)
    P0 = (P1 @ Rg.T + Tg.T).astype(np.float32)
    R, T, c = eng.solve_rt(torch.from_numpy(P0).to(dev), torch.from_numpy(P1).to(dev))
    oR, oT, oc = orc.SolveRT(P0, P1)
    return "%s dR=%.2e dT=%.2e cred=%d/%d" % ("ok" if np.abs(R.cpu().numpy() - oR).max() < 1e-5 else "FAIL", np.abs(R.cpu().numpy() - oR).max(), np.abs(T.cpu().numpy() - oT).max(), int(c.item()), oc)


@stage("timing (eager)")
def _():
    torch.cuda.synchronize()
    for _ in range(3):
        fa = eng.extract(pcd)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    for _ in range(n):
        fa = eng.extract(pcd)
    torch.cuda.synchronize()
    t1 = time.time()
    rand = torch.from_numpy(ransac_draws(0)).to(dev)
    for _ in range(n):
        eng.match_pose(S["fa"], S["fb"], rand)
    torch.cuda.synchronize()
    t2 = time.time()
    return "extract %.3f ms/frame, match+ransac %.3f ms/pair" % ((t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3)


bad = [n for n, ok in results if not ok]
print("SUMMARY: %d/%d stages ok; bad=%s" % (len(results) - len(bad), len(results), bad))
sys.exit(1 if bad else 0)
