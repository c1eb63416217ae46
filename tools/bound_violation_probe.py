"""Where a RANSAC certificate's upper bound falls below the reference's count (tools/parity_soak.py counts such violations; this probe
finds and dumps them): the pipeline's own key points and matches of a scene family, the oracle's RANSAC4RT trace per pair, the kernels'
`hi` per hypothesis.  Every violating hypothesis is printed (pair, trial, hi, count, the four sample rows) and saved for CPU analysis.
    python tools/bound_violation_probe.py [scene=clutter] [frames=150] [seed_base=12000] [out=gpurun_out/bound_violations.npz]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cae-lo_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import _ffi
from caelo.engine import Engine, ransac_draws
import oracle as orc
import parity_soak as ps

def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "clutter"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    seed_base = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(REPO, "gpurun_out", "bound_violations.npz")
    eng = Engine()
    dev = eng.device
    scans = ps.make_scans(scene, n)
    dpcs = [torch.from_numpy(pc).to(dev) for pc in scans]
    draws = [ransac_draws(seed_base + i) for i in range(n)]
    rnd = [torch.from_numpy(d).to(dev) for d in draws]
    pipe = eng.pipeline(8)
    out = pipe.run(dpcs, rnd, certify=True, rands_host=draws)
    torch.cuda.synchronize()
    fl = out.flags.cpu().numpy()
    tied = [i for i in range(n) if (fl[i] & 2).any()]
    eng.resolve_ties_many([(out.frame(i), dpcs[i]) for i in range(n)], batch=out)
    redo = sorted({j for t in tied for j in (t, t + 1) if 1 <= j < n})
    if redo:
        rs_, ms_, xs_ = eng.match_pose_exact_many([(out.frame(i - 1), out.frame(i)) for i in redo], [rnd[i] for i in redo], [draws[i] for i in redo])
        for k_, i in enumerate(redo):
            out.pair_idx[i].copy_(xs_[k_])
    torch.cuda.synchronize()
    rows = out.rows.cpu().numpy(); nkey = out.n_key.cpu().numpy(); pidx = out.pair_idx.cpu().numpy()
    found = []
    checks = 0
    for i in range(1, n):
        k0, k = int(nkey[i - 1]), int(nkey[i])
        kp0, kp1 = np.ascontiguousarray(rows[i - 1, :k0, 60:63]), np.ascontiguousarray(rows[i, :k, 60:63])
        idx = pidx[i, :k].astype(np.int64)
        cert = eng.new_cert(1)
        eng.ransac(torch.from_numpy(kp0).to(dev), torch.from_numpy(kp1).to(dev), torch.from_numpy(idx).to(dev), rnd[i], cert=cert[0])
        trace = []
        orc.RANSAC4RT(kp0[idx], kp1, rng=np.random.RandomState(seed_base + i), trace=trace)
        lvl0 = np.array([t[1] for t in trace if t[2] == 0.4], np.int64)
        rec = cert.cpu().numpy().view(_ffi.CERT_DTYPE).reshape(-1)[0]
        hi = rec["hi"][:len(lvl0)].astype(np.int64)
        checks += len(lvl0)
        for t in np.flatnonzero(hi < lvl0):
            smp = rec["idx"][t]
            print("pair %d (frames %d,%d) trial %d: hi %d < reference count %d; sample rows %s; n_pairs %d" % (i, i - 1, i, t, hi[t], lvl0[t], smp.tolist(), k))
            found.append(dict(pair=i, trial=int(t), hi=int(hi[t]), count=int(lvl0[t]), sample=smp.copy(), P0=kp0[idx].copy(), P1=kp1.copy(), seed=seed_base + i))
    print("%s: %d frames, seed base %d: %d hypothesis counts checked, %d violations" % (scene, n, seed_base, checks, len(found)))
    if found:
        np.savez_compressed(out_path, **{"v%d_%s" % (j, k_): np.asarray(v_) for j, f in enumerate(found) for k_, v_ in f.items()})
        print("saved", out_path)


if __name__ == "__main__":
    main()
