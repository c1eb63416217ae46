"""Host half of the exact RANSAC (csrc/certify.hip, caelo/hostexact.py) against the oracle -- CPU only, no GPU.

The reference's SolveRT / RANSAC4RT (Match.py:138-218) are NumPy statements whose bits depend on NumPy's BLAS; the host half
calls the same cblas_sgemm / cblas_sgemv / dgesdd entry points (caelo/hostblas.py).  Everything below is BIT-EXACT:
the oracle's SolveRT / RANSAC4RT / SolveRelativePose are the same NumPy statements as the reference's.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _pairs(a, b, p):
    f0 = np.load(os.path.join(GOLDEN, "frame_%s.npz" % a))
    f1 = np.load(os.path.join(GOLDEN, "frame_%s.npz" % b))
    pr = np.load(os.path.join(GOLDEN, "pair_%s.npz" % p))
    idx = pr["pair_idx"].astype(np.int64)
    return np.ascontiguousarray(f0["keypts_demo"][idx], np.float32), np.ascontiguousarray(f1["keypts_demo"], np.float32), pr


PAIRS = [("0", "1", "0_1"), ("q0", "q1", "q0_q1"), ("c0", "c1", "c0_c1")]


def test_host_blas_is_numpys_and_solve_rt_is_bit_exact(orc):
    from caelo import hostblas, hostexact, _ffi
    info = hostblas.bind(_ffi.load())
    assert os.path.exists(info["library"])
    rng = np.random.RandomState(3)
    n_checked = n_flip = 0
    for a, b, p in PAIRS:
        P0, P1, _ = _pairs(a, b, p)
        for t in range(3000):
            n = 4 if t % 10 else int(rng.randint(5, 300))
            idx = (rng.random_sample(n) * len(P0)).astype(np.int32)      # with replacement: repeated points are common
            R, T, cred = hostexact.solve_rt(P0[idx], P1[idx])
            Ro, To, co = orc.SolveRT(P0[idx], P1[idx])
            assert Ro.dtype == np.float32 and To.dtype == np.float32
            assert np.array_equal(R, Ro) and np.array_equal(T, To) and cred == co, (p, t, idx)
            n_checked += 1
            n_flip += cred < 0
    assert n_checked == 9000 and n_flip > 10      # the reflection branch (Match.py:151-155) is exercised


@pytest.mark.parametrize("a,b,p", PAIRS)
def test_host_ransac_equals_the_oracle_with_and_without_bounds(orc, a, b, p):
    """caelo_host_ransac: (1) without bounds = the reference's loop, every hypothesis evaluated; (2) with ANY valid upper
    bounds on the first level's counts the same result from a handful of evaluations -- the exact counts themselves, the
    counts plus slack, and 'no information' (N everywhere)."""
    from caelo import hostexact, _ffi
    P0, P1, _ = _pairs(a, b, p)
    N = len(P0)
    for seed in (11, 12, 13):
        draws = np.random.RandomState(seed).random_sample(6000)
        trace = []
        R, T, ok, m, thr = orc.RANSAC4RT(P0, P1, rng=np.random.RandomState(seed), trace=trace)
        want_iters = len([t for t in trace if t[2] == thr])
        r, mask, ev = hostexact.ransac(P0, P1, draws)
        assert np.array_equal(mask, m) and bool(r["success"]) == ok and abs(float(r["threshold"]) - thr) < 1e-6
        assert np.array_equal(r["R_ransac"].reshape(3, 3), R) and np.array_equal(r["T_ransac"].reshape(3, 1), T)
        assert int(r["iterations"]) == want_iters and ev == len(trace)
        Rf, Tf, _ = orc.SolveRT(P0[m], P1[m])       # SolveRelativePose's refit (Match.py:273-282)
        assert np.array_equal(r["R"].reshape(3, 3), Rf) and np.array_equal(r["T"].reshape(3, 1), Tf)
        # exact counts of ALL 500 first-level hypotheses (the reference's loop may stop earlier)
        cnt = np.zeros(500, np.int32)
        for t in range(500):
            idx = (draws[4 * t:4 * t + 4] * N).astype(np.int32)
            Rh, Th, _ = orc.SolveRT(P0[idx], P1[idx])
            cnt[t] = int((np.linalg.norm(P0 - (np.dot(Rh, P1.T) + Th).T, axis=1) < 0.4).sum())
        rs = np.random.RandomState(seed)
        for hi in (cnt, cnt + rs.randint(0, 4, 500).astype(np.int32), np.minimum(cnt + 40, N).astype(np.int32), np.full(500, N, np.int32)):
            r2, mask2, ev2 = hostexact.ransac(P0, P1, draws, hi=hi)
            assert np.array_equal(mask2, m) and r2.tobytes() == r.tobytes(), (seed, ev2)
            assert ev2 <= 501      # (+1: the winner once more for its mask when another candidate was evaluated after it)
        r3, _, ev3 = hostexact.ransac(P0, P1, draws, hi=cnt)
        assert ev3 <= 3                                 # tight bounds: the winner (and a tie) is all that is evaluated
        # an INVALID bound (below the true count of the hypothesis the replay lands on): the host half notices -- every count it
        # evaluates is compared with its bound -- decides the pair by the reference's loop without bounds, and counts the event
        lib = _ffi.load()
        v0 = int(lib.caelo_host_bound_violations())
        bad = cnt.copy()
        bad[int(r["best_trial"])] -= 1
        r4, mask4, ev4 = hostexact.ransac(P0, P1, draws, hi=bad)
        assert np.array_equal(mask4, m) and r4.tobytes() == r.tobytes() and ev4 > len(trace)
        assert int(lib.caelo_host_bound_violations()) == v0 + 1


def test_host_ransac_escalation_failure_and_tiny_inputs(orc):
    from caelo import hostexact, _ffi
    pr = np.load(os.path.join(GOLDEN, "pair_0_1.npz"))
    for k in ("esc", "fail"):
        P0, P1 = np.ascontiguousarray(pr[k + "_P0"]), np.ascontiguousarray(pr[k + "_P1"])
        for seed in (5, 6):
            draws = np.random.RandomState(seed).random_sample(6000)
            R, T, ok, m, thr = orc.RANSAC4RT(P0, P1, rng=np.random.RandomState(seed))
            r, mask, ev = hostexact.ransac(P0, P1, draws)
            assert bool(r["success"]) == ok and abs(float(r["threshold"]) - thr) < 1e-6 and np.array_equal(mask, m)
            if ok:
                assert np.array_equal(r["R_ransac"].reshape(3, 3), R.astype(np.float32)) and np.array_equal(r["T_ransac"].reshape(3, 1), T.astype(np.float32))
            else:
                assert int(r["best_trial"]) == -1 and np.array_equal(r["R"].reshape(3, 3), np.eye(3, dtype=np.float32))
            # a certificate whose bounds say "nobody reaches leastInliers" at 0.4 m: the higher levels run on the host
            r2, mask2, _ = hostexact.ransac(P0, P1, draws, hi=np.full(500, len(P0), np.int32))
            assert r2.tobytes() == r.tobytes() and np.array_equal(mask2, mask)
    # N < 5: leastInliers = 0, every hypothesis admissible (Match.py:166,:195-205)
    rs = np.random.RandomState(2)
    for n in (1, 2, 4):
        P1 = (rs.standard_normal((n, 3)) * 10).astype(np.float32)
        P0 = P1 + np.float32(0.01)
        draws = np.random.RandomState(9).random_sample(6000)
        R, T, ok, m, thr = orc.RANSAC4RT(P0, P1, rng=np.random.RandomState(9))
        r, mask, _ = hostexact.ransac(P0, P1, draws)
        assert bool(r["success"]) == ok and np.array_equal(mask, m) and abs(float(r["threshold"]) - thr) < 1e-6


def test_certificates_of_the_higher_levels(orc):
    """Round 6: a record that carries bounds for the 0.8 m / 1.6 m levels (caelo_ransac_cert::hi_up, written by k_ransac_hyp_up for a pair
    whose first level fails).  The host half then decides an escalating pair from a handful of evaluations WITHOUT the draws, with the
    reference's bits: exact counts as bounds, counts plus slack, and -- the failing pair -- bounds that say nobody reaches leastInliers
    at any level.  Without hi_up the same records need the draws (status 1) and ~1000 evaluations."""
    from caelo import hostexact, _ffi
    pr = np.load(os.path.join(GOLDEN, "pair_0_1.npz"))
    for k in ("esc", "fail"):
        P0, P1 = np.ascontiguousarray(pr[k + "_P0"]), np.ascontiguousarray(pr[k + "_P1"])
        N = len(P0)
        for seed in (5, 6):
            draws = np.random.RandomState(seed).random_sample(6000)
            want, wmask, wev = hostexact.ransac(P0, P1, draws)            # (== the oracle: test_host_ransac_escalation_failure_and_tiny_inputs)
            idx = (draws.reshape(3, 500, 4) * N).astype(np.int32)
            cnt = np.zeros((3, 500), np.int32)
            for l in range(3):
                for t in range(500):
                    Rh, Th, _ = orc.SolveRT(P0[idx[l, t]], P1[idx[l, t]])
                    cnt[l, t] = int((np.linalg.norm(P0 - (np.dot(Rh, P1.T) + Th).T, axis=1) < np.float32(0.4 * 2 ** l)).sum())
            assert cnt[0].max() < min(100, int(0.2 * N))                   # the first level fails: the pair escalates
            for slack in (0, 3):
                hi = np.minimum(cnt + slack, N).astype(np.int32)
                rec = hostexact.make_record(P0, P1, hi[0], idx[0], hi[1:], idx[1:])
                res, masks, evals, status = hostexact.certify_records(rec, None, 1)     # no draws given: none needed
                assert status[0] == 0 and res[0].tobytes() == want.tobytes() and np.array_equal(masks[0, :N].astype(bool), wmask)
                assert evals[0] <= (12 if slack == 0 else 80) and evals[0] < wev // 10, (k, seed, slack, evals[0], wev)
            # the same record without the higher levels' bounds: the draws are needed, every hypothesis of the higher levels is evaluated
            rec0 = hostexact.make_record(P0, P1, cnt[0], idx[0])
            with pytest.raises(_ffi.CaeloError):
                hostexact.certify_records(rec0, None, 1)
            res0, masks0, evals0, status0 = hostexact.certify_records(rec0, [draws], 1)
            assert status0[0] == 0 and res0[0].tobytes() == want.tobytes() and evals0[0] >= 100
            # an INVALID higher-level bound is noticed like a first-level one (the level is redone without bounds: needs the draws)
            if want["success"]:
                lvl = int(want["best_trial"]) // 500
                bad = cnt.copy(); bad[lvl, int(want["best_trial"]) % 500] -= 1
                recb = hostexact.make_record(P0, P1, bad[0], idx[0], bad[1:], idx[1:])
                v0 = int(_ffi.load().caelo_host_bound_violations())
                resb, masksb, _, stb = hostexact.certify_records(recb, [draws], 1)
                assert stb[0] == 0 and resb[0].tobytes() == want.tobytes() and int(_ffi.load().caelo_host_bound_violations()) == v0 + 1


def test_certify_records_threads_and_statuses(orc):
    """caelo_host_certify over hand-made records: many pairs on several threads, a record without bounds, an empty slot."""
    from caelo import hostexact, _ffi
    recs, want = [], []
    for (a, b, p), seed in zip(PAIRS * 4, range(12)):
        P0, P1, _ = _pairs(a, b, p)
        N = len(P0)
        draws = np.random.RandomState(40 + seed).random_sample(6000)
        idx = (draws[:2000].reshape(500, 4) * N).astype(np.int32)
        recs.append(hostexact.make_record(P0, P1, np.full(500, N, np.int32), idx))
        want.append(hostexact.ransac(P0, P1, draws))
    empty = np.zeros((1, _ffi.CERT_DTYPE.itemsize), np.uint8)
    nob = hostexact.make_record(P0, P1, np.zeros(500, np.int32), idx).copy()
    nob.view(_ffi.CERT_DTYPE)["flags"] = _ffi.CERT_NO_BOUNDS
    allrecs = np.concatenate(recs + [empty, nob])
    for threads in (1, 4):
        res, masks, evals, status = hostexact.certify_records(allrecs, None, threads)
        assert list(status) == [0] * 12 + [3, 2]
        for i, (r, m, ev) in enumerate(want):
            assert res[i].tobytes() == r.tobytes() and np.array_equal(masks[i, :len(m)].astype(bool), m) and not masks[i, len(m):].any()
