"""Host half of the exact RANSAC (csrc/certify.hip): thin NumPy-facing wrappers.  Needs no GPU.

The device scores the 500 hypotheses of a pair and leaves, per hypothesis, an upper bound on the inlier count the
reference's arithmetic can reach (the certificate, include/caelo.h).  The functions here replay Match.py:181-214 over those
bounds and re-evaluate the deciding hypotheses through the BLAS / LAPACK entry points of this process's NumPy
(caelo/hostblas.py), so that inlier sets, R_star / T_star and the refit are the reference's bits on this host.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi, hostblas

MAX_K = 1024


def _lib():
    lib = _ffi.load()
    hostblas.bind(lib)
    return lib


def solve_rt(p0, p1):
    """SolveRT (Match.py:138-158) by NumPy's own call sequence -> R [3,3] f32, T [3,1] f32, isCredible."""
    lib = _lib()
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    assert p0.shape == p1.shape and p0.ndim == 2 and p0.shape[1] == 3 and p0.shape[0] > 0
    R = np.empty((3, 3), np.float32)
    T = np.empty((3, 1), np.float32)
    cred = C.c_int32(0)
    _ffi.check(lib.caelo_host_solve_rt(p0.ctypes.data, p1.ctypes.data, p0.shape[0], R.ctypes.data, T.ctypes.data, C.byref(cred)))
    return R, T, int(cred.value)


def ransac(pairs0, pairs1, draws, hi=None):
    """RANSAC4RT + the refit of SolveRelativePose (Match.py:162-218,:269-283) on host arrays [n,3].  Without ``hi`` every
    hypothesis is evaluated like the reference's loop; with ``hi`` (upper bounds of the first level's 500 counts) only the
    deciding ones.  -> (result record (_ffi.POSE_DTYPE), mask [n] bool, hypotheses evaluated)."""
    lib = _lib()
    pairs0 = np.ascontiguousarray(pairs0, np.float32)
    pairs1 = np.ascontiguousarray(pairs1, np.float32)
    draws = np.ascontiguousarray(draws, np.float64).reshape(-1)
    assert draws.size >= 6000 and pairs0.shape == pairs1.shape
    n = pairs0.shape[0]
    res = np.zeros(1, dtype=_ffi.POSE_DTYPE)
    mask = np.zeros(max(1, n), dtype=np.uint8)
    ev = C.c_int32(0)
    hip = None
    if hi is not None:
        hi = np.ascontiguousarray(hi, np.int32)
        assert hi.size >= 500
        hip = hi.ctypes.data
    _ffi.check(lib.caelo_host_ransac(pairs0.ctypes.data, pairs1.ctypes.data, n, draws.ctypes.data, hip, res.ctypes.data, mask.ctypes.data,
                                     C.byref(ev)))
    return res[0], mask[:n].astype(bool), int(ev.value)


def certify_records(recs, rands=None, threads=None):
    """recs: [k, sizeof(caelo_ransac_cert)] u8 HOST array (copied from the device).  ``rands``: the pairs' draws (sequence of
    host arrays / tensors, or None): read only for a pair that escalates beyond the 0.4 m level.  -> (results record array
    [k], masks [k,1024] u8, evals [k] i32, status [k] i32: 0 exact, 2 no bounds in the record (> 1024 pairs), 3 no record)."""
    lib = _lib()
    recs = np.ascontiguousarray(recs)
    k = int(recs.shape[0])
    assert recs.dtype == np.uint8 and recs.ndim == 2 and recs.shape[1] == _ffi.CERT_DTYPE.itemsize
    results = np.zeros(k, dtype=_ffi.POSE_DTYPE)
    masks = np.zeros((k, MAX_K), dtype=np.uint8)
    evals = np.zeros(k, dtype=np.int32)
    status = np.zeros(k, dtype=np.int32)
    nt = int(threads) if threads else max(1, min(16, (os.cpu_count() or 2) // 2, (k + 7) // 8))
    _ffi.check(lib.caelo_host_certify(recs.ctypes.data, k, None, results.ctypes.data, masks.ctypes.data, MAX_K, evals.ctypes.data,
                                      status.ctypes.data, nt))
    for i in np.flatnonzero(status == 1):   # rare: the pair escalates beyond 0.4 m and the host half needs its draws
        if rands is None:
            raise _ffi.CaeloError("pair %d escalates beyond the first RANSAC level: its draws are needed (rands)" % i)
        r = rands[int(i)]
        if hasattr(r, "detach"):
            r = r.detach().cpu().numpy()
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
        assert r.size >= 6000
        ptrs = (C.c_void_p * 1)(r.ctypes.data)
        st1 = np.zeros(1, np.int32)
        _ffi.check(lib.caelo_host_certify(recs[i:i + 1].ctypes.data, 1, ptrs, results[i:i + 1].ctypes.data, masks[i:i + 1].ctypes.data, MAX_K,
                                          evals[i:i + 1].ctypes.data, st1.ctypes.data, 1))
        status[i] = st1[0]
    return results, masks, evals, status


def make_record(pairs0, pairs1, hi, idx, hi_up=None, idx_up=None):
    """A caelo_ransac_cert record from host arrays (tests, and callers that hold pairs on the host).  ``hi_up`` / ``idx_up`` [2,500] /
    [2,500,4]: the bounds and sample indices of the 0.8 m and 1.6 m levels (what k_ransac_hyp_up leaves for a pair that escalates)."""
    n = len(pairs0)
    assert n <= _ffi.CERT_MAX_PAIRS
    rec = np.zeros(1, dtype=_ffi.CERT_DTYPE)
    rec["magic"], rec["n_pairs"] = _ffi.CERT_MAGIC, n
    rec["hi"][0, :500] = hi
    rec["idx"][0, :500] = idx
    if hi_up is not None:
        rec["levels_up"] = 1
        rec["hi_up"][0, :, :500] = hi_up
        rec["idx_up"][0, :, :500] = idx_up
    rec["p0"][0, :n] = pairs0
    rec["p1"][0, :n] = pairs1
    return rec.view(np.uint8).reshape(1, -1)
