"""Device engine: one HIP context + weights + reusable buffers per process (one process per GPU).

``Engine`` owns the libcaelo context, loads the two Keras ``.h5`` files through ``h5lite`` and
exposes (a) stage methods on device tensors, one per C-ABI entry point, and (b) the fused hot
path ``extract`` (scan -> keypoints + 60-d descriptors) and ``match_pose`` (two frames -> rigid
pose) that run without any host synchronisation until the caller reads a result.

PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os
import threading
import time

import numpy as np
import torch

from . import _ffi
from .h5lite import H5File

RING_H, RING_W, RING_C = 69, 1800, 5
NET_H, NET_W = 64, 1792
MAX_K = 1024

ST_COL_OOB, ST_VOXEL_OOB, ST_MAP_FULL, ST_FEW_VOXELS, ST_FEW_KEYPTS, ST_NONFINITE = 1, 2, 4, 8, 16, 32

_DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "weights")
RESPOND_H5 = os.path.join(_DEFAULT_WEIGHTS, "SphericalRingPCRespondLayer.h5")
ENCODER_H5 = os.path.join(_DEFAULT_WEIGHTS, "EncoderModel4VoxelPatch.h5")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _hptr(a):
    return a.ctypes.data_as(C.c_void_p)


def raise_status(st):
    """Map device status bits to the exception the reference would raise (SURVEY 8b 'Errors')."""
    if st & ST_COL_OOB:
        raise IndexError("index 1800 is out of bounds for axis 1 with size 1800")  # SphericalRing.py:91
    if st & ST_VOXEL_OOB:
        raise IndexError("voxel index out of bounds for axis with size 64")        # Voxel.py:139
    if st & ST_NONFINITE:
        raise ValueError("cannot convert float NaN to integer")                    # SphericalRing.py:86-88, Voxel.py:122-124
    if st & ST_FEW_VOXELS:
        raise ValueError("Expected n_neighbors <= n_samples (n_neighbors = 496)")  # Voxel.py:195-196
    if st & ST_FEW_KEYPTS:
        raise AssertionError("KeyPts.shape[0] > 50")                               # SphericalRing.py:286
    if st & ST_MAP_FULL:
        raise _ffi.CaeloError("voxel map overflow")


def read_keras_weights(path):
    """-> ("respond", [w1,b1,w2,b2]) or ("encoder", [10 arrays]) from a Keras 2.2 .h5 file.  The layer stack is read
    from the file's ``model_config`` and compared field by field with what the kernels implement (keras_config.check):
    any other activation / padding / stride / data format / shape is refused."""
    from . import keras_config
    h = H5File(path)
    kind = keras_config.check(keras_config.layers(h))
    names = [n.decode() for n in h.attrs("/model_weights")["layer_names"]]
    ws = []
    for ln in names:
        for wn in h.attrs("/model_weights/" + ln).get("weight_names", []):
            ws.append(np.ascontiguousarray(h.dataset("/model_weights/%s/%s" % (ln, wn.decode())), np.float32))
    want = [864, 32, 256, 8] if kind == "respond" else [216, 8, 3456, 16, 13824, 32, 409600, 200, 4000, 20]
    if [w.size for w in ws] != want:
        raise ValueError("%s: weight shapes %s do not match the %s architecture" % (path, [w.shape for w in ws], kind))
    return kind, ws


class VoxelMap:
    def __init__(self, eng, max_points):
        self.eng, self.max_points = eng, int(max_points)
        h = C.c_void_p()
        _ffi.check(eng.lib.caelo_voxmap_create(eng.ctx, self.max_points, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h:
                self.eng.lib.caelo_voxmap_destroy(self.h)
                self.h = None
        except Exception:
            pass


class FrameFeatures:
    """Device-resident result of Engine.extract for one scan.  ``rows`` [1024,64] f32 holds the
    60-d descriptor (cols 0:60, 16-byte aligned rows for the match kernel's vector loads) | xyz (60:63)
    | valid flag (63): the unit the multi-GPU all-gather moves; key_pts / features are strided views."""
    __slots__ = ("rows", "key_pts", "key_pixels", "features", "n_key", "status", "flags")

    def __init__(self, rows, key_pixels, n_key, status, flags):
        self.rows = rows
        self.key_pts, self.features = rows[:, 60:63], rows[:, 0:60]
        self.key_pixels, self.n_key, self.status, self.flags = key_pixels, n_key, status, flags

    @classmethod
    def from_rows(cls, rows):
        """Rebuild from gathered rows (another rank's frame): n_key = number of valid rows."""
        n_key = rows[:, 63].sum().round().to(torch.int32).reshape(1)
        return cls(rows, None, n_key, None, None)


class FrameBatch:
    """Device-resident outputs of Pipeline.run for K frames (one allocation per field)."""

    def __init__(self, eng, k):
        self.k = k
        self.rows = eng.empty((k, MAX_K, 64), torch.float32)       # the multi-GPU all-gather payload
        self.key_pixels = eng.empty((k, MAX_K, 2), torch.int64)
        self.n_key = eng.empty((k,), torch.int32)
        self.flags = eng.empty((k, MAX_K, 3), torch.uint8)
        self.status = eng.empty((k, 4), torch.int32)
        self.result = eng.zeros((k, C.sizeof(_ffi.PoseResult)), torch.uint8)   # pair (i-1, i) in slot i
        self.inlier_mask = eng.zeros((k, MAX_K), torch.uint8)
        self.pair_idx = eng.zeros((k, MAX_K), torch.int64)
        self.cert = None       # [k, sizeof(caelo_ransac_cert)] u8 once a run asked for certificates (Pipeline.run(certify=True))
        self.exact = None      # host copy of the certified results: (results [k], masks [k,1024], evals [k], status [k])
        self._rands_host = None
        self._exact_buf = None

    def ensure_cert(self, eng):
        if self.cert is None:
            self.cert = eng.zeros((self.k, _ffi.CERT_DTYPE.itemsize), torch.uint8)
        return self.cert

    def frame(self, i):
        return FrameFeatures(self.rows[i], self.key_pixels[i], self.n_key[i:i + 1], self.status[i], self.flags[i])

    def view(self, start, n):
        """Frames start .. start + n as a FrameBatch over the same storage."""
        v = object.__new__(FrameBatch)
        v.k = n
        for f in ("rows", "key_pixels", "n_key", "flags", "status", "result", "inlier_mask", "pair_idx"):
            setattr(v, f, getattr(self, f)[start:start + n])
        v.cert = self.cert[start:start + n] if self.cert is not None else None
        v.exact = None if self.exact is None else tuple(a[start:start + n] for a in self.exact)
        v._rands_host = None
        v._exact_buf = None
        return v


class Pipeline:
    """caelo_pipeline (include/caelo.h): `batch` consecutive frames share one launch of every front kernel, one
    encoder launch set and one match / RANSAC launch; the three stages of successive batches overlap on three HIP
    streams.  ``run`` submits K scans and returns without waiting for the GPU."""

    def __init__(self, eng, batch=4, buffers=3, max_points=None):
        self.eng = eng
        h = C.c_void_p()
        _ffi.check(eng.lib.caelo_pipeline_create(eng.ctx, int(batch), int(buffers), int(max_points or eng.max_points), C.byref(h)))
        self.h, self.batch, self.buffers = h, int(batch), int(buffers)
        self.pace = int(eng.lib.caelo_pipeline_get_pace(h))   # the library's default until set_pace (asked, not re-derived from the environment)

    def __del__(self):
        try:
            if self.h:
                self.eng.lib.caelo_pipeline_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def stats(self):
        """Host-side counters since the last call: jobs, us/frame the calling thread spent issuing launches, batches."""
        out = (C.c_int64 * 6)()
        _ffi.check(self.eng.lib.caelo_pipeline_stats(self.h, out))
        n = max(int(out[0]), 1)
        return {"jobs": int(out[0]), "issue_us_per_frame": out[1] / n / 1e3, "batches": int(out[2]), "batch": int(out[3]),
                "buffers": int(out[4]), "streams": int(out[5])}

    def _jobs(self, ptrs, counts, rands, prev, out, pairs, dist_channels, exact_voxels, dedup, certify=False, rands_host=None):
        """The run's jobs as one record array, filled column-wise, handed over in ONE foreign call (a ctypes call per frame
        costs ~10 us: 380 us for a 20-frame run, most of it before the first launch)."""
        k = len(ptrs)
        jobs = np.zeros(k, dtype=_ffi.JOB_DTYPE)
        idx = np.arange(k, dtype=np.uint64)
        jobs["pc"], jobs["n"] = ptrs, counts
        jobs["dist_channels"], jobs["mode"] = int(dist_channels), (1 if exact_voxels else 0) | (0 if dedup else 2)
        jobs["rows"] = out.rows.data_ptr() + idx * (MAX_K * 256)
        jobs["key_pixels"] = out.key_pixels.data_ptr() + idx * (MAX_K * 16)
        jobs["n_key"] = out.n_key.data_ptr() + idx * 4
        jobs["flags"] = out.flags.data_ptr() + idx * (MAX_K * 3)
        jobs["status"] = out.status.data_ptr() + idx * 16
        if pairs and k > 0:
            jobs["pair"] = _ffi.PAIR_CHAIN
            if prev is not None:
                assert prev.rows.is_contiguous()
                jobs["pair"][0], jobs["prev_rows"][0] = _ffi.PAIR_EXPLICIT, prev.rows.data_ptr()
                jobs["prev_n_key"][0] = prev.n_key.data_ptr() if prev.n_key is not None else 0
            else:
                jobs["pair"][0] = _ffi.PAIR_NONE
            jobs["rand"] = rands[:k] if isinstance(rands, np.ndarray) else [rands[i].data_ptr() for i in range(k)]   # (an array: device addresses)
        else:
            jobs["pair"] = _ffi.PAIR_NONE
        jobs["result"] = out.result.data_ptr() + idx * out.result.shape[1]
        jobs["inlier_mask"] = out.inlier_mask.data_ptr() + idx * MAX_K
        jobs["pair_idx"] = out.pair_idx.data_ptr() + idx * (MAX_K * 8)
        if certify and pairs and k > 0:
            # the exact RANSAC: certificates on the device, and (certify = "host", the default meaning of True) the pipeline's
            # certifier thread writes the exact results to host arrays while later batches run (include/caelo.h)
            if not getattr(self.eng, "_blas_bound", False):
                self.eng.host_blas()
                self.eng._blas_bound = True
            if certify == "device" or os.environ.get("CAELO_CERT_ZEROCOPY") == "0":
                jobs["cert"] = out.ensure_cert(self.eng).data_ptr() + idx * _ffi.CERT_DTYPE.itemsize
            if certify != "device":   # (the kernels write these pairs' certificates straight into the pipeline's pinned host memory)
                if out._exact_buf is None:   # host arrays of the exact results: allocated once per FrameBatch
                    out._exact_buf = (np.zeros(out.k, dtype=_ffi.POSE_DTYPE), np.zeros((out.k, MAX_K), dtype=np.uint8),
                                      np.zeros((out.k, 2), dtype=np.int32))
                res, masks, info = out._exact_buf
                info[:k, 0] = 0
                info[:k, 1] = 3                                  # "no record" until the certifier says otherwise
                has = jobs["pair"] != _ffi.PAIR_NONE
                out._has_pair = np.zeros(out.k, dtype=bool)
                out._has_pair[:k] = has
                jobs["result_host"] = np.where(has, res.ctypes.data + idx * _ffi.POSE_DTYPE.itemsize, 0)
                jobs["mask_host"] = np.where(has, masks.ctypes.data + idx * MAX_K, 0)
                jobs["info_host"] = np.where(has, info.ctypes.data + idx * 8, 0)
                if isinstance(rands_host, np.ndarray) and rands_host.dtype == np.uint64:   # host ADDRESSES of the draws (Pipeline.run_loaded: the loader's ring)
                    jobs["rand_host"] = rands_host[:k]
                elif rands_host is not None:   # (float64, contiguous, >= 6000 draws each: checked once per distinct array)
                    seen = {}
                    ptrs = np.empty(k, dtype=np.uint64)
                    for i in range(k):
                        r = rands_host[i]
                        a = seen.get(id(r))
                        if a is None:
                            assert isinstance(r, np.ndarray) and r.dtype == np.float64 and r.flags["C_CONTIGUOUS"] and r.size >= 6000
                            a = seen[id(r)] = r.ctypes.data
                        ptrs[i] = a
                    jobs["rand_host"] = ptrs
                    out._rands_host = rands_host
                out.exact = (res, masks, info[:, 0], info[:, 1])
        return jobs

    def wait_encoded(self, stream):
        """caelo_pipeline_wait_encoded: ``stream`` (a torch stream) waits for the rows of every frame of the batches issued so far."""
        _ffi.check(self.eng.lib.caelo_pipeline_wait_encoded(self.h, C.c_void_p(stream.cuda_stream)))

    def sync_encoded(self, lag=0):
        """caelo_pipeline_sync_encoded: the calling thread waits until the rows of every batch issued so far, except the last
        ``lag``, are written (no device-side wait is left in any queue)."""
        _ffi.check(self.eng.lib.caelo_pipeline_sync_encoded(self.h, int(lag)))

    def run(self, scans, rands=None, prev=None, dist_channels=5, exact_voxels=False, out=None, pairs=True, dedup=True, on_batch=None,
            on_encoded=None, certify=False, rands_host=None, publish=True):
        """scans: K device tensors [n,4] f32; rands: K device tensors of RANSAC draws ([1500,4] f64).
        Frame i is matched against frame i-1 (pose in ``result[i]``); frame 0 against ``prev``
        (FrameFeatures) when given; ``pairs=False`` extracts only (BASELINE configs[1]).  Returns a FrameBatch; the
        current stream has waited for all lanes.  ``certify=True``: the exact RANSAC -- every pair leaves its certificate
        (``out.cert``) and the pipeline's certifier thread runs the host half (csrc/certify.hip) on it while later batches are on
        the GPU; when ``run`` returns, ``out.exact`` = (results record array [k], masks [k,1024] u8, evals [k], status [k]) holds
        the reference's inlier sets, R_star / T_star and refits bit for bit (frames without a pair: status 3) and
        ``out.result`` / ``out.inlier_mask`` on the device are overwritten with them.  ``rands_host``: host copies of the draws
        (only read for a pair that escalates beyond 0.4 m; fetched from the device otherwise).  ``certify="device"``: certificates
        only (``Engine.certify_batch`` runs the host half later).  ``publish=False``: the exact results stay in ``out.exact`` (host
        arrays, where the reference's own results live) and ``out.result`` / ``out.inlier_mask`` on the device are NOT written at all
        (with the host half active the kernels stop at the certificate: no k_ransac_finish).  ``on_encoded(lo, hi)`` is called, in order, once the rows of frames [lo, hi) are
        WRITTEN (the calling thread has waited for them: caelo_pipeline_sync_encoded, one batch behind the issue) -- what it
        enqueues on any stream may read them at once; a caller ships finished rows that way while later batches run.
        ``on_batch(lo, hi)`` is called right after frames [lo, hi) have been issued (with ``wait_encoded`` the device-side form of
        the same hand-over, which costs the pipeline a quarter of its rate: DESIGN.md 6)."""
        eng, lib, k = self.eng, self.eng.lib, len(scans)
        out = out or FrameBatch(eng, k)
        assert out.k >= k and (not pairs or len(rands) >= k)
        stream = eng.stream
        per_batch = on_batch is not None or on_encoded is not None
        # an even batch plan for runs that are not whole batches (caelo.h); with a per-batch callback: full batches, remainder last
        _ffi.check(lib.caelo_pipeline_expect(self.h, 0 if per_batch else k))
        _ffi.check(lib.caelo_pipeline_begin(self.h, stream))
        for pc in scans:
            assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] == 4 and pc.is_contiguous()
        _t0 = time.perf_counter()
        jobs = self._jobs([pc.data_ptr() for pc in scans], [pc.shape[0] for pc in scans], rands, prev, out, pairs, dist_channels,
                          exact_voxels, dedup, certify, rands_host)
        _t1 = time.perf_counter()
        tail = None   # a partial last batch is only issued by the flush: its callbacks come after that
        issued = []   # batches issued, not yet reported to on_encoded
        try:
            if not per_batch:
                _ffi.check(lib.caelo_pipeline_submit_many(self.h, jobs.ctypes.data, k))
            else:
                for lo in range(0, k, self.batch):
                    hi = min(k, lo + self.batch)
                    _ffi.check(lib.caelo_pipeline_submit_many(self.h, jobs[lo:hi].ctypes.data, hi - lo))
                    if hi - lo == self.batch:
                        if on_batch:
                            on_batch(lo, hi)
                        issued.append((lo, hi))
                        if on_encoded and len(issued) > 1:
                            self.sync_encoded(1)      # (returns at once under the default pacing of the issue)
                            while len(issued) > 1:
                                on_encoded(*issued.pop(0))
                    else:
                        tail = (lo, hi)
        finally:
            rc = lib.caelo_pipeline_flush(self.h, stream)
        _ffi.check(rc)
        if tail:
            if on_batch:
                on_batch(*tail)
            issued.append(tail)
        if on_encoded and issued:
            self.sync_encoded(0)
            for lo, hi in issued:
                on_encoded(lo, hi)
        _t2 = time.perf_counter()
        if publish:
            self._publish_exact(out, k, certify, pairs)
        elif certify and certify != "device" and pairs and k > 0 and (out.exact[3][:k] == 2).any():
            raise _ffi.CaeloError("a pair holds more than 1024 matches: no certificate")
        self.last_times = {"jobs_ms": 1e3 * (_t1 - _t0), "submit_flush_ms": 1e3 * (_t2 - _t1), "publish_ms": 1e3 * (time.perf_counter() - _t2)}
        return out

    def _publish_exact(self, out, k, certify, pairs):
        """After the flush of a certified run: the certifier thread has written every exact result (caelo_pipeline_flush waits
        for it); put them into the device tensors too."""
        if not (certify and certify != "device" and pairs and k > 0):
            return
        res, masks, evals, status = out.exact
        if (status[:k] == 2).any():
            raise _ffi.CaeloError("a pair holds more than 1024 matches: no certificate")
        has = getattr(out, "_has_pair", None)
        if has is not None and (status[:k][has[:k]] != 0).any():   # (the flush reports it too: a job WITH a pair must come back exact)
            bad = np.flatnonzero(has[:k] & (status[:k] != 0))
            raise _ffi.CaeloError("pairs of frames %s came back without an exact result (status %s)" % (bad.tolist(), status[:k][bad].tolist()))
        ok = status[:k] == 0
        if ok.any():
            with torch.cuda.stream(torch.cuda.current_stream(self.eng.device)):
                if ok.all():
                    out.result[:k].copy_(torch.from_numpy(res[:k].view(np.uint8).reshape(k, -1)))
                    out.inlier_mask[:k].copy_(torch.from_numpy(masks[:k]))
                else:
                    sel = torch.from_numpy(np.flatnonzero(ok)).to(self.eng.device)
                    out.result[sel] = torch.from_numpy(np.ascontiguousarray(res[:k].view(np.uint8).reshape(k, -1)[ok])).to(self.eng.device)
                    out.inlier_mask[sel] = torch.from_numpy(np.ascontiguousarray(masks[:k][ok])).to(self.eng.device)

    def cert_stats(self):
        """Host half inside the pipeline since the last call: pairs certified, hypotheses evaluated per pair, certifier-thread us per pair."""
        o = (C.c_int64 * 4)()
        _ffi.check(self.eng.lib.caelo_pipeline_cert_stats(self.h, o))
        n = max(int(o[0]), 1)
        return {"pairs": int(o[0]), "evals_per_pair": o[1] / n, "host_us_per_pair": o[2] / n / 1e3, "handover_us_per_pair": o[3] / n / 1e3}

    def set_pace(self, lag):
        """caelo_pipeline_set_pace: after issuing a batch the calling thread waits for the encoder of the batch ``lag`` before it
        (-1: never)."""
        _ffi.check(self.eng.lib.caelo_pipeline_set_pace(self.h, int(lag)))
        self.pace = int(lag)

    def run_uploading(self, host_scans, rands=None, prev=None, dist_channels=5, out=None, pairs=True, dedup=True, ahead=4, certify=False,
                      rands_host=None):
        """``run`` for scans that live in (pinned) HOST memory: a copy stream uploads batch b + ``ahead`` while the pipeline works on
        batch b, into ``ahead + 2`` sets of device buffers -- the overlap of the reference's producer process, which prepares
        frame i + 1 while frame i is matched (PoseEstimation.py:214-245).  The hand-overs are paced by the calling thread, not by
        waits in the device queues: per batch it issues the launches, then the copies of the batch ``ahead`` further on, then waits
        for the encoder of the batch before (caelo_pipeline_sync_encoded: the GPU keeps one batch queued; the buffers the next
        copies overwrite were read by a front stage at least two batches back) and for the arrival of the next batch's scans (an
        event on the copy stream).  ``ahead``: 13.5 / 15.3 / 15.9 / 16.1 k frames/s for 1 / 2 / 3 / 4 (18.6 k resident; an arrival is
        late by up to 0.3 ms now and then, and a batch of scans is 17 MB of device memory).  Device-side waits for the same hand-overs (caelo_pipeline_wait_stream / _release_scans) cost
        8 - 15 % of the resident rate EACH, however rarely they were issued."""
        te0_ = time.perf_counter()
        eng, lib, k, B = self.eng, self.eng.lib, len(host_scans), self.batch
        out = out or FrameBatch(eng, k)
        assert out.k >= k and (not pairs or len(rands) >= k) and ahead >= 1
        for pc in host_scans:
            assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] == 4 and pc.is_contiguous()   # (host tensors, pinned for overlap; device tensors work too)
        big = max(int(pc.shape[0]) for pc in host_scans)
        slots = ahead + 2
        nb = (k + B - 1) // B
        src_p = np.array([pc.data_ptr() for pc in host_scans], dtype=np.uint64)
        nbytes = np.array([int(pc.shape[0]) * 16 for pc in host_scans], dtype=np.uint64)
        on_host = all(not pc.is_cuda for pc in host_scans)
        # A producer that leaves the scans of a batch in ONE pinned block at a fixed pitch (run_sequence.py's loader does) gets one
        # copy command per batch: the device slots mirror the pitch.  Measured (tools/upload_contention_probe.py): eight 2 MB copies per
        # batch cost a resident pipeline running beside them 20 % (41 GB/s), one 16 MB copy 0-3 % (54 GB/s) -- the commands, not the
        # bytes, are what the copies cost.
        pitch = 0
        if on_host and k > 1:
            d = np.diff(src_p.astype(np.int64))
            inside = np.ones(k - 1, bool); inside[B - 1::B] = False          # (the step from one batch to the next may be anything)
            if inside.any() and (d[inside] == d[inside][0]).all() and int(d[inside][0]) >= int(nbytes.max()) and int(d[inside][0]) % 16 == 0:
                # ... and the scans of a batch must be views of ONE allocation that spans the whole copy: equal spacing alone would also
                # be true of separately pinned tensors that happen to sit at a fixed distance, and one copy over them would read the gaps
                stor = [(pc.untyped_storage().data_ptr(), pc.untyped_storage().nbytes()) for pc in host_scans]
                one_block = all(len({stor[i][0] for i in range(lo, min(k, lo + B))}) == 1 and
                                int(src_p[min(k, lo + B) - 1] + nbytes[min(k, lo + B) - 1]) <= stor[lo][0] + stor[lo][1]
                                for lo in range(0, k, B))
                if one_block:
                    pitch = int(d[inside][0])
        key = ("upload", B, slots, pitch)
        st = self._upload.get(key) if hasattr(self, "_upload") else None
        if st is None or st[1] < big:
            # (sized once: by the pitch, or by the engine's scan capacity -- a later call with a slightly longer scan must not find
            # itself allocating device memory behind the caller's clock)
            rows = pitch // 16 if pitch else max(big, int(eng.max_points))
            if pitch:
                blocks = [torch.empty((B * pitch,), dtype=torch.uint8, device=eng.device) for _ in range(slots)]
                bufs = [[blk[i * pitch:(i + 1) * pitch] for i in range(B)] for blk in blocks]
            else:
                bufs = [[torch.empty((rows, 4), dtype=torch.float32, device=eng.device) for _ in range(B)] for _ in range(slots)]
            st = (bufs, rows, torch.cuda.Stream(device=eng.device))
            self._upload = {key: st}
        bufs, _, copy = st
        stream = eng.stream
        jobs = self._jobs([bufs[(i // B) % slots][i % B].data_ptr() for i in range(k)], [int(pc.shape[0]) for pc in host_scans], rands, prev, out,
                          pairs, dist_channels, False, dedup, certify, rands_host)
        arrived = [torch.cuda.Event() for _ in range(nb)]

        # the copies of a batch go out behind ONE native call (caelo_upload_many): eight sliced torch copies cost the issuing thread
        # ~150 us per batch, and that thread's time per batch is what bounds this mode
        dst_p = np.array([bufs[(i // B) % slots][i % B].data_ptr() for i in range(k)], dtype=np.uint64)
        copy_h = C.c_void_p(copy.cuda_stream)
        if pitch:   # per batch: (first frame's slot, first frame's source, bytes up to the end of the last frame)
            one_n = np.array([int(src_p[min(k, (b + 1) * B) - 1] + nbytes[min(k, (b + 1) * B) - 1] - src_p[b * B]) for b in range(nb)], dtype=np.uint64)

        up_calls = []

        def upload(b):   # into the slot batch b - slots used
            lo, hi = b * B, min(k, (b + 1) * B)
            if pitch:
                tu0_ = time.perf_counter()
                _ffi.check(lib.caelo_upload_many(dst_p[lo:lo + 1].ctypes.data, src_p[lo:lo + 1].ctypes.data, one_n[b:b + 1].ctypes.data, 1, copy_h))
                up_calls.append(1e6 * (time.perf_counter() - tu0_))
            elif on_host:
                _ffi.check(lib.caelo_upload_many(dst_p[lo:hi].ctypes.data, src_p[lo:hi].ctypes.data, nbytes[lo:hi].ctypes.data, hi - lo, copy_h))
            else:   # (device sources: the probe that separates the protocol's cost from PCIe's)
                with torch.cuda.stream(copy):
                    for i in range(lo, hi):
                        bufs[b % slots][i % B][:host_scans[i].shape[0]].copy_(host_scans[i], non_blocking=True)
            arrived[b].record(copy)

        if pitch:
            # one copy command per batch: the whole paced loop natively (caelo_pipeline_run_uploading, round 6 -- from Python the
            # interpreter's time between the pacing wait and the next batch's front launches cost the mode 20 % of its rate)
            dstb, srcb = np.ascontiguousarray(dst_p[::B]), np.ascontiguousarray(src_p[::B])
            tns = (C.c_int64 * 4)()
            tf0_ = time.perf_counter()
            _ffi.check(lib.caelo_pipeline_run_uploading(self.h, C.c_void_p(jobs.ctypes.data), k, nb, None, 0, C.c_void_p(dstb.ctypes.data),
                                                        C.c_void_p(srcb.ctypes.data), C.c_void_p(one_n.ctypes.data), slots, None, 0, int(ahead), copy_h, stream, tns))
            tf1_ = time.perf_counter()
            self._publish_exact(out, k, certify, pairs)
            self.last_upload_times = dict(wait_arrival_ms=tns[1] / 1e6, submit_ms=tns[2] / 1e6, upload_issue_and_wait_encoded_ms=tns[3] / 1e6, native_loop_ms=1e3 * (tf1_ - tf0_),
                                          publish_ms=1e3 * (time.perf_counter() - tf1_), prepare_ms=1e3 * (tf0_ - te0_))
            return out
        _ffi.check(lib.caelo_pipeline_expect(self.h, 0))   # full batches, the remainder last: the slots are laid out that way
        copy.wait_stream(torch.cuda.current_stream(eng.device))   # (an earlier run may still read the slots)
        pace = self.pace
        _ffi.check(lib.caelo_pipeline_set_pace(self.h, -1))       # this loop paces itself: the copies go out BEFORE the thread waits
        te1_ = time.perf_counter()
        _ffi.check(lib.caelo_pipeline_begin(self.h, stream))
        te2_ = time.perf_counter()
        try:
            for b in range(min(ahead, nb)):
                upload(b)
            tw = [0.0, 0.0, 0.0, 0.0]
            for b in range(nb):
                t0_ = time.perf_counter()
                arrived[b].synchronize()                 # batch b's scans are in device memory
                t1_ = time.perf_counter()
                lo, hi = b * B, min(k, (b + 1) * B)
                _ffi.check(lib.caelo_pipeline_submit_many(self.h, jobs[lo:hi].ctypes.data, hi - lo))
                t2_ = time.perf_counter()
                if b + ahead < nb:
                    upload(b + ahead)                    # slot of batch b - 2: encoded (hence read) before batch b was issued
                t3_ = time.perf_counter()
                if hi - lo == B:
                    self.sync_encoded(1)       # (a partial last batch is only issued by the flush)
                t4_ = time.perf_counter()
                tw[0] += t1_ - t0_; tw[1] += t2_ - t1_; tw[2] += t3_ - t2_; tw[3] += t4_ - t3_
            self.last_upload_times = dict(wait_arrival_ms=1e3 * tw[0], submit_ms=1e3 * tw[1], upload_issue_ms=1e3 * tw[2], wait_encoded_ms=1e3 * tw[3])
        finally:
            tf0_ = time.perf_counter()
            rc = lib.caelo_pipeline_flush(self.h, stream)
            lib.caelo_pipeline_set_pace(self.h, pace)
        _ffi.check(rc)
        tf1_ = time.perf_counter()
        self._publish_exact(out, k, certify, pairs)
        self.last_upload_times['copy_call_us'] = [round(x) for x in up_calls[:40]]
        self.last_upload_times.update(flush_ms=1e3 * (tf1_ - tf0_), publish_ms=1e3 * (time.perf_counter() - tf1_), prepare_ms=1e3 * (te1_ - te0_), begin_ms=1e3 * (te2_ - te1_))
        return out


class SeqLoader:
    """caelo_seqloader (include/caelo.h, csrc/seqload.hip): native threads read the scan files of a sequence batch after batch into a
    pinned ring and generate each pair's RANSAC draws (NumPy's MT19937 stream, bit for bit).  ``Pipeline.run_loaded`` consumes it."""

    def __init__(self, eng, paths, first_frame=0, batch=8, seed_base=1000, threads=16, ring=10, keep=96, cap=None):
        import os
        self.eng, lib = eng, eng.lib
        self.batch, self.cap, self.n = int(batch), int(cap or eng.max_points), len(paths)
        self.ring, self.keep_n = int(ring), int(keep)
        self.n_batches = (self.n + self.batch - 1) // self.batch
        self.slot_bytes = int(lib.caelo_seqloader_slot_bytes(self.batch, self.cap))
        self.scan_bytes = self.batch * self.cap * 16
        self.ring_h = torch.empty((self.ring, self.slot_bytes), dtype=torch.uint8, pin_memory=True)
        self.keep_h = np.empty((self.keep_n, self.batch, 6000), np.float64)
        self._arr = (C.c_char_p * self.n)(*[os.fsencode(p_) for p_ in paths])
        h = C.c_void_p()
        _ffi.check(lib.caelo_seqloader_create(self._arr, self.n, int(first_frame), self.batch, self.ring, self.cap, C.c_void_p(self.ring_h.data_ptr()),
                                              C.c_void_p(self.keep_h.ctypes.data), self.keep_n, int(seed_base), int(threads), C.byref(h)))
        self.h = h
        self._slot = C.c_int32(0)
        self._npts = (C.c_int64 * self.batch)()

    def wait(self, b):
        """-> (ring slot, point counts of the batch's frames [batch] int64); blocks until batch b is loaded."""
        _ffi.check(self.eng.lib.caelo_seqloader_wait(self.h, int(b), C.byref(self._slot), self._npts))
        return int(self._slot.value), np.frombuffer(self._npts, dtype=np.int64).copy()

    def release(self, b):
        _ffi.check(self.eng.lib.caelo_seqloader_release(self.h, int(b)))

    def stats(self):
        o = (C.c_int64 * 3)()
        _ffi.check(self.eng.lib.caelo_seqloader_stats(self.h, o))
        return {"read_s": o[0] / 1e9, "draws_s": o[1] / 1e9, "wait_slot_s": o[2] / 1e9}

    def close(self):
        if self.h:
            self.eng.lib.caelo_seqloader_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _run_loaded(self, loader, b0, nb, prev=None, out=None, pairs=True, dist_channels=5, dedup=True, certify=True, ahead=4, publish=True):
    """Batches [b0, b0 + nb) of a SeqLoader through the pipeline: a batch's scans AND draws go up behind ONE copy command (the loader's
    slot layout, mirrored on the device), the jobs are built column-wise for the whole call, and nothing here is per-frame Python.
    Paced like ``run_uploading`` (the copies go out before the thread waits for the encoder).  -> (FrameBatch, frames)."""
    eng, lib, B = self.eng, self.eng.lib, self.batch
    assert loader.batch == B and ahead >= 1 and b0 + nb <= loader.n_batches
    k = min(loader.n - b0 * B, nb * B)
    out = out or FrameBatch(eng, k)
    assert out.k >= k
    slots = ahead + 2
    key = ("loaded", B, slots, loader.slot_bytes)
    st = getattr(self, "_loaded", {}).get(key)
    if st is None:
        st = ([torch.empty((loader.slot_bytes,), dtype=torch.uint8, device=eng.device) for _ in range(slots)], torch.cuda.Stream(device=eng.device))
        self._loaded = {key: st}
    dslots, copy = st
    f = np.arange(k, dtype=np.uint64)
    lb, j = f // B, f % B                                # local batch, frame within it
    base = np.array([d.data_ptr() for d in dslots], dtype=np.uint64)[((b0 + lb.astype(np.int64)) % slots)]
    pcs = base + j * np.uint64(loader.cap * 16)
    rnd = base + np.uint64(loader.scan_bytes) + j * np.uint64(6000 * 8)
    rnd_h = np.uint64(loader.keep_h.ctypes.data) + (((b0 + lb) % np.uint64(loader.keep_n)) * np.uint64(B) + j) * np.uint64(6000 * 8)
    if not getattr(eng, "_blas_bound", False) and certify:
        eng.host_blas()
        eng._blas_bound = True
    tj0_ = time.perf_counter()
    jobs = self._jobs(pcs, np.zeros(k, np.int64), rnd, prev, out, pairs, dist_channels, False, dedup, certify, rnd_h if certify else None)
    tj1_ = time.perf_counter()
    stream = eng.stream
    # the whole paced loop natively (caelo_pipeline_run_uploading: wait for a batch's arrival, submit it, queue the copy `ahead` further
    # on, stay one batch behind the encoder) -- from Python the ~60 us between the pacing wait and the next front launches cost 20 %
    dst = (C.c_void_p * slots)(*[d.data_ptr() for d in dslots])
    tns = (C.c_int64 * 4)()
    tf0_ = time.perf_counter()
    _ffi.check(lib.caelo_pipeline_run_uploading(self.h, C.c_void_p(jobs.ctypes.data), k, nb, loader.h, int(b0), dst, None, None, slots,
                                                C.c_void_p(loader.ring_h.data_ptr()), loader.slot_bytes, int(ahead), C.c_void_p(copy.cuda_stream), stream, tns))
    tw = [tns[0] / 1e9, tns[1] / 1e9, tns[2] / 1e9, tns[3] / 1e9]
    tf1_ = time.perf_counter()
    if publish:
        self._publish_exact(out, k, certify, pairs)
    elif certify and pairs and k > 0:   # (the exact results stay on the host, where the caller reads them: out.exact)
        has = getattr(out, "_has_pair", None)
        st_ = out.exact[3][:k]
        if (st_ == 2).any() or (has is not None and (st_[has[:k]] != 0).any()):
            raise _ffi.CaeloError("pairs of frames %s came back without an exact result" % np.flatnonzero(st_ != 0).tolist()[:8])
    self.last_loaded_times = dict(starved_s=tw[0], wait_arrival_s=tw[1], submit_s=tw[2], upload_and_pace_s=tw[3], jobs_s=tj1_ - tj0_,
                                  flush_s=tf1_ - tf0_, publish_s=time.perf_counter() - tf1_)
    return out, k


Pipeline.run_loaded = _run_loaded


class Engine:
    def __init__(self, respond_h5=RESPOND_H5, encoder_h5=ENCODER_H5, device=None, max_points=160000):
        if not torch.cuda.is_available():
            raise _ffi.CaeloError("no HIP device visible: libcaelo has no CPU fallback")
        self.lib = _ffi.load()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        _ffi.check(self.lib.caelo_create(C.byref(ctx), self.device_index))
        self.ctx = ctx
        self.max_points = int(max_points)
        self._maps = {}
        self._wss = {}      # (HIP stream, kind) -> scratch bytes: workspaces and voxel maps are per stream, so
                            # frames issued on different streams ("lanes") run concurrently without sharing scratch
        if respond_h5:
            self.load_weights(respond_h5)
        if encoder_h5:
            self.load_weights(encoder_h5)

    def __del__(self):
        try:
            self._maps.clear()
            if self.ctx:
                self.lib.caelo_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ---- plumbing ----------------------------------------------------------------------------
    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def load_weights(self, path):
        kind, ws = read_keras_weights(path)
        if kind == "respond":
            _ffi.check(self.lib.caelo_set_respond_weights(self.ctx, *[_hptr(w) for w in ws]))
        else:
            _ffi.check(self.lib.caelo_set_encoder_weights(self.ctx, *[_hptr(w) for w in ws]))
        return kind

    def _ws(self, kind, need):
        """Scratch bytes of the current stream for one entry point (grown on demand, never shared across streams)."""
        key = (torch.cuda.current_stream(self.device).cuda_stream, kind)
        t = self._wss.get(key)
        if t is None or t.numel() < need:
            # zero-filled: caelo_match / caelo_ransac keep their tickets in the workspace and leave them zero
            t = self._wss[key] = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        return t

    def set_encoder_reference(self, on=True):
        """caelo_set_encoder_reference: stage 1 of this engine's encoder = the exact-f32 kernel (precision reference; slower)."""
        _ffi.check(self.lib.caelo_set_encoder_reference(self.ctx, 1 if on else 0))

    def lane_faults(self):
        """caelo_lane_faults: wavefronts of the pose kernels whose lanes disagreed on a hypothesis they all derive from the
        same inputs, plus descriptors the encoder wrote that are not a finite value within [-1, 1] (self-checks of the two
        halves; synchronises).  0 on healthy hardware with sane weights."""
        out = C.c_int64(0)
        _ffi.check(self.lib.caelo_lane_faults(self.ctx, C.byref(out)))
        return int(out.value)

    def pipeline(self, batch=4, buffers=3):
        """The native frame executor (caelo_pipeline): `batch` frames per launch, `buffers` batches of patches in flight
        between the front and the encoder; created once per configuration."""
        key = ("pipeline", int(batch), int(buffers))
        if key not in self._maps:
            self._maps[key] = Pipeline(self, batch, buffers)
        return self._maps[key]

    def voxmap(self, max_points=None, slot=0):
        n = self.max_points if max_points is None else int(max_points)
        key = (torch.cuda.current_stream(self.device).cuda_stream, slot, n)
        if key not in self._maps:
            self._maps[key] = VoxelMap(self, n)
        return self._maps[key]

    def _encode_ws(self, n_patches):
        return self._ws("encode", int(self.lib.caelo_encode_ws_bytes(int(n_patches))))

    # ---- stages (device tensors in, device tensors out, no sync) ---------------------------------
    def project(self, pc, status=None):
        assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] == 4 and pc.is_contiguous()
        ring = self.empty((RING_H, RING_W, RING_C), torch.float32)
        counter = self.empty((RING_H, RING_W), torch.int32)
        winner = self.empty((RING_H * RING_W,), torch.int32)
        status = self.zeros((1,), torch.int32) if status is None else status
        _ffi.check(self.lib.caelo_project(self.ctx, _ptr(pc), pc.shape[0], _ptr(ring), _ptr(counter), _ptr(winner),
                                          _ptr(status), self.stream))
        return ring, counter, status

    def respond(self, img):
        """img [rows>=64, w>=1792, c>=3] f32 -> [64,1792,8]."""
        assert img.dtype == torch.float32 and img.dim() == 3 and img.is_contiguous()
        resp = self.empty((NET_H, NET_W, 8), torch.float32)
        _ffi.check(self.lib.caelo_respond(self.ctx, _ptr(img), img.shape[1], img.shape[2], _ptr(resp), self.stream))
        return resp

    def keypoints(self, ring, counter, resp, status=None):
        assert ring.dtype == torch.float32 and counter.dtype == torch.int32 and resp.dtype == torch.float32
        assert ring.is_contiguous() and counter.is_contiguous() and resp.is_contiguous()
        kpix = self.zeros((MAX_K, 2), torch.int64)
        kpts = self.zeros((MAX_K, 3), torch.float32)
        nkey = self.empty((1,), torch.int32)
        status = self.zeros((1,), torch.int32) if status is None else status
        _ffi.check(self.lib.caelo_keypoints(self.ctx, _ptr(ring), ring.shape[1], ring.shape[2], _ptr(counter),
                                            counter.shape[1], _ptr(resp), _ptr(self._ws("keypoints", self.lib.caelo_keypoints_ws_bytes())),
                                            _ptr(kpix), _ptr(kpts),
                                            _ptr(nkey), _ptr(status), self.stream))
        return kpts, kpix, nkey, status

    def extend_keypts(self, ring, counter, key_pixels, n_key=None):
        """ExtendKeyPtsInShpericalRing on device tensors: ring [H,W,C] f32, counter [Hc,Wc] i32 (MUTATED: the windows
        are zeroed like SphericalRing.py:307), key_pixels [K,2] i64 -> (ext_pts [K*169,3] f32, n_ext [1] i32)."""
        assert ring.dtype == torch.float32 and counter.dtype == torch.int32 and key_pixels.dtype == torch.int64
        assert ring.is_contiguous() and counter.is_contiguous() and key_pixels.is_contiguous()
        rows, cols = min(ring.shape[0], counter.shape[0]), min(ring.shape[1], counter.shape[1])
        k = key_pixels.shape[0]
        ext = self.empty((k * 169, 3), torch.float32)
        n_ext = self.empty((1,), torch.int32)
        ws = self._ws("extend", int(self.lib.caelo_extend_ws_bytes(rows, cols)))
        _ffi.check(self.lib.caelo_extend_keypts(self.ctx, _ptr(ring), ring.shape[1], ring.shape[2], _ptr(counter), counter.shape[1],
                                                rows, cols, _ptr(key_pixels), k, _ptr(n_key), _ptr(ext), _ptr(n_ext), _ptr(ws),
                                                self.stream))
        return ext, n_ext

    def icp_step(self, pc0, pc1, threshold, min_inliers=100):
        """One ICP iteration (caelo_icp_step): pc1 [n1,3] f32 is moved IN PLACE.  -> (rt [12] f32, n_inliers [1] i32)."""
        for t in (pc0, pc1):
            assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 3 and t.is_contiguous()
        rt = self.empty((12,), torch.float32)
        n_in = self.empty((1,), torch.int32)
        ws = self._ws("icp", int(self.lib.caelo_icp_ws_bytes(pc1.shape[0])))
        _ffi.check(self.lib.caelo_icp_step(self.ctx, _ptr(pc0), pc0.shape[0], _ptr(pc1), pc1.shape[0], float(threshold),
                                           int(min_inliers), _ptr(rt), _ptr(n_in), _ptr(ws), self.stream))
        return rt, n_in

    def icp(self, pc0, pc1, planar0=None, planar1=None, **kw):
        """The reference's ICP loops entirely on the device (caelo_icp): pc1 [n1,3] (and planar1 [m1,6]) are MOVED in place.
        kw: threshold0/1, decay0/1, small_shift, ep, max_iter, min_iter, min_pairs, fail_only_first.  -> IcpResult bytes tensor
        (read it with ``icp_result``: the only synchronisation)."""
        for t in (pc0, pc1):
            assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 3 and t.is_contiguous()
        use_planar = planar0 is not None
        if use_planar:
            for t in (planar0, planar1):
                assert t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous() and (t.shape[0] == 0 or t.shape[1] == 6)
        prm = _ffi.IcpParams(threshold0=kw.get("threshold0", 0.5), threshold1=kw.get("threshold1", 2.0), decay0=kw.get("decay0", 0.9),
                             decay1=kw.get("decay1", 0.5), small_shift=kw.get("small_shift", 0.05), ep=kw.get("ep", 0.001),
                             max_iter=kw.get("max_iter", 50), min_iter=kw.get("min_iter", 19), min_pairs=kw.get("min_pairs", 100),
                             fail_only_first=kw.get("fail_only_first", 0), use_planar=1 if use_planar else 0, reserved=0)
        m0 = planar0.shape[0] if use_planar else 0
        m1 = planar1.shape[0] if use_planar else 0
        res = self.empty((C.sizeof(_ffi.IcpResult),), torch.uint8)
        ws = self._ws("icp_loop", int(self.lib.caelo_icp_loop_ws_bytes(pc1.shape[0], m1)))
        _ffi.check(self.lib.caelo_icp(self.ctx, _ptr(pc0), pc0.shape[0], _ptr(pc1), pc1.shape[0], _ptr(planar0) if m0 else None, m0,
                                      _ptr(planar1) if m1 else None, m1, C.byref(prm), _ptr(res), _ptr(ws), self.stream))
        return res

    @staticmethod
    def icp_result(res):
        """Synchronising read of a caelo_icp_result."""
        return _ffi.IcpResult.from_buffer_copy(res.cpu().numpy().tobytes())

    def voxelize(self, pc, vmap=None, status=None):
        assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] >= 3 and pc.is_contiguous()
        vmap = vmap or self.voxmap(max(self.max_points, pc.shape[0]))
        status = self.zeros((1,), torch.int32) if status is None else status
        _ffi.check(self.lib.caelo_voxelize(self.ctx, vmap.h, _ptr(pc), pc.shape[0], pc.shape[1], _ptr(status), self.stream))
        return vmap, status

    def voxelize_fast(self, pc, vmap=None, status=None):
        """The one-pass build ``extract`` uses (caelo_voxelize_fast): same voxel sets as ``voxelize``, no first-touch order."""
        assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] >= 3 and pc.is_contiguous()
        vmap = vmap or self.voxmap(max(self.max_points, pc.shape[0]))
        status = self.zeros((1,), torch.int32) if status is None else status
        _ffi.check(self.lib.caelo_voxelize_fast(self.ctx, vmap.h, _ptr(pc), pc.shape[0], pc.shape[1], _ptr(status), self.stream))
        return vmap, status

    def voxmap_voxels(self, vmap, scale, capacity=1 << 18):
        """Voxel set of one scale of a device voxel map -> sorted int32 [n,3] (host; caelo_voxmap_dump, synchronises)."""
        keys = self.empty((capacity,), torch.int64)
        bits = self.empty((capacity, 8), torch.int64)
        count = self.empty((1,), torch.int32)
        _ffi.check(self.lib.caelo_voxmap_dump(self.ctx, vmap.h, int(scale), _ptr(keys), _ptr(bits), capacity, _ptr(count), self.stream))
        n = int(count.item())
        assert n <= capacity, "raise capacity"
        k = keys[:n].cpu().numpy().astype(np.uint64)
        b = np.unpackbits(bits[:n].cpu().numpy().view(np.uint8).reshape(n, 8, 8), axis=2, bitorder="little")  # [brick, x&7, y&7, z&7]
        br, x, y, z = np.nonzero(b.reshape(n, 8, 8, 8))
        base = np.stack([(k >> np.uint64(40)) & np.uint64(0xFFFFF), (k >> np.uint64(20)) & np.uint64(0xFFFFF), k & np.uint64(0xFFFFF)], 1).astype(np.int64) * 8
        vox = (base[br] + np.stack([x, y, z], 1)).astype(np.int32)
        return vox[np.lexsort((vox[:, 2], vox[:, 1], vox[:, 0]))]

    def voxmap_export(self, vmap, capacity):
        outs = [self.empty((capacity, 3), torch.int16) for _ in range(3)]
        counts = self.empty((3,), torch.int64)
        _ffi.check(self.lib.caelo_voxmap_export(self.ctx, vmap.h, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                                capacity, _ptr(counts), self.stream))
        n = counts.cpu().tolist()          # (the call itself does not wait for the device; this read does)
        if max(n) > capacity:
            raise _ffi.CaeloError("caelo_voxmap_export: %d voxels exceed the output capacity %d" % (max(n), capacity))
        return [o[:k] for o, k in zip(outs, n)]

    def voxmap_order(self, vmap, scale_mask=7):
        """caelo_voxmap_order: the map (filled by ``voxelize``) records its voxel lists' first-touch order on the device, so that
        ``patches`` on it redoes tie-split patches of the scales in ``scale_mask`` in scikit-learn's kd-tree order.  No sync."""
        _ffi.check(self.lib.caelo_voxmap_order(self.ctx, vmap.h, int(scale_mask), self.stream))
        return vmap

    def voxmap_from_lists(self, a0, a1, a2, vmap=None, status=None):
        for a in (a0, a1, a2):
            assert a.dtype == torch.int16 and a.dim() == 2 and a.shape[1] == 3 and a.is_contiguous()
        vmap = vmap or self.voxmap(max(self.max_points, a0.shape[0], a1.shape[0], a2.shape[0]), slot=1)
        status = self.zeros((1,), torch.int32) if status is None else status
        _ffi.check(self.lib.caelo_voxmap_from_lists(self.ctx, vmap.h, _ptr(a0), a0.shape[0], _ptr(a1), a1.shape[0],
                                                    _ptr(a2), a2.shape[0], _ptr(status), self.stream))
        return vmap, status

    def patches(self, vmap, pts, n_key=None, status=None):
        """-> bits [K,3,64] int64 (bit-packed 16^3 patches), flags [K,3] uint8."""
        assert pts.dtype == torch.float32 and pts.dim() == 2 and pts.shape[1] == 3 and pts.is_contiguous()
        k = pts.shape[0]
        bits = self.empty((k, 3, 64), torch.int64)
        flags = self.empty((k, 3), torch.uint8)
        status = self.zeros((1,), torch.int32) if status is None else status
        _ffi.check(self.lib.caelo_patches(self.ctx, vmap.h, _ptr(pts), k, _ptr(n_key), _ptr(bits), _ptr(flags),
                                          _ptr(status), self.stream))
        return bits, flags

    def unpack_patches(self, bits):
        n = bits.numel() // 64
        dense = self.empty((n, 16, 16, 16, 1), torch.float32)
        _ffi.check(self.lib.caelo_unpack_patches(self.ctx, _ptr(bits), n, _ptr(dense), self.stream))
        return dense

    def pack_patches(self, dense):
        assert dense.dtype == torch.float32 and dense.is_contiguous()
        n = dense.numel() // 4096
        bits = self.empty((n, 64), torch.int64)
        _ffi.check(self.lib.caelo_pack_patches(self.ctx, _ptr(dense), n, _ptr(bits), self.stream))
        return bits

    def encode(self, bits, group=1):
        """bits [..., 64] int64 -> features [n/group, 20*group] f32."""
        assert bits.dtype == torch.int64 and bits.is_contiguous()
        n = bits.numel() // 64
        assert n % group == 0
        out = self.empty((n // group, 20 * group), torch.float32)
        ws = self._encode_ws(n)
        _ffi.check(self.lib.caelo_encode(self.ctx, _ptr(bits), n, group, _ptr(out), 20 * group, _ptr(ws), self.stream))
        return out

    def encode_layers(self, bits):
        """Test aid: encode (group 1, every patch) and return the activations the four encoder kernels leave in the
        workspace -- P2 [n,1024] (after pool2), F3 [n,2048] (after conv3), the Dense(200) pre-activations summed over the
        k slices [n,200] (bias not added) -- and the descriptors [n,20].  Workspace layout: caelo_encode_ws_layout."""
        n = bits.numel() // 64
        out = self.encode(bits, 1)
        ws = self._encode_ws(n)
        lay = (C.c_int64 * 6)()
        _ffi.check(self.lib.caelo_encode_ws_layout(n, C.cast(lay, C.c_void_p)))
        o_p2, o_f3, o_part, np_, used, sized = (int(v) for v in lay)
        p2 = ws[o_p2:o_p2 + np_ * 4096].view(torch.float32).view(np_, 1024)[:n]
        f3 = ws[o_f3:o_f3 + np_ * 8192].view(torch.float32).view(np_, 2048)[:n]
        part = ws[o_part:o_part + sized * np_ * 208 * 4].view(torch.float32).view(sized, np_, 208)[:used, :n, :200].sum(dim=0)
        return p2.clone(), f3.clone(), part, out

    # ---- BASELINE.json configs[4]: 32^3 patches (stress case, not a reference code path; csrc/config5.hip) -----
    @staticmethod
    def seeded_dense1_32(seed=5):
        """The stand-in dense_1 of the 32^3 encoder: N(0, 1/16384) [16384,200] f32 and a zero bias (SURVEY.md
        section 7 item 7 -- no trained weights exist at this size)."""
        import numpy as np
        rs = np.random.RandomState(seed)
        return (rs.standard_normal((16384, 200)) / 128.0).astype(np.float32), np.zeros(200, np.float32)

    def set_encoder32_dense(self, wd1, bd1):
        import numpy as np
        wd1 = np.ascontiguousarray(wd1, np.float32)
        bd1 = np.ascontiguousarray(bd1, np.float32)
        assert wd1.shape == (16384, 200) and bd1.shape == (200,)
        _ffi.check(self.lib.caelo_set_encoder32_dense(self.ctx, _hptr(wd1), _hptr(bd1)))

    def patches32(self, vmap, pts, n_key=None):
        """-> bits [K,3,512] int64: bit-packed 32^3 patches, window [-16,16)^3, no 496-NN cap."""
        assert pts.dtype == torch.float32 and pts.dim() == 2 and pts.stride(1) == 1 and pts.shape[1] >= 3
        k = pts.shape[0]
        bits = self.empty((k, 3, 512), torch.int64)
        _ffi.check(self.lib.caelo_patches32(self.ctx, vmap.h, _ptr(pts), int(pts.stride(0)), k, _ptr(n_key), _ptr(bits),
                                            self.stream))
        return bits

    def encode32(self, bits, group=1):
        """bits [..., 512] int64 -> features [n/group, 20*group] f32."""
        assert bits.dtype == torch.int64 and bits.is_contiguous()
        n = bits.numel() // 512
        assert n % group == 0
        out = self.empty((n // group, 20 * group), torch.float32)
        ws = self._ws("encode32", int(self.lib.caelo_encode32_ws_bytes(n)))
        _ffi.check(self.lib.caelo_encode32(self.ctx, _ptr(bits), n, group, _ptr(out), 20 * group, _ptr(ws), self.stream))
        return out

    def encode32_profile(self, bits, group=1):
        """encode32 + per-launch HIP-event timings (ms): conv1+pool, conv2+pool, conv3, Dense(200)+head.  Synchronises."""
        n = bits.numel() // 512
        out = self.empty((n // group, 20 * group), torch.float32)
        ws = self._ws("encode32", int(self.lib.caelo_encode32_ws_bytes(n)))
        ms = (C.c_float * 4)()
        _ffi.check(self.lib.caelo_encode32_profile(self.ctx, _ptr(bits), n, group, _ptr(out), 20 * group, _ptr(ws), self.stream,
                                                   C.cast(ms, C.c_void_p)))
        return out, list(ms)

    def extract32(self, pc, dist_channels=5):
        """Config-5 frame features: the key points of ``extract`` (project -> response -> top-K rule), described by
        32^3 patches instead of 16^3 ones.  Staged calls; the 16^3 descriptors ``extract`` also produced are
        overwritten in the rows."""
        ff = self.extract(pc, dist_channels, vmap=self.voxmap(max(self.max_points, pc.shape[0])))
        vmap = self.voxmap(max(self.max_points, pc.shape[0]))
        bits = self.patches32(vmap, ff.key_pts, ff.n_key)
        ff.rows[:, 0:60] = self.encode32(bits, group=3)
        return ff

    def encode_profile(self, bits, group=1):
        """encode + per-kernel HIP-event timings (ms): stage1, conv3, dense1, head, then the MFMA instructions stage 1
        executed (millions) and the FLOPs of one of them.  Synchronises."""
        n = bits.numel() // 64
        out = self.empty((n // group, 20 * group), torch.float32)
        ws = self._encode_ws(n)
        ms = (C.c_float * 6)()
        _ffi.check(self.lib.caelo_encode_profile(self.ctx, _ptr(bits), n, group, _ptr(out), 20 * group, _ptr(ws),
                                                 self.stream, C.cast(ms, C.c_void_p)))
        return out, list(ms)

    @staticmethod
    def _ld(t):
        assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
        return int(t.stride(0))

    def match(self, f0, f1, n0=None, n1=None):
        """f0 [k0,dim], f1 [k1,dim] (row-strided views allowed) -> pair_idx [k1] int64."""
        idx = self.zeros((f1.shape[0],), torch.int64)
        kmax = max(int(f0.shape[0]), int(f1.shape[0]))
        ws = self._ws("match%d" % kmax, int(self.lib.caelo_match_ws_bytes(kmax)))   # sized by the larger frame (caelo.h)
        _ffi.check(self.lib.caelo_match(self.ctx, _ptr(f0), self._ld(f0), f0.shape[0], _ptr(n0), _ptr(f1), self._ld(f1),
                                        f1.shape[0], _ptr(n1), f0.shape[1], _ptr(idx), _ptr(ws), self.stream))
        return idx

    def match_profile(self, frames, repeats=20):
        """The pipeline's launch shape of the NN match: len(frames) - 1 pairs of consecutive FrameFeatures behind one k_match_prep +
        one k_match_screen launch, timed with HIP events (caelo_match_profile).  -> (ms both kernels, ms prep alone, pair_idx [n,1024])."""
        n = len(frames) - 1
        assert 1 <= n <= 8 and all(f.rows.is_contiguous() for f in frames)
        rows = (C.c_void_p * (n + 1))(*[f.rows.data_ptr() for f in frames])
        nk = (C.c_void_p * (n + 1))(*[f.n_key.data_ptr() for f in frames])
        idx = self.zeros((n, MAX_K), torch.int64)
        ws = self._ws("match_profile", n * int(self.lib.caelo_match_ws_bytes(MAX_K)))
        ms = (C.c_float * 2)()
        _ffi.check(self.lib.caelo_match_profile(self.ctx, rows, n, nk, _ptr(idx), _ptr(ws), int(repeats), self.stream, C.cast(ms, C.c_void_p)))
        return float(ms[0]), float(ms[1]), idx

    def solve_rt(self, p0, p1):
        assert p0.shape == p1.shape and p0.dtype == torch.float32 and p0.is_contiguous() and p1.is_contiguous()
        R = self.empty((3, 3), torch.float32)
        T = self.empty((3, 1), torch.float32)
        cred = self.empty((1,), torch.int32)
        _ffi.check(self.lib.caelo_solve_rt(self.ctx, _ptr(p0), _ptr(p1), p0.shape[0], _ptr(R), _ptr(T), _ptr(cred), self.stream))
        return R, T, cred

    def ransac(self, pc0, pc1, pair_idx, rand, n1=None, cert=None):
        """rand: [1500,4] f64 uniform draws (device).  -> (result bytes tensor, mask [k1] uint8).  ``cert``: a
        [sizeof(caelo_ransac_cert)] u8 device tensor that receives the pair's certificate (``new_cert``)."""
        assert pair_idx.dtype == torch.int64 and pair_idx.is_contiguous()
        assert rand.dtype == torch.float64 and rand.numel() >= 6000 and rand.is_contiguous()
        res = self.empty((C.sizeof(_ffi.PoseResult),), torch.uint8)
        mask = self.empty((pc1.shape[0],), torch.uint8)
        _ffi.check(self.lib.caelo_ransac(self.ctx, _ptr(pc0), self._ld(pc0), _ptr(pc1), self._ld(pc1), _ptr(pair_idx),
                                         pc1.shape[0], _ptr(n1), _ptr(rand), _ptr(res), _ptr(mask),
                                         _ptr(self._ws("ransac", self.lib.caelo_ransac_ws_bytes())), _ptr(cert), self.stream))
        return res, mask

    def new_cert(self, k=1):
        """k zeroed caelo_ransac_cert records on the device."""
        return self.zeros((k, _ffi.CERT_DTYPE.itemsize), torch.uint8)

    # ---- the host half of the exact RANSAC (csrc/certify.hip, caelo/hostexact.py) ------------------------------------
    def bound_violations(self):
        """caelo_host_bound_violations: exact hypothesis counts the host half found above the device's upper bound since the process
        started (such a pair is decided by the reference's loop without bounds: still exact).  0 on every run so far."""
        return int(self.lib.caelo_host_bound_violations())

    def host_blas(self):
        """Bind libcaelo's host half to the BLAS / LAPACK of this process's NumPy (once; caelo/hostblas.py)."""
        from . import hostblas
        return hostblas.bind(self.lib)

    def certify(self, certs, rands=None, threads=None):
        """certs: [k, sizeof(caelo_ransac_cert)] u8 device tensor.  Copies the records to the host (synchronises) and runs
        caelo_host_certify: Match.py:181-214 replayed over the device's bounds, the deciding hypotheses re-evaluated through
        NumPy's own BLAS / LAPACK entry points.  ``rands``: the pairs' draws (list of host arrays or device tensors, or None) --
        only read for a pair that escalates beyond the 0.4 m level.  -> (results record array [k] (_ffi.POSE_DTYPE),
        masks [k,1024] u8, evals [k] i32, status [k] i32: 0 exact, 2 no bounds in the record (> 1024 pairs), 3 no record)."""
        k = int(certs.shape[0])
        nb = _ffi.CERT_DTYPE.itemsize
        assert certs.dtype == torch.uint8 and certs.dim() == 2 and certs.shape[1] == nb and certs.is_contiguous()
        host = self._pinned("cert", k * nb)[:k * nb].view(k, nb)
        host.copy_(certs, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        from . import hostexact
        return hostexact.certify_records(host.numpy(), rands, threads)

    def certify_batch(self, out, rands=None, threads=None):
        """Engine.certify over a FrameBatch that ran with ``certify=True``: the exact results replace ``out.result`` /
        ``out.inlier_mask`` on the device (frames without a pair keep what they had) and stay on the host in ``out.exact`` =
        (results, masks, evals, status).  Synchronises."""
        assert out.cert is not None, "run the pipeline with certify=True"
        res, masks, evals, status = self.certify(out.cert[:out.k], rands, threads)
        ok = status == 0
        if (status == 2).any():
            raise _ffi.CaeloError("a pair holds more than 1024 matches: no certificate (use api.RANSAC4RT for such inputs)")
        if ok.all():
            out.result[:out.k].copy_(torch.from_numpy(res.view(np.uint8).reshape(out.k, -1)), non_blocking=False)
            out.inlier_mask[:out.k].copy_(torch.from_numpy(masks), non_blocking=False)
        elif ok.any():
            sel = torch.from_numpy(np.flatnonzero(ok)).to(self.device)
            out.result[sel] = torch.from_numpy(np.ascontiguousarray(res.view(np.uint8).reshape(out.k, -1)[ok])).to(self.device)
            out.inlier_mask[sel] = torch.from_numpy(np.ascontiguousarray(masks[ok])).to(self.device)
        out.exact = (res, masks, evals, status)
        return out.exact

    def _pinned(self, kind, nbytes):
        t = self._pin.get(kind) if hasattr(self, "_pin") else None
        if t is None or t.numel() < nbytes:
            if not hasattr(self, "_pin"):
                self._pin = {}
            t = self._pin[kind] = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
        return t

    @staticmethod
    def pose_result(res):
        """Synchronising read of a caelo_pose_result."""
        return _ffi.PoseResult.from_buffer_copy(res.cpu().numpy().tobytes())

    # ---- fused hot path ------------------------------------------------------------------------------
    def extract(self, pc, dist_channels=5, vmap=None, rows=None, exact_voxels=False, dedup=True):
        """scan [N,4] f32 (device) -> FrameFeatures, ONE C-ABI call (caelo_extract), no host sync:
        project -> response CNN -> keypoints -> voxelize -> patch gather -> 3x encoder.
        dist_channels: 5 = demo calling mode (SphericalRing.py:414), 3 = batch mode
        (BatchPreprocess.py:97-98,131-136).  ``rows``: optional [1024,64] f32 output slot.
        The default one-pass voxelization and ``exact_voxels=True`` (the two-pass first-touch kernels of
        ``voxelize``, Voxel.py:139-141) produce the same voxel sets on every cloud, points on voxel faces
        included (metrically quantised scans have some in every frame).  ``dedup=False`` encodes
        every patch even when it is a bit-identical copy of another one of the frame (same result, more work)."""
        assert pc.dtype == torch.float32 and pc.dim() == 2 and pc.shape[1] == 4 and pc.is_contiguous()
        ws = self._ws("extract", int(self.lib.caelo_extract_ws_bytes()))
        vmap = vmap or self.voxmap(max(self.max_points, pc.shape[0]))
        rows = self.empty((MAX_K, 64), torch.float32) if rows is None else rows
        assert rows.shape == (MAX_K, 64) and rows.is_contiguous()
        kpix = self.empty((MAX_K, 2), torch.int64)
        nkey = self.empty((1,), torch.int32)
        flags = self.empty((MAX_K, 3), torch.uint8)
        status = self.empty((4,), torch.int32)
        base = rows.data_ptr()
        _ffi.check(self.lib.caelo_extract(self.ctx, vmap.h, _ptr(pc), pc.shape[0], dist_channels,
                                          (1 if exact_voxels else 0) | (0 if dedup else 2),
                                          C.c_void_p(base + 240), 64, C.c_void_p(base), 64, C.c_void_p(base + 252), 64,
                                          _ptr(kpix), _ptr(nkey), _ptr(flags), _ptr(status), _ptr(ws),
                                          self.stream))
        return FrameFeatures(rows, kpix, nkey, status, flags)

    def resolve_ties(self, ff, pc):
        """The fused path (extract / Pipeline.run) builds voxel SETS; where the 496-nearest cut of Voxel.py:195-196 splits a class
        of equidistant voxels it uses a canonical rule and sets flag bit 2, because scikit-learn's choice depends on the ORDER of
        the voxel lists.  This redoes such a frame's patches the reference's way -- the exact voxelization (caelo_voxelize: first
        touch recorded), the lists of the scales that need it ordered on the device (caelo_voxmap_order), caelo_patches with the
        kd-tree order (kdorder.hip), caelo_encode -- and writes the descriptors into ``ff.rows`` in place.  Synchronises; returns
        the number of tie-split patches it found (0: nothing done).  Rare: 0 of 614 400 patches on the KITTI-shaped scene, 115 on
        the clutter scene.  Many frames at once: ``resolve_ties_many``."""
        per = ((ff.flags & 2) != 0).sum(dim=0).cpu().tolist()          # tie-split patches per scale
        n_tie = int(sum(per))
        if n_tie == 0:
            return 0
        cap = max(self.max_points, pc.shape[0])
        vm, st = self.voxelize(pc, self.voxmap(cap, slot=2))
        self.voxmap_order(vm, sum(1 << s_ for s_ in range(3) if per[s_]))
        k = int(ff.n_key.item())
        bits, flags = self.patches(vm, ff.key_pts[:k].contiguous())
        raise_status(int(st.item()))
        ff.rows[:k, 0:60] = self.encode(bits.reshape(-1, 64), group=3)
        ff.flags[:k] = flags
        return n_tie

    def resolve_ties_many(self, items, lanes=8, batch=None):
        """resolve_ties over many frames at once: ``items`` = [(FrameFeatures, scan), ...] (``batch``: the FrameBatch whose frames
        0 .. len(items) - 1 they are, if so).  One read of the flags and key point counts for all of them, then each tied frame's
        redo is issued -- without any further host read -- to one of ``lanes`` side streams (own voxel maps and scratch per stream),
        so the kd-tree builds and queries of different frames, a few workgroups each, overlap instead of queueing (more lanes than
        hardware queues, 8, buy nothing).  The current stream waits for the lanes; one read of the status words at the end.
        -> (indices into ``items`` that were redone, their numbers of tie-split patches)."""
        t0_ = time.perf_counter()
        if not items:
            return [], []
        if batch is not None:          # the frames are frames 0 .. len(items) - 1 of a FrameBatch: two reductions instead of two per frame
            tie = ((batch.flags[:len(items)] & 2) != 0).sum(dim=1)
            nk = batch.n_key[:len(items)].reshape(-1)
        else:
            tie = torch.stack([((ff.flags & 2) != 0).sum(dim=0) for ff, _ in items])      # [frames, 3 scales]
            nk = torch.stack([ff.n_key.reshape(()) for ff, _ in items])
        both = torch.cat([tie.to(torch.int64), nk.to(torch.int64).reshape(-1, 1)], dim=1).cpu().numpy()
        tied = [i for i in range(len(items)) if both[i, :3].any()]
        if not tied:
            return [], []
        if not hasattr(self, "_tie_lanes") or len(self._tie_lanes) < lanes:
            self._tie_lanes = [torch.cuda.Stream(self.device) for _ in range(lanes)]
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        t1_ = time.perf_counter()
        statuses = []
        # Groups of up to eight tied frames per side stream (round 6): each frame's exact voxelization and list ordering one after the
        # other, then ONE caelo_patches_many -- the kd-tree builds and queries of the whole group behind one launch of each kd kernel
        # (a redo's chain of ~150 dependent quickselect passes costs a millisecond or two whatever the GPU has free: eight side by side
        # cost what one does) -- and ONE encoder launch set over the group's patches.  Frame by frame on eight lanes this took 1.2 ms per
        # tied frame (profiles/r05_ties_many.txt, r06_ties_many.txt).
        G = 8
        groups = [tied[g0:g0 + G] for g0 in range(0, len(tied), G)]
        n_lanes = min(lanes, len(groups))
        for gi, grp in enumerate(groups):
            lane = self._tie_lanes[gi % lanes]
            if gi < lanes:
                lane.wait_event(start)
            with torch.cuda.stream(lane):
                n = len(grp)
                gbits = self.empty((n, MAX_K, 3, 64), torch.int64)
                gflags = self.empty((n, MAX_K, 3), torch.uint8)
                gpts = self.empty((n, MAX_K, 3), torch.float32)
                maps, sts = [], []
                for q, i in enumerate(grp):
                    ff, pc = items[i]
                    cap = max(self.max_points, pc.shape[0])
                    vm, st = self.voxelize(pc, self.voxmap(cap, slot=10 + q))      # exact build: first touch recorded
                    self.voxmap_order(vm, sum(1 << s_ for s_ in range(3) if both[i, s_]))
                    gpts[q].copy_(ff.key_pts)
                    maps.append(vm)
                    sts.append(st)
                    statuses.append(st)
                arr = lambda xs: (C.c_void_p * n)(*xs)
                _ffi.check(self.lib.caelo_patches_many(self.ctx, n, arr([m.h for m in maps]), arr([gpts[q].data_ptr() for q in range(n)]), MAX_K,
                                                       arr([items[i][0].n_key.data_ptr() for i in grp]), arr([gbits[q].data_ptr() for q in range(n)]),
                                                       arr([gflags[q].data_ptr() for q in range(n)]), arr([st.data_ptr() for st in sts]), self.stream))
                feats = self.encode(gbits.reshape(-1, 64), group=3).reshape(n, MAX_K, 60)
                for q, i in enumerate(grp):
                    ff = items[i][0]
                    k = int(both[i, 3])
                    ff.rows[:k, 0:60] = feats[q, :k]
                    ff.flags[:k] = gflags[q, :k]
                for t in (gbits, gflags, gpts, feats):
                    t.record_stream(lane)
        for lane in self._tie_lanes[:n_lanes]:
            cur.wait_stream(lane)
        t4_ = time.perf_counter()
        raise_status(int(np.bitwise_or.reduce(torch.stack([s.reshape(()) for s in statuses]).cpu().numpy())))
        # a redone patch carries flag 4 instead of 2 (kdorder.hip); one that still carries 2 on a list long enough for the library's
        # kd-tree was LEFT on the canonical rule (the tree build gave up on a quickselect): not the reference's patch -- say so
        left = int(torch.stack([((items[i][0].flags[:int(both[i, 3])] & 2) != 0).sum() for i in tied]).sum().item())
        self.last_tie_unresolved = left
        if left:
            import warnings
            warnings.warn("%d tie-split patch(es) were left on the canonical rule by the kd-tree redo (flag 2 still set)" % left)
        t5_ = time.perf_counter()
        self.last_tie_times = dict(find_ms=1e3 * (t1_ - t0_), issue_ms=1e3 * (t4_ - t1_), wait_ms=1e3 * (t5_ - t4_))
        return tied, [int(both[i, :3].sum()) for i in tied]

    def match_pose_exact_many(self, pairs, rands, rands_host=None, threads=None, lanes=8):
        """match_pose_exact over many pairs [(fa, fb), ...]: every match and hypothesis launch is issued first, the certificates
        come back in one copy and the host half runs over them on ``threads`` threads.  -> (results [k] (_ffi.POSE_DTYPE),
        masks [k,1024] u8 (host), [pair_idx (device)] * k).  Synchronises once."""
        k = len(pairs)
        cur = torch.cuda.current_stream(self.device)
        cert = self.new_cert(k)
        idxs = []
        # pair j goes to side stream j % lanes (match and hypothesis kernels of one pair are ~0.1 ms of dependent launches that
        # leave most of the GPU idle; the workspaces are per stream)
        lanes = min(int(lanes), k)
        if lanes > 1:
            if not hasattr(self, "_tie_lanes") or len(self._tie_lanes) < lanes:
                self._tie_lanes = [torch.cuda.Stream(self.device) for _ in range(max(lanes, 8))]
            start = torch.cuda.Event()
            start.record(cur)
        for j, (fa, fb) in enumerate(pairs):
            s_ = self._tie_lanes[j % lanes] if lanes > 1 else cur
            if lanes > 1 and j < lanes:
                s_.wait_event(start)
            for f in (fa, fb):
                f.rows.record_stream(s_)
                f.n_key.record_stream(s_)
            with torch.cuda.stream(s_):
                idx = self.match(fa.features, fb.features, fa.n_key, fb.n_key)
                res_, mask_ = self.ransac(fa.key_pts, fb.key_pts, idx, rands[j], fb.n_key, cert=cert[j])
            for t in (idx, res_, mask_):
                t.record_stream(cur)
            idxs.append(idx)
        if lanes > 1:
            for s_ in self._tie_lanes[:lanes]:
                cur.wait_stream(s_)
        results, masks, _, status = self.certify(cert, list(rands_host) if rands_host is not None else list(rands), threads)
        if (status != 0).any():
            raise _ffi.CaeloError("pairs %s could not be certified (status %s)" % (np.flatnonzero(status != 0).tolist(), status[status != 0].tolist()))
        return results, masks, idxs

    def checked(self, ff, pc, dist_channels=5):
        """Synchronising status check of an extract() result: raises what the reference would raise."""
        raise_status(int(ff.status[0].item()))
        return ff

    def match_pose(self, fa, fb, rand):
        """Relative pose between two FrameFeatures (frame0 = fa, frame1 = fb), Match.py:241-283."""
        cur = torch.cuda.current_stream(self.device)
        for f in (fa, fb):          # rows produced on another lane: keep the allocator from recycling them early
            f.rows.record_stream(cur)
            f.n_key.record_stream(cur)
        idx = self.match(fa.features, fb.features, fa.n_key, fb.n_key)
        res, mask = self.ransac(fa.key_pts, fb.key_pts, idx, rand, fb.n_key)
        return res, mask, idx


    def match_pose_exact(self, fa, fb, rand, rand_host=None):
        """match_pose + the host half: -> (result record (_ffi.POSE_DTYPE), mask [1024] u8 (host), pair_idx (device)).  Synchronises."""
        cur = torch.cuda.current_stream(self.device)
        for f in (fa, fb):
            f.rows.record_stream(cur)
            f.n_key.record_stream(cur)
        idx = self.match(fa.features, fb.features, fa.n_key, fb.n_key)
        cert = self.new_cert(1)
        self.ransac(fa.key_pts, fb.key_pts, idx, rand, fb.n_key, cert=cert[0])
        results, masks, _, status = self.certify(cert, [rand_host if rand_host is not None else rand])
        if status[0] != 0:
            raise _ffi.CaeloError("pair could not be certified (status %d)" % status[0])
        return results[0], masks[0], idx


_default = None


def default_engine():
    global _default
    if _default is None:
        _default = Engine()
    return _default


_draws_tls = threading.local()


def ransac_draws(seed_or_rng, n=6000):
    """The uniform doubles RANSAC4RT would pull from NumPy's Mersenne Twister (Match.py:182).  An integer seed re-seeds a
    per-thread generator (the same stream as ``RandomState(seed)``): constructing a RandomState costs 116 us under the GIL -- four
    times the 6 000 draws, and what bounded run_sequence.py's loader threads."""
    if hasattr(seed_or_rng, "random_sample"):
        return seed_or_rng.random_sample(n)
    rs = getattr(_draws_tls, "rs", None)
    if rs is None:
        rs = _draws_tls.rs = np.random.RandomState(0)
    rs.seed(seed_or_rng)
    return rs.random_sample(n)
