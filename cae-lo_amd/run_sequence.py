#!/usr/bin/env python
"""run_sequence.py -- the counterpart of the reference's PoseEstimation.py loop on libcaelo (SURVEY.md 8c harness row,
8f-1): consecutive scans -> per-pair relative pose (R, T, nInliers, thr) -> chained KITTI poses [F,12] -> poses file.

    python run_sequence.py --synthetic 20 --out poses_/00.txt
    python run_sequence.py --scans <seq>/velodyne --calib <calib>/00/calib_.txt --out poses_/00.txt --save-artifacts
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_sequence.py --synthetic 800 ...

Frames are sharded contiguously over the ranks (one process per GPU); every rank runs its frames through the
native pipeline, ONE all-gather moves the boundary frame rows, pose rows are gathered to rank 0, which chains
them on the host (PoseEstimation.py:253-267) and writes `np.savetxt` rows like PoseEstimation.py:277.

RANSAC draws: the reference uses NumPy's unseeded global generator (Match.py:182); here pair (i, i+1) consumes
RandomState(seed_base + i).random_sample(...), so results do not depend on sharding or chunking.
"""
import argparse
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from caelo import _ffi, stageio, synth  # noqa: E402
import caelo  # noqa: E402
caelo.configure_runtime()  # this script owns its process: ask for 8 hardware queues before HIP starts
from caelo import dist as cdist  # noqa: E402
from caelo.engine import Engine, FrameBatch, FrameFeatures, raise_status, ransac_draws  # noqa: E402


def pose_rows(batch, k):
    """FrameBatch.result -> numpy: rel_rt [k,12], success, threshold, n_inliers (one host copy)."""
    return _parse_poses(batch.result[:k].cpu().numpy(), k)


def _parse_poses(raw, k):
    out = np.zeros((k, 12), np.float32)
    ok = np.zeros(k, bool); thr = np.zeros(k, np.float32); nin = np.zeros(k, np.int32)
    for i in range(k):
        r = _ffi.PoseResult.from_buffer_copy(raw[i].tobytes())
        out[i, :9], out[i, 9:] = np.array(r.R, np.float32), np.array(r.T, np.float32)
        ok[i], thr[i], nin[i] = bool(r.success), r.threshold, r.n_inliers
    return out, ok, thr, nin


def run_local(eng, load, lo, hi, seed_base, chunk, dist_channels, batch_frames, keep=None, strict_ties=True, tie_log=None, host_times=None,
              loader_threads=4, certify=True):
    """Frames [lo, hi) of this rank.  Returns per-pair rows for pairs (i-1, i), i in (lo, hi) -- the pair (lo-1, lo)
    is the caller's (it needs the previous rank's last frame) -- plus the first and last frame's features.

    Three things overlap, as in the reference's producer / consumer split (PoseEstimation.py:214-245, where a generator process
    prepares frame i + 1 while the main loop matches frame i):
      * a loader thread reads (or synthesises) the scans and RANSAC draws of chunk c + 1 into pinned host memory;
      * inside a chunk, a copy stream uploads batch b + 4 while the pipeline works on batch b (Pipeline.run_uploading, paced by this thread);
      * the poses and status words of chunk c come back through pinned buffers on a side stream and are parsed after chunk
        c + 1 has been issued.
    """
    import queue
    import threading
    t_setup = time.time()
    chunks = [(c0, min(hi, c0 + chunk)) for c0 in range(lo, hi, chunk)]
    q = queue.Queue(maxsize=2)

    pinned = {}   # a scan served twice (--pool) is pinned once

    def pin(a):
        t = pinned.get(id(a))
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).pin_memory()
            if getattr(load, "repeats", False):
                pinned[id(a)] = t
        return t

    ht = host_times if host_times is not None else {}   # seconds per host activity (what a "frames/s incl. loading" figure is made of)
    for k_ in ("load", "pin", "draws", "starved", "pipeline", "ties", "parse", "setup"):
        ht.setdefault(k_, 0.0)

    # The loader works with a few threads (file reads, NumPy's Mersenne Twister and the ray caster all release the GIL for most of
    # their time) and fills pinned buffers that are allocated once: a pinned allocation per chunk cost more than the copy it serves.
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, loader_threads)
    workers = ThreadPoolExecutor(max_workers=n_thr)
    draw_ring = [torch.empty((min(chunk, hi - lo), 6000), dtype=torch.float64, pin_memory=True) for _ in range(4)]   # queue of 2 + one in use + one being filled

    # Scans that come from files are read STRAIGHT into pinned slots that are allocated once and reused every fourth chunk (the queue
    # holds two chunks, one is in use, one is being filled): page-locking a fresh 2 MB buffer per scan cost 1.4 ms, three times the
    # read itself, and a copy on top.  (A source without `into` -- the synthetic pool -- is pinned per distinct scan as before.)
    ring_slots = {}
    cap = int(eng.max_points)

    ring_lock = threading.Lock()

    def pinned_slot(ci, j):
        # one pinned block per BATCH of a ring position, frame j at a fixed pitch inside it: the scans of a batch are contiguous and go
        # up behind ONE copy command (Pipeline.run_uploading detects the pitch; eight commands per batch cost the pipeline 20 %).
        # Allocated by whichever loader thread gets there first (page-locking 20 MB takes milliseconds: not under a common lock).
        key = (ci % 4, j // batch_frames)
        with ring_lock:
            lk = ring_slots.setdefault(("lock",) + key, threading.Lock())
        with lk:
            blk = ring_slots.get(key)
            if blk is None:
                blk = ring_slots[key] = torch.empty((batch_frames, cap, 4), dtype=torch.float32, pin_memory=True)
        return blk[j % batch_frames]

    def loader():
        try:
            for ci, (c0, c1) in enumerate(chunks):
                t_ = time.time()
                draws = draw_ring[ci % len(draw_ring)][:c1 - c0]
                dn = draws.numpy()

                def fill(j):
                    dn[j] = ransac_draws(seed_base + c0 + j - 1)
                drawn = [workers.submit(fill, j) for j in range(c1 - c0)]   # (beside the reads: both release the GIL for most of their time)
                if hasattr(load, "into"):
                    scans = list(workers.map(lambda j: load.into(c0 + j, pinned_slot(ci, j), pin), range(c1 - c0)))
                    t1_ = t2_ = time.time()
                else:
                    raw = list(workers.map(load, range(c0, c1)))
                    t1_ = time.time()
                    scans = [pin(a) for a in raw]
                    t2_ = time.time()
                for f_ in drawn:
                    f_.result()
                ht["load"] += t1_ - t_; ht["pin"] += t2_ - t1_; ht["draws"] += time.time() - t2_   # (draws: what was left of them after the scans)
                q.put((c0, c1, scans, draws))
        except BaseException as e:   # surfaced in the consumer
            q.put(e)

    threading.Thread(target=loader, daemon=True).start()
    # (the loader is already at work on the first chunks while the pipeline's buffers are allocated)
    pipe = eng.pipeline(batch_frames)
    import gc
    gc.collect()
    gc.freeze()       # what exists now is never scanned again: a full collection of this process (40-90 ms) no longer lands between two batches
    side = torch.cuda.Stream(device=eng.device)
    ht["setup"] = time.time() - t_setup
    rel, ok, thr, nin = [], [], [], []
    prev, first = None, None
    pending = None   # (k, has_prev, pinned result, pinned status, event)
    back_ring, n_back = [], 0

    def collect(p):
        k, has_prev, res_h, st_h, ev = p
        ev.synchronize()
        t_ = time.time()
        for st in st_h.numpy()[:, 0]:
            raise_status(int(st))
        r, o, t, n = _parse_poses(res_h.numpy(), k)
        s = 0 if has_prev else 1                                   # slot 0 of the first chunk has no predecessor here
        rel.append(r[s:]); ok.append(o[s:]); thr.append(t[s:]); nin.append(n[s:])
        ht["parse"] += time.time() - t_

    t_loop = time.time()
    for _ in chunks:
        t_ = time.time()
        item = q.get()
        ht["starved"] += time.time() - t_
        if isinstance(item, BaseException):
            raise item
        c0, c1, scans, draws = item
        t_ = time.time()
        draws_d = draws.to(eng.device, non_blocking=True)
        # certify: the exact RANSAC (the pipeline's certifier thread runs the host half on every pair while later batches are on the
        # GPU; the call returns when this chunk's inlier sets and poses -- the reference's bits -- are in batch.result / inlier_mask)
        dn_ = draws.numpy()
        batch = pipe.run_uploading(scans, [draws_d[i] for i in range(c1 - c0)], prev=prev, dist_channels=dist_channels,
                                   certify=certify, rands_host=[dn_[i] for i in range(c1 - c0)] if certify else None)
        ht["pipeline"] += time.time() - t_
        t_ = time.time()
        if strict_ties:
            # Frames whose 496-nearest cut (Voxel.py:195-196) splits a class of equidistant voxels: the fused path's canonical rule is
            # replaced by scikit-learn's kd-tree order (Engine.resolve_ties: ordered voxel lists, all on the device), then the pairs
            # such a frame is part of are matched again.  One synchronisation per chunk; rare (none on KITTI-shaped scans).
            if bool((batch.flags[:c1 - c0] & 2).any().item()):
                items = [(batch.frame(j), scans[j].to(eng.device)) for j in range(c1 - c0)]
                tied, n_t = eng.resolve_ties_many(items, batch=batch)         # the redos overlap on side streams
                if tie_log is not None:
                    tie_log.extend((c0 + j, n_) for j, n_ in zip(tied, n_t))
                redo = sorted({t for u in tied for t in (u, u + 1) if t < c1 - c0 and (t > 0 or prev is not None)})
                pairs_ = [(prev if j == 0 else batch.frame(j - 1), batch.frame(j)) for j in redo]
                if certify and redo:
                    rs_, ms_, xs_ = eng.match_pose_exact_many(pairs_, [draws_d[j] for j in redo], [dn_[j] for j in redo])
                    sel = torch.tensor(redo, device=eng.device)
                    batch.result[sel] = torch.from_numpy(rs_.view(np.uint8).reshape(len(redo), -1).copy()).to(eng.device)
                    batch.inlier_mask[sel] = torch.from_numpy(ms_).to(eng.device)
                    for j, x_ in zip(redo, xs_):
                        batch.pair_idx[j].copy_(x_)
                else:
                    for j, (fa_, fb_) in zip(redo, pairs_):
                        r_, m_, x_ = eng.match_pose(fa_, fb_, draws_d[j])
                        batch.result[j].copy_(r_); batch.inlier_mask[j].copy_(m_); batch.pair_idx[j].copy_(x_)
        ht["ties"] += time.time() - t_
        # read this chunk's small outputs back without stalling the stream that issues the next chunk
        done = torch.cuda.Event()
        done.record()
        if not back_ring:   # pinned read-back buffers, allocated once (three: one being parsed, one in flight, one being issued)
            for _i in range(3):
                back_ring.append((torch.empty((min(chunk, hi - lo),) + tuple(batch.result.shape[1:]), dtype=batch.result.dtype, pin_memory=True),
                                  torch.empty((min(chunk, hi - lo),) + tuple(batch.status.shape[1:]), dtype=batch.status.dtype, pin_memory=True)))
        res_h, st_h = (t[:c1 - c0] for t in back_ring[n_back % 3])
        n_back += 1
        with torch.cuda.stream(side):
            side.wait_event(done)
            res_h.copy_(batch.result[:c1 - c0], non_blocking=True)
            st_h.copy_(batch.status[:c1 - c0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        if pending is not None:
            collect(pending)
        pending = (c1 - c0, prev is not None, res_h, st_h, ev)
        if first is None:
            first = batch.frame(0)
        prev = batch.frame(c1 - c0 - 1)
        if keep is not None:
            keep(c0, batch)
        del batch, item, scans      # (back to the caching allocator before the next chunk asks for the same sizes)
    ht["loop"] = time.time() - t_loop
    if pending is not None:
        collect(pending)
    cat = (lambda xs, d: np.concatenate(xs) if xs else np.zeros((0,) + d))
    return cat(rel, (12,)), cat(ok, ()), cat(thr, ()), cat(nin, ()), first, prev


def _parse_poses_fast(raw, k):
    """_parse_poses without per-frame Python: the pinned read-back viewed as the record type of caelo_pose_result."""
    r = np.frombuffer(raw[:k].tobytes(), dtype=_ffi.POSE_DTYPE, count=k)
    out = np.concatenate([r["R"].reshape(k, 9), r["T"].reshape(k, 3)], axis=1).astype(np.float32)
    return out, r["success"] != 0, r["threshold"].astype(np.float32), r["n_inliers"].astype(np.int32)


def run_local_files(eng, files, lo, hi, seed_base, chunk, dist_channels, batch_frames, keep=None, strict_ties=True, tie_log=None, host_times=None,
                    loader_threads=16, certify=True, device_results=False):
    """run_local for scans that are FILES (round 6): the native loader (caelo_seqloader: pread into a pinned ring + the RANSAC draws,
    csrc/seqload.hip) works ahead on its own threads, a chunk of batches goes through Pipeline.run_loaded (one copy command per batch
    for scans and draws, jobs built column-wise), results come back through pinned buffers and are parsed as one record array.
    Same returns as run_local; same bits (the draws are NumPy's stream, the pipeline is the same)."""
    from caelo.engine import SeqLoader
    t_setup = time.time()
    ht = host_times if host_times is not None else {}
    for k_ in ("load", "pin", "draws", "starved", "pipeline", "ties", "parse", "setup"):
        ht.setdefault(k_, 0.0)
    B = batch_frames
    # a ring slot holds the LARGEST scan of this rank's files, not the engine's capacity: a batch goes up as one copy of the whole slot, and
    # what the copy engine moves beside the pipeline is what the upload mode costs (20.5 MB per batch at 160 000 points, 16.2 MB at 126 k)
    biggest = max(os.path.getsize(f_) for f_ in files[lo:hi]) // 16
    assert biggest <= eng.max_points, "a scan holds %d points, the engine was created for %d" % (biggest, eng.max_points)
    cap = min(int(eng.max_points), (int(biggest) + 1023) // 1024 * 1024)
    # (page-locking the ring takes tens of milliseconds: on a thread of its own, beside the allocation of the pipeline's buffers)
    import threading
    box = {}

    def make_loader():
        try:
            box["loader"] = SeqLoader(eng, files[lo:hi], first_frame=lo, batch=B, seed_base=seed_base, threads=loader_threads, ring=7, cap=cap)
        except BaseException as e:
            box["error"] = e
    th = threading.Thread(target=make_loader)
    th.start()
    pipe = eng.pipeline(B)
    th.join()
    if "error" in box:
        raise box["error"]
    loader = box["loader"]
    import gc
    gc.collect()
    gc.freeze()
    side = torch.cuda.Stream(device=eng.device)
    per_chunk = max(1, chunk // B)
    outs = [FrameBatch(eng, per_chunk * B) for _ in range(2)]
    back = [(torch.empty((per_chunk * B,) + tuple(outs[0].result.shape[1:]), dtype=outs[0].result.dtype, pin_memory=True),
             torch.empty((per_chunk * B,) + tuple(outs[0].status.shape[1:]), dtype=outs[0].status.dtype, pin_memory=True)) for _ in range(3)]
    ht["setup"] = time.time() - t_setup
    rel, ok, thr, nin = [], [], [], []
    prev, first, pending = None, None, None

    def collect(p):
        k, has_prev, res_h, st_h, ev = p
        ev.synchronize()
        t_ = time.time()
        st = st_h.numpy()[:k, 0]
        if st.any():
            for v in st[st != 0]:
                raise_status(int(v))
        r, o, t, n = _parse_poses_fast(res_h if isinstance(res_h, np.ndarray) else res_h.numpy(), k)
        s_ = 0 if has_prev else 1
        rel.append(r[s_:]); ok.append(o[s_:]); thr.append(t[s_:]); nin.append(n[s_:])
        ht["parse"] += time.time() - t_

    t_loop = time.time()
    for ci, b0 in enumerate(range(0, loader.n_batches, per_chunk)):
        nb = min(per_chunk, loader.n_batches - b0)
        t_ = time.time()
        # certified runs: the exact results are written by the host half into host arrays (batch.exact) -- they are read THERE, not
        # published to the device and copied back (publish=False: 51 ms of 0.40 s for 4 541 frames)
        # (device_results -- the artefact writer reads masks and pair indices from the device tensors -- publishes them as before)
        batch, k = pipe.run_loaded(loader, b0, nb, prev=prev, out=outs[ci % 2], dist_channels=dist_channels, certify=certify,
                                   publish=device_results or not certify)
        host_res = batch.exact[0][:k].copy().view(np.uint8).reshape(k, -1) if certify else None
        ht["pipeline"] += time.time() - t_
        ht["starved"] += pipe.last_loaded_times["starved_s"]
        for k_, v_ in pipe.last_loaded_times.items():
            ht["loaded_" + k_] = ht.get("loaded_" + k_, 0.0) + v_
        c0 = lo + b0 * B
        t_ = time.time()
        if strict_ties and bool((batch.flags[:k] & 2).any().item()):
            # (as run_local: the tied frames' scans are read again -- rare -- and redone in scikit-learn's kd-tree order, their pairs matched again)
            fl = (batch.flags[:k] & 2).reshape(k, -1).any(dim=1).cpu().numpy()
            items = [(batch.frame(j), torch.from_numpy(stageio.read_scan(files[c0 + j])).to(eng.device) if fl[j] else None) for j in range(k)]
            tied, n_t = eng.resolve_ties_many(items, batch=batch)
            if tie_log is not None:
                tie_log.extend((c0 + j, n_) for j, n_ in zip(tied, n_t))
            redo = sorted({t for u in tied for t in (u, u + 1) if t < k and (t > 0 or prev is not None)})
            pairs_ = [(prev if j == 0 else batch.frame(j - 1), batch.frame(j)) for j in redo]
            if redo:
                dn_ = [np.frombuffer(_draws_of(seed_base + c0 + j - 1)) for j in redo]
                dd_ = [torch.from_numpy(d_).to(eng.device) for d_ in dn_]
                if certify:
                    rs_, ms_, xs_ = eng.match_pose_exact_many(pairs_, dd_, dn_)
                    host_res[redo] = rs_.view(np.uint8).reshape(len(redo), -1)
                    if device_results:
                        sel = torch.tensor(redo, device=eng.device)
                        batch.inlier_mask[sel] = torch.from_numpy(ms_).to(eng.device)
                        for j, x_ in zip(redo, xs_):
                            batch.pair_idx[j].copy_(x_)
                else:
                    for j, (fa_, fb_), d_ in zip(redo, pairs_, dd_):
                        r_, m_, x_ = eng.match_pose(fa_, fb_, d_)
                        batch.result[j].copy_(r_); batch.inlier_mask[j].copy_(m_); batch.pair_idx[j].copy_(x_)
        ht["ties"] += time.time() - t_
        done = torch.cuda.Event()
        done.record()
        res_h, st_h = back[ci % 3]
        with torch.cuda.stream(side):
            side.wait_event(done)
            if not certify:
                res_h[:k].copy_(batch.result[:k], non_blocking=True)
            st_h[:k].copy_(batch.status[:k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        if pending is not None:
            collect(pending)
        pending = (k, prev is not None, host_res if certify else res_h, st_h, ev)
        if first is None:
            first = FrameFeatures.from_rows(batch.rows[0].clone())      # (the chunk buffers are reused two chunks later)
        last_rows = batch.rows[k - 1].clone()
        prev = FrameFeatures.from_rows(last_rows)
        if keep is not None:
            keep(c0, batch.view(0, k))
    ht["loop"] = time.time() - t_loop
    if pending is not None:
        collect(pending)
    ls = loader.stats()
    ht["load"], ht["draws"] = ls["read_s"], ls["draws_s"]      # (summed over the loader's threads)
    loader.close()
    cat = (lambda xs, d: np.concatenate(xs) if xs else np.zeros((0,) + d))
    return cat(rel, (12,)), cat(ok, ()), cat(thr, ()), cat(nin, ()), first, prev


def _draws_of(seed):
    return ransac_draws(seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic 64x2000 scans (caelo.synth)")
    ap.add_argument("--quantum", type=float, default=0.0, help="round the synthetic coordinates to multiples of this (m), e.g. 0.001")
    ap.add_argument("--scans", help="directory of KITTI velodyne .bin files")
    ap.add_argument("--calib", help="calib_.txt (PoseEstimation.py:199-203) or KITTI calib.txt; identity if omitted")
    ap.add_argument("--out", default="poses_/00.txt")
    ap.add_argument("--seed-base", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=960, help="frames resident on the GPU at a time (the rows of a chunk: 0.26 MB per frame)")
    ap.add_argument("--dist-channels", type=int, default=5, choices=(3, 5), help="5 = demo mode, 3 = batch mode (SURVEY 8a-3')")
    ap.add_argument("--batch", type=int, default=8, help="frames per launch (caelo_pipeline)")
    ap.add_argument("--scene", default="boxes", choices=("boxes", "clutter"), help="synthetic scene (caelo.synth)")
    ap.add_argument("--trajectory", default="circuit", choices=synth.TRAJECTORIES,
                    help="synthetic sensor path (caelo.synth.sensor_pose): 'circuit' = a periodic world with structure at every frame index; "
                         "'line' = the law of the goldens, which leaves the scene after ~150 frames")
    ap.add_argument("--pool", type=int, default=0, help="synthesise only this many distinct scans and walk them back and forth (0 1 .. P-1 "
                                                        "P-2 .. 0 1 ..: every pair stays a pair of neighbours); ray casting a scan costs ~0.5 s of CPU")
    ap.add_argument("--loader-threads", type=int, default=min(16, os.cpu_count() or 1), help="threads that read / synthesise scans and draw RANSAC's random numbers")
    ap.add_argument("--python-loader", action="store_true", help="--scans through round 5's Python loader threads instead of the native loader (caelo_seqloader)")
    ap.add_argument("--save-artifacts", action="store_true", help="write Features/*.mat and InliersIdx/*.mat next to the scans")
    ap.add_argument("--no-strict-ties", action="store_true", help="keep the fused path's canonical rule where the 496-nearest cut splits a "
                                                                  "tie class (default: such frames are redone in scikit-learn's kd-tree order)")
    ap.add_argument("--no-certify", action="store_true", help="the kernels' own RANSAC results (float64 fits) without the host half that makes "
                                                              "inlier sets and poses the reference's bits (csrc/certify.hip)")
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")),
                    help="ranks = GPUs; without a launcher the script starts them itself (caelo.dist.ensure_ranks)")
    args = ap.parse_args()

    world, rank, local_rank = cdist.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("CAELO_DIST_BACKEND", "nccl")   # gloo: several ranks on one GPU (functional tests)
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    eng = Engine(device=local_rank)
    if world > 1:
        dist.init_process_group(backend=backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
    if args.scans:
        files = sorted(glob.glob(os.path.join(args.scans, "*.bin")))
        n = len(files)

        def load(i):
            return stageio.read_scan(files[i])

        def load_into(i, slot, pin):   # the file's bytes into a pinned slot [capacity, 4]; a scan larger than the slot goes the old way
            nbytes = os.path.getsize(files[i])
            if nbytes % 16 or nbytes // 16 > slot.shape[0]:
                return pin(stageio.read_scan(files[i]))
            raw = slot.numpy().reshape(-1).view(np.uint8)
            with open(files[i], "rb") as f:
                got = f.readinto(memoryview(raw)[:nbytes])
            assert got == nbytes, "short read: %s" % files[i]
            return slot[:nbytes // 16]
        if not os.environ.get("CAELO_RUN_NO_PINNED_RING"):   # (the old path, for comparisons)
            load.into = load_into
    else:
        import threading
        cache, locks, guard = {}, {}, threading.Lock()

        def load(i):   # (called from the loader's threads: a pooled scan is synthesised once, by whoever asks first)
            if args.pool <= 1:
                return synth.make_scan(i, quantum=args.quantum or None, scene_kind=args.scene, trajectory=args.trajectory)
            i %= 2 * (args.pool - 1)
            i = i if i < args.pool else 2 * (args.pool - 1) - i
            with guard:
                lk = locks.setdefault(i, threading.Lock())
            with lk:
                if i not in cache:
                    cache[i] = synth.make_scan(i, quantum=args.quantum or None, scene_kind=args.scene, trajectory=args.trajectory)
            return cache[i]
        load.repeats = args.pool > 1

        def synth_into(i, slot, pin):   # a synthesised scan copied into its pinned ring slot (a batch's scans then go up in one copy)
            a = load(i)
            if a.shape[0] > slot.shape[0]:
                return pin(a)
            slot.numpy()[:a.shape[0]] = a
            return slot[:a.shape[0]]
        # opt-in: synthesising the scans is what bounds a synthetic run (1.5 of 2 s for 4 541 frames), and copying each into the ring
        # costs the loader more (+0.25 s) than the single copy command per batch saves the pipeline calls (0.88 -> 0.58 s)
        if os.environ.get("CAELO_RUN_SYNTH_RING"):
            load.into = synth_into
        n = args.synthetic
        files = [os.path.join(os.path.dirname(os.path.abspath(args.out)), "synthetic", "velodyne", "%06d.bin" % i) for i in range(n)]
    assert n >= 2, "need at least two scans (--synthetic N or --scans DIR)"
    Tr = stageio.read_calib_tr(args.calib) if args.calib else None

    lo, hi = cdist.shard_frames(n, rank, world)
    t0 = time.time()

    def keep(c0, batch):
        if not args.save_artifacts:
            return
        rows = batch.rows.cpu().numpy(); nk = batch.n_key.cpu().numpy()
        idx = batch.pair_idx.cpu().numpy(); mask = batch.inlier_mask.cpu().numpy().astype(bool)
        for j in range(len(nk)):
            k = int(nk[j])
            stageio.save_features(files[c0 + j], rows[j, :k, 60:63], rows[j, :k, 0:60])
            if c0 + j > lo:
                m = mask[j, :k]
                stageio.save_inliers(os.path.dirname(os.path.dirname(files[c0 + j])), c0 + j - 1, c0 + j, idx[j, :k][m], np.arange(k)[m])

    tie_log = []
    host_times = {}
    if args.scans and not args.python_loader:
        rel, ok, thr, nin, first, last = run_local_files(eng, files, lo, hi, args.seed_base, args.chunk, args.dist_channels,
                                                         args.batch, keep, strict_ties=not args.no_strict_ties, tie_log=tie_log, host_times=host_times,
                                                         loader_threads=args.loader_threads, certify=not args.no_certify, device_results=args.save_artifacts)
    else:
        rel, ok, thr, nin, first, last = run_local(eng, load, lo, hi, args.seed_base, args.chunk, args.dist_channels,
                                                   args.batch, keep, strict_ties=not args.no_strict_ties, tie_log=tie_log, host_times=host_times,
                                                   loader_threads=args.loader_threads, certify=not args.no_certify)
    if tie_log:
        print("rank %d: %d frame(s) redone in scikit-learn's tie order (%d patches): %s" % (
            rank, len(tie_log), sum(n for _, n in tie_log), [f for f, _ in tie_log][:20]), file=sys.stderr)
    if world > 1:   # the pair that straddles the rank boundary: ONE all-gather of the boundary rows
        gathered = cdist.all_gather_boundary(last.rows)
        if rank > 0:
            prev = FrameFeatures.from_rows(gathered[rank - 1])
            bd = ransac_draws(args.seed_base + lo - 1)
            if args.no_certify:
                r = eng.pose_result(eng.match_pose(prev, first, torch.from_numpy(bd).to(eng.device))[0])
            else:   # exact like every other pair
                r = _ffi.PoseResult.from_buffer_copy(eng.match_pose_exact(prev, first, torch.from_numpy(bd).to(eng.device), bd)[0].tobytes())
            row = np.r_[np.array(r.R, np.float32), np.array(r.T, np.float32)][None]
            rel = np.concatenate([row, rel]); ok = np.r_[bool(r.success), ok]; thr = np.r_[np.float32(r.threshold), thr]
            nin = np.r_[np.int32(r.n_inliers), nin]
        extra = torch.from_numpy(np.c_[rel, ok, thr, nin].astype(np.float32)).to(eng.device)
        allrows = cdist.gather_poses(extra, n).cpu().numpy()
        rel, ok, thr, nin = allrows[:, :12], allrows[:, 12] > 0, allrows[:, 13], allrows[:, 14].astype(np.int32)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if rank == 0:
        poses = stageio.chain_poses(rel, Tr)
        stageio.write_poses(args.out, poses)
        for i in range(len(rel) if len(rel) <= 200 else 0):
            print("%06d-%06d ok=%d thr=%.1f inliers=%4d T=[% .3f % .3f % .3f]" % (i, i + 1, ok[i], thr[i], nin[i], rel[i, 9], rel[i, 10], rel[i, 11]))
        print("%d frames, %d pairs on %d GPU(s) in %.2f s (%.1f frames/s incl. scan loading / synthesis, upload and read-back; %d of %d poses solved) -> %s" % (
            n, len(rel), world, dt, n / dt, int(np.sum(ok)), len(rel), args.out))
        h = host_times
        print("rank 0 host seconds -- loader thread: reading / synthesising scans %.2f, pinning %.2f, RANSAC draws %.2f; issuing thread: "
              "pipeline creation + heap freeze %.2f, waiting for the loader (%d threads) %.2f, pipeline calls (uploads paced, %d frames) %.2f, tie check + read-back issue %.2f, "
              "parsing results %.2f (the chunk loop as a whole %.2f)" % (h["load"], h["pin"], h["draws"], h["setup"], args.loader_threads, h["starved"], hi - lo, h["pipeline"], h["ties"], h["parse"], h.get("loop", 0.0)))
        if any(k_.startswith("loaded_") for k_ in h):
            print("        inside the pipeline calls (Pipeline.run_loaded): " + ", ".join("%s %.3f" % (k_[7:], v_) for k_, v_ in sorted(h.items()) if k_.startswith("loaded_")))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
