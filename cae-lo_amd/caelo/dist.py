"""Frame sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" == RCCL on
ROCm, "gloo" in the CPU tests).

The path shards by frames: feature extraction is independent per frame
(the reference fans frames out to worker processes: BatchPreprocess.py:215-228,
PoseEstimation.py:79-99) and matching needs consecutive frames only (PoseEstimation.py:241-251).
Each rank extracts a contiguous block of frames, ONE all-gather moves per-frame rows
(60-d descriptor | key point xyz | valid flag) over xGMI, then every rank matches the pairs whose
second frame it owns -- the pair straddling a block boundary takes its first frame from the
gathered rows of the previous rank.  ``all_gather_frames`` moves every frame (what a global
loop-closure search would need); ``all_gather_boundary`` moves only each rank's last frame, which is
all that consecutive-pair odometry (PoseEstimation.py:241-251) reads from another rank.  Pose chaining (PoseEstimation.py:253-267) is a prefix product
of the gathered per-pair (R, T) on rank 0 (host, 3x4 algebra).

Nothing here is device specific: the same code runs under gloo on CPU tensors in tests/.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist

ROW = 64  # floats per key point row: descriptor (60) + xyz (3) + valid (1)


def _all_gather_into(recv, send, group=None, async_op=False):
    """dist.all_gather_into_tensor; device tensors under the gloo backend (functional tests that put several ranks
    on one GPU) are staged through the host (always synchronously).  ``async_op``: returns the Work handle instead of making
    the current stream wait for the collective (c10d's synchronous form ends with a device-side wait of the current stream for
    RCCL's stream)."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
        return None
    return dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)


def ensure_ranks(n_ranks, script, argv):
    """``script --gpus N`` started WITHOUT a launcher (no WORLD_SIZE in the environment) starts its N ranks itself: the process
    re-executes the script under ``torch.distributed.run`` -- one rank per GPU, rendezvous on 127.0.0.1 and a free port -- and
    exits with the launcher's status.  The counterpart of the reference's own fan-out (PoseEstimation.py:79-99,
    BatchPreprocess.py:215-228 start one worker process per slice of the frame list).  Returns (world, rank, local_rank) when
    this process IS a rank (or N == 1).  Fails loudly -- exit status 2, nothing printed on stdout -- when

    * the launcher's WORLD_SIZE differs from ``--gpus``, or
    * fewer than N GPUs are visible and the functional-test backend (``CAELO_DIST_BACKEND=gloo``: several ranks share a GPU,
      rows staged through the host) was not asked for: a run that silently used fewer ranks would print a line that looks
      like an N-GPU result."""
    backend = os.environ.get("CAELO_DIST_BACKEND", "nccl")
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != n_ranks:
            sys.stderr.write("%s: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it with --nproc-per-node == "
                             "--gpus, or without a launcher\n" % (os.path.basename(script), n_ranks, world))
            sys.exit(2)
        return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if n_ranks <= 1:
        return 1, 0, 0
    visible = torch.cuda.device_count()
    if backend == "nccl" and visible < n_ranks:
        sys.stderr.write("%s: --gpus %d needs %d visible GPUs (RCCL: one rank per device), %d visible; "
                         "CAELO_DIST_BACKEND=gloo runs the ranks on the visible GPU(s) as a functional test\n"
                         % (os.path.basename(script), n_ranks, n_ranks, visible))
        sys.exit(2)
    if visible < 1:
        sys.stderr.write("%s: no GPU visible\n" % os.path.basename(script))
        sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    sys.exit(subprocess.call(cmd, env=env))


def shard_frames(n_frames, rank, world):
    """Contiguous block [lo, hi) of frame indices owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(frame, n_frames, world):
    base, rem = divmod(n_frames, world)
    cut = rem * (base + 1)
    return frame // (base + 1) if frame < cut else rem + (frame - cut) // max(base, 1)


def pack_rows(key_pts, features, n_key):
    """[K,3] f32, [K,60] f32, n_key (int or 0-d/1-elem tensor) -> [K,64] rows with a valid column."""
    k = key_pts.shape[0]
    nk = n_key if torch.is_tensor(n_key) else torch.tensor([n_key], device=key_pts.device)
    valid = (torch.arange(k, device=key_pts.device) < nk.reshape(-1)[0]).to(key_pts.dtype).unsqueeze(1)
    return torch.cat([features, key_pts, valid], dim=1)


def unpack_rows(rows):
    """[K,64] -> (key_pts [K,3], features [K,60], n_key int32[1]) -- contiguous copies."""
    n_key = rows[:, 63].sum().round().to(torch.int32).reshape(1)
    return rows[:, 60:63].contiguous(), rows[:, 0:60].contiguous(), n_key


def all_gather_frames(local_rows, n_frames, group=None):
    """local_rows [F_local, K, 64] -> [n_frames, K, 64] on every rank (ONE collective).
    Blocks are padded to the largest block so a single all_gather_into_tensor suffices."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_rows
    rank = dist.get_rank(group)
    fmax = -(-n_frames // world)
    lo, hi = shard_frames(n_frames, rank, world)
    assert local_rows.shape[0] == hi - lo
    k = local_rows.shape[1]
    send = local_rows
    if hi - lo < fmax:
        send = torch.cat([local_rows, local_rows.new_zeros((fmax - (hi - lo), k, ROW))], dim=0)
    recv = local_rows.new_empty((world * fmax, k, ROW))
    _all_gather_into(recv, send.contiguous(), group)
    parts = []
    for r in range(world):
        rlo, rhi = shard_frames(n_frames, r, world)
        parts.append(recv[r * fmax: r * fmax + (rhi - rlo)])
    return torch.cat(parts, dim=0)


class ChunkedFrameGather:
    """``all_gather_frames`` in pieces that overlap the extraction: the rows of frames [lo, hi) of EVERY rank are gathered on a side
    stream as soon as the caller says they are complete (``chunk`` is called once they are WRITTEN -- ``Pipeline.run(on_encoded=
    g.chunk)``, paced by the host -- or once the side stream has been made to wait for them, ``Pipeline.wait_encoded(g.side)``, the
    device-side form that costs the pipeline a quarter of its rate), while the pipeline works on the next batch.  Every rank must call ``chunk``
    with the same (lo, hi) sequence (equal blocks: n_local frames per rank).  Buffers are allocated up front."""

    def __init__(self, rows, n_local, chunk_frames, group=None, even_alone=False, timed=False):
        """``timed``: every collective is bracketed by two events on the side stream and issued synchronously there
        (``collective_ms``); default: issued asynchronously -- the collective runs on RCCL's own stream and NO stream of ours waits
        for it until ``finish`` (a wait that sits unsatisfied in a queue costs the frame pipeline beside it throughput:
        DESIGN.md 5)."""
        self.rows, self.n, self.group = rows, n_local, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.alone = self.world == 1 and not (even_alone and dist.is_initialized())   # (a world of one still runs the collectives in tests)
        self.bounds = [(lo, min(n_local, lo + chunk_frames)) for lo in range(0, n_local, chunk_frames)]
        self.recv = [rows.new_empty((self.world * (hi - lo),) + tuple(rows.shape[1:])) for lo, hi in self.bounds]
        self.side = torch.cuda.Stream(device=rows.device) if rows.is_cuda else None
        self.done = 0
        self.timed = bool(timed)
        self.events = []
        self.works = []

    def chunk(self, lo, hi):
        assert self.bounds[self.done] == (lo, hi), "chunks arrive in order, the same on every rank"
        recv = self.recv[self.done]
        self.done += 1
        if self.alone:
            return
        if self.side is None:
            _all_gather_into(recv, self.rows[lo:hi].contiguous(), self.group)
            return
        with torch.cuda.stream(self.side):
            if self.timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _all_gather_into(recv, self.rows[lo:hi], self.group)   # (a contiguous slice of the frame rows)
                e1.record()
                self.events.append((e0, e1))
            else:
                w = _all_gather_into(recv, self.rows[lo:hi], self.group, async_op=True)
                if w is not None:
                    self.works.append(w)

    def finish(self):
        """the current stream waits for the collectives (and the side stream); returns ``frame(rank, i)`` -> rows [K, 64] of frame
        i of ``rank``."""
        assert self.done == len(self.bounds)
        for w in self.works:
            w.wait()                       # (device-side: the current stream waits for RCCL's)
        self.works = []
        if self.side is not None:
            torch.cuda.current_stream(self.rows.device).wait_stream(self.side)

        def frame(rank, i):
            if self.alone:
                return self.rows[i]
            c = next(j for j, (lo, hi) in enumerate(self.bounds) if lo <= i < hi)
            lo, hi = self.bounds[c]
            return self.recv[c][rank * (hi - lo) + (i - lo)]
        return frame

    def nbytes(self):
        return sum(r.numel() * r.element_size() for r in self.recv) if not self.alone else 0

    def collective_ms(self):
        """sum of the collectives' own durations on the side stream (after a synchronize); None when they were not timed"""
        return sum(e0.elapsed_time(e1) for e0, e1 in self.events) if self.timed or self.alone or self.side is None else None


def all_gather_boundary(last_rows, group=None):
    """last_rows [K, 64] (this rank's LAST frame) -> [world, K, 64]: the only remote rows consecutive-pair
    matching needs (rank r matches its first frame against row r-1).  Same single collective as
    all_gather_frames with 1/F_local of the payload (262 KB per rank instead of F_local x 262 KB)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return last_rows.unsqueeze(0)
    recv = last_rows.new_empty((world * last_rows.shape[0],) + tuple(last_rows.shape[1:]))  # concatenated form
    _all_gather_into(recv, last_rows.contiguous(), group)
    return recv.view((world,) + tuple(last_rows.shape))


def local_pairs(n_frames, rank, world):
    """Pairs (f-1, f) matched by ``rank``: those whose second frame it owns (frame 0 has none)."""
    lo, hi = shard_frames(n_frames, rank, world)
    return [(f - 1, f) for f in range(max(lo, 1), hi)]


def gather_poses(local_rt, n_frames, group=None):
    """local_rt [P_local, w] f32 rows of this rank's pairs (w = 12: R row-major | T; more columns ride along) ->
    [n_frames-1, w] on every rank (second collective, 48 B per pair)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_rt
    pmax = -(-n_frames // world)
    w = local_rt.shape[1]
    send = local_rt.new_zeros((pmax, w))
    send[: local_rt.shape[0]] = local_rt
    recv = local_rt.new_empty((world * pmax, w))
    _all_gather_into(recv, send, group)
    parts = []
    for r in range(world):
        parts.append(recv[r * pmax: r * pmax + len(local_pairs(n_frames, r, world))])
    return torch.cat(parts, dim=0)


def chain_poses(rel_rt, Tr=None):
    """PoseEstimation.py:230-267 (see caelo.stageio.chain_poses): host-side prefix product over the gathered
    per-pair (R, T) rows, float32 like the reference."""
    from .stageio import chain_poses as _chain
    return _chain(rel_rt, Tr)
