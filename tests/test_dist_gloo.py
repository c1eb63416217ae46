"""CPU, world_size 2 (gloo): the multi-GPU path -- shard frames, ONE all-gather of per-frame rows,
match the pairs whose second frame a rank owns, gather the poses."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _worker(rank, world, n_frames, port, out_dir):
    sys.path.insert(0, os.path.join(REPO, "cae-lo_amd"))
    from caelo import dist as cd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K = 16
    lo, hi = cd.shard_frames(n_frames, rank, world)

    def frame_rows(f):  # deterministic stand-in for Engine.extract(frame f)
        g = torch.Generator().manual_seed(1000 + f)
        kp = torch.rand((K, 3), generator=g)
        ft = torch.rand((K, 60), generator=g)
        return cd.pack_rows(kp, ft, K - (f % 3))

    local = torch.stack([frame_rows(f) for f in range(lo, hi)])
    allrows = cd.all_gather_frames(local, n_frames)
    assert allrows.shape == (n_frames, K, cd.ROW)
    for f in range(n_frames):
        assert torch.equal(allrows[f], frame_rows(f)), "rank %d frame %d" % (rank, f)
    # "match" = a cheap order-sensitive function of the two frames' rows
    rts = []
    for (a, b) in cd.local_pairs(n_frames, rank, world):
        kpa, fa, na = cd.unpack_rows(allrows[a])
        kpb, fb, nb = cd.unpack_rows(allrows[b])
        assert int(na) == K - (a % 3) and int(nb) == K - (b % 3)
        v = (fa[: int(na)].sum() - 2 * fb[: int(nb)].sum() + kpa.sum()).item()
        rts.append([v] + [float(a), float(b)] + [0.0] * 9)
    rts = torch.tensor(rts, dtype=torch.float32).reshape(-1, 12)
    # the boundary variant moves only each rank's last frame
    last = cd.all_gather_boundary(local[-1])
    assert last.shape == (world, K, cd.ROW)
    for r in range(world):
        assert torch.equal(last[r], frame_rows(cd.shard_frames(n_frames, r, world)[1] - 1))
    # the gather in pieces (what bench.py overlaps with the extraction): equal blocks, chunks of 3 frames, the same rows
    if n_frames % world == 0:
        n_local = n_frames // world
        g = cd.ChunkedFrameGather(local, n_local, 3)
        assert g.side is None and len(g.bounds) == -(-n_local // 3)
        for clo, chi in g.bounds:
            g.chunk(clo, chi)
        frame_of = g.finish()
        for r in range(world):
            for i in range(n_local):
                assert torch.equal(frame_of(r, i), frame_rows(r * n_local + i))
        assert g.nbytes() == world * n_local * K * cd.ROW * 4
    allrt = cd.gather_poses(rts, n_frames)
    assert allrt.shape == (n_frames - 1, 12)
    assert allrt[:, 1].tolist() == [float(f) for f in range(n_frames - 1)]
    np.save(os.path.join(out_dir, "rt_%d.npy" % rank), allrt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_sharding(tmp_path):
    world = 2
    for n_frames, port in ((7, 29611), (8, 29612)):
        mp.spawn(_worker, args=(world, n_frames, port, str(tmp_path)), nprocs=world, join=True)
        a = np.load(tmp_path / "rt_0.npy")
        b = np.load(tmp_path / "rt_1.npy")
        assert np.array_equal(a, b) and a.shape == (n_frames - 1, 12)
