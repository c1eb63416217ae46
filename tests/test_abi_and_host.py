"""CPU: the C-ABI library loads and exports every declared symbol; host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, WEIGHTS


def test_library_exports_every_declared_symbol():
    from caelo import _ffi
    lib = _ffi.load()
    hdr = open(os.path.join(REPO, "include", "caelo.h")).read()
    declared = set(re.findall(r"\b(caelo_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    typed = {name for name, _, _ in _ffi.SIGNATURES}
    assert declared == typed, "header vs ctypes table: %s" % sorted(declared ^ typed)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.caelo_abi_version() == _ffi.ABI_VERSION == 5
    assert ctypes.sizeof(_ffi.PoseResult) == (9 + 3 + 9 + 3 + 1) * 4 + 5 * 4


def test_no_cpu_fallback():
    import torch
    from caelo import _ffi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    rc = _ffi.load().caelo_create(ctypes.byref(ctx), 0)
    assert rc != 0 and _ffi.load().caelo_last_error()
    from caelo.engine import Engine
    with pytest.raises(_ffi.CaeloError):
        Engine()


def test_host_only_entry_points_without_a_gpu():
    """Entry points that touch no device: the encoder workspace layout agrees with the size the library asks for (a test that
    looks into the workspace asks for offsets instead of knowing a header size), and caelo_upload_many rejects null tables
    before any HIP call (n = 0 is a no-op)."""
    from caelo import _ffi
    lib = _ffi.load()
    for n in (1, 64, 3072, 24576):
        lay = (ctypes.c_int64 * 6)()
        assert lib.caelo_encode_ws_layout(n, ctypes.cast(lay, ctypes.c_void_p)) == 0
        o_p2, o_f3, o_part, np_, used, sized = (int(v) for v in lay)
        assert np_ >= n and np_ % 64 == 0 and 0 < used <= sized
        assert 0 < o_p2 < o_f3 < o_part and o_f3 - o_p2 == np_ * 1024 * 4 and o_part - o_f3 == np_ * 2048 * 4
        assert o_part + sized * np_ * 208 * 4 == lib.caelo_encode_ws_bytes(n)
    assert lib.caelo_encode_ws_layout(0, None) != 0
    assert lib.caelo_upload_many(None, None, None, 0, None) == 0
    assert lib.caelo_upload_many(None, None, None, 1, None) != 0 and lib.caelo_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "cae-lo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "import oracle" not in txt and "liboracle" not in txt and "/root/reference" not in txt, f


def test_h5lite_reads_keras_models():
    from caelo.engine import read_keras_weights
    kind, ws = read_keras_weights(os.path.join(WEIGHTS, "SphericalRingPCRespondLayer.h5"))
    assert kind == "respond" and [w.shape for w in ws] == [(3, 3, 3, 32), (32,), (1, 1, 32, 8), (8,)]
    kind, ws = read_keras_weights(os.path.join(WEIGHTS, "EncoderModel4VoxelPatch.h5"))
    assert kind == "encoder" and ws[6].shape == (2048, 200) and ws[0].dtype == np.float32
    # spot values pinned from h5py in the build container
    assert abs(float(np.abs(ws[6]).max()) - 0.18720529973506927) < 1e-9
    assert abs(float(np.abs(ws[8]).max()) - 0.7936325073242188) < 1e-9
    import hashlib
    sums = {"SphericalRingPCRespondLayer.h5": "34c9762bbe51f0c9a689943c6a71f4a258ddedb3da3ecb71818b9f1d6b349d45",
            "EncoderModel4VoxelPatch.h5": "89d923f7ae625fda1e60e67d5fcbe6785539166f4401ac9138a24e64aa5ec527"}
    for f, s in sums.items():
        assert hashlib.sha256(open(os.path.join(WEIGHTS, f), "rb").read()).hexdigest() == s


def test_frame_sharding_helpers():
    from caelo import dist as cd
    for n, w in ((10, 4), (8, 8), (4541, 8), (7, 3), (3, 4)):
        blocks = [cd.shard_frames(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
        for f in range(n):
            r = cd.owner_of(f, n, w)
            assert blocks[r][0] <= f < blocks[r][1]
        pairs = sum((cd.local_pairs(n, r, w) for r in range(w)), [])
        assert pairs == [(f - 1, f) for f in range(1, n)]


def test_pose_chaining_matches_float32_reference_formula():
    from caelo import dist as cd
    from caelo import synth
    rel = []
    for f in range(5):
        R, T = synth.relative_pose_gt(f, f + 1)
        rel.append(np.r_[R.reshape(9), T.reshape(3)])
    poses = cd.chain_poses(np.array(rel, np.float32))
    assert poses.shape == (6, 12) and poses.dtype == np.float32
    # with Tr = I the chained pose of frame f is the sensor pose of frame f
    (t, yaw) = synth.sensor_pose(5)
    assert np.allclose(poses[5].reshape(3, 4)[:, 3], t, atol=1e-4)
    assert abs(np.arctan2(poses[5][4], poses[5][0]) - yaw) < 1e-5
    # a non-trivial calibration conjugates the motion (PoseEstimation.py:259-262)
    Tr = np.array([[0, -1, 0, 0.1], [0, 0, -1, -0.2], [1, 0, 0, 0.3]], np.float32)
    p2 = cd.chain_poses(np.array(rel, np.float32), Tr)
    assert np.allclose(p2[1].reshape(3, 4)[:, :3] @ p2[1].reshape(3, 4)[:, :3].T, np.eye(3), atol=1e-5)


def test_multi_rank_request_fails_loudly_or_self_launches():
    """`bench.py --gpus N` / `run_sequence.py --gpus N` never print a one-rank line for an N-rank request (missing 1):
    without a launcher they start the ranks themselves (covered on the GPU box); with a launcher whose WORLD_SIZE differs, or
    with fewer than N visible GPUs and no gloo override, they exit 2 with nothing on stdout.  (PoseEstimation.py:79-99 is the
    reference's fan-out: it always starts the worker processes it was asked for.)"""
    import subprocess
    import sys
    import torch
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CAELO_DIST_BACKEND")}
    for script, extra in ((os.path.join(REPO, "bench.py"), ["--steps", "1", "--warmup", "0"]),
                          (os.path.join(REPO, "cae-lo_amd", "run_sequence.py"), ["--synthetic", "4"])):
        r = subprocess.run([sys.executable, script, "--gpus", "3"] + extra, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                           capture_output=True, timeout=300)
        assert r.returncode == 2 and r.stdout == b"" and b"--gpus 3" in r.stderr
        if torch.cuda.device_count() < 64:
            r = subprocess.run([sys.executable, script, "--gpus", "64"] + extra, env=env, capture_output=True, timeout=300)
            assert r.returncode == 2 and r.stdout == b"" and b"visible" in r.stderr


def test_ransac_draws_are_numpys_stream():
    """engine.ransac_draws re-seeds a per-thread RandomState: the same doubles as a fresh RandomState(seed) (Match.py:182), from any thread"""
    import threading
    from caelo.engine import ransac_draws
    want = {s: np.random.RandomState(s).random_sample(6000) for s in (0, 1, 999, 2 ** 31 - 1, 123456789)}
    got = {}

    def work(seeds):
        for s in seeds:
            got[s] = ransac_draws(s)
    ts = [threading.Thread(target=work, args=(list(want)[i::2],)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(np.array_equal(got[s], want[s]) for s in want)
    rng = np.random.RandomState(5)
    assert np.array_equal(ransac_draws(rng), np.random.RandomState(5).random_sample(6000))


def test_native_random_sample_is_numpys_stream():
    """caelo_host_random_sample == numpy.random.RandomState(seed).random_sample(n), bit for bit (MT19937, init_genrand seeding, 53-bit
    doubles): the stream RANSAC4RT's samples come from (Match.py:182-184) -- seeds 0, 1, the largest, and the run_sequence.py range."""
    import ctypes as C
    from caelo import _ffi
    lib = _ffi.load()
    for seed in (0, 1, 5, 999, 1000, 1001, 5540, 123456789, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1):
        for n in (1, 4, 6000, 6001):
            out = np.empty(n, np.float64)
            assert lib.caelo_host_random_sample(C.c_uint32(seed), n, C.c_void_p(out.ctypes.data)) == 0
            assert np.array_equal(out, np.random.RandomState(seed).random_sample(n)), (seed, n)


def test_seqloader_reads_files_and_draws(tmp_path):
    """caelo_seqloader on host buffers: 21 scan files of ragged lengths through a ring of 3 batches of 4 on 5 threads -- every batch's
    bytes and point counts equal the files', the draws equal RandomState(seed_base + first_frame + i - 1), slots are only reused after
    their release; a truncated file fails the wait with an error that names it."""
    import ctypes as C
    from caelo import _ffi
    lib = _ffi.load()
    rs = np.random.RandomState(3)
    n, batch, ring, cap = 21, 4, 3, 700
    scans = [rs.standard_normal((int(rs.randint(1, cap + 1)), 4)).astype(np.float32) for _ in range(n)]
    scans[7] = np.zeros((0, 4), np.float32)     # an empty file is a scan without points
    paths = []
    for i, a in enumerate(scans):
        p = tmp_path / ("%06d.bin" % i)
        a.tofile(str(p))
        paths.append(str(p).encode())

    def run(paths_, first=40, seed_base=1000):
        arr = (C.c_char_p * len(paths_))(*paths_)
        sb = int(lib.caelo_seqloader_slot_bytes(batch, cap))
        assert sb == batch * cap * 16 + batch * 6000 * 8
        ring_h = np.full((ring, sb), 0xFF, np.uint8)
        keep_h = np.zeros((5, batch, 6000), np.float64)      # a longer ring for the host half's copy of the draws
        h = C.c_void_p()
        _ffi.check(lib.caelo_seqloader_create(arr, len(paths_), first, batch, ring, cap, C.c_void_p(ring_h.ctypes.data), C.c_void_p(keep_h.ctypes.data), 5,
                                              seed_base, 5, C.byref(h)))
        try:
            slot = C.c_int32(-1)
            npts = (C.c_int64 * batch)()
            nb = (len(paths_) + batch - 1) // batch
            for b in range(nb):
                _ffi.check(lib.caelo_seqloader_wait(h, b, C.byref(slot), npts))
                assert slot.value == b % ring
                sc = ring_h[slot.value, :batch * cap * 16].view(np.float32).reshape(batch, cap, 4)
                dr = ring_h[slot.value, batch * cap * 16:].view(np.float64).reshape(batch, 6000)
                for j in range(batch):
                    i = b * batch + j
                    if i >= len(paths_):
                        assert npts[j] == 0
                        continue
                    assert npts[j] == len(scans[i])
                    assert np.array_equal(sc[j, :npts[j]], scans[i])
                    want = np.random.RandomState(seed_base + first + i - 1).random_sample(6000)
                    assert np.array_equal(dr[j], want) and np.array_equal(keep_h[b % 5, j], want)
                ring_h[slot.value] = 0xFF       # (a slot handed back may be overwritten at once)
                _ffi.check(lib.caelo_seqloader_release(h, b))
            st = (C.c_int64 * 3)()
            _ffi.check(lib.caelo_seqloader_stats(h, st))
        finally:
            lib.caelo_seqloader_destroy(h)

    run(paths)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\0" * 24)
    with pytest.raises(_ffi.CaeloError, match="bad.bin"):
        run(paths[:5] + [str(bad).encode()] + paths[5:])
