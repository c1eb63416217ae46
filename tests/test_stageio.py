"""CPU: SURVEY 8f-1 / 8f-2 -- pose chaining, KITTI calibration / pose files and the reference's .mat stage
artefacts, against fixtures produced by the reference itself (tools/make_goldens.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def seq():
    return np.load(os.path.join(GOLDEN, "sequence_20.npz"))


def test_chain_poses_equals_reference_statements(seq):
    """PoseEstimation.py:202-207,:232,:253-267 were EXECUTED from the reference source when the fixture was made."""
    from caelo import stageio
    from caelo import dist as cdist
    for name in ("identity", "kitti"):
        tr = seq["tr_" + name].reshape(3, 4)
        poses = stageio.chain_poses(seq["rel_rt"], tr)
        assert poses.dtype == np.float32 and poses.shape == (20, 12)
        assert np.abs(poses - seq["poses_" + name]).max() <= 2e-6 * np.abs(seq["poses_" + name]).max()
    assert np.array_equal(cdist.chain_poses(seq["rel_rt"]), stageio.chain_poses(seq["rel_rt"], None))
    # the synthetic trajectory moves ~0.9 m per frame along x: 19 pairs
    assert 15.0 < seq["poses_identity"][-1, 3] < 19.0


def test_calib_and_pose_files(tmp_path, seq):
    from caelo import stageio
    tr = seq["tr_kitti"]
    ref_style = tmp_path / "calib_.txt"          # label-free rows, Tr is row 4 (PoseEstimation.py:202-203)
    np.savetxt(ref_style, np.vstack([np.arange(12.0) + i for i in range(4)] + [tr]))
    kitti_style = tmp_path / "calib.txt"
    kitti_style.write_text("".join("P%d: %s\n" % (i, " ".join("%.12e" % v for v in np.arange(12.0))) for i in range(4))
                           + "Tr: " + " ".join("%.12e" % v for v in tr) + "\n")
    for p in (ref_style, kitti_style):
        got = stageio.read_calib_tr(str(p))
        assert got.dtype == np.float32 and got.shape == (3, 4) and np.array_equal(got, tr.reshape(3, 4).astype(np.float32))
    out = tmp_path / "poses_" / "00.txt"
    stageio.write_poses(str(out), seq["poses_kitti"])
    back = stageio.read_poses(str(out))
    assert back.shape == (20, 12) and np.array_equal(back.astype(np.float32), seq["poses_kitti"])
    assert len(out.read_text().splitlines()[0].split()) == 12                      # KITTI devkit row


def test_block_structures_from_allvoxels0():
    """VoxelModel/*.mat carries avlBlocksList / cntVoxelsLength / AllVoxels (BatchVoxelization.py:61-62); the engine
    only produces AllVoxels0/1/2 -- the block structures follow from AllVoxels0 exactly."""
    from caelo import stageio
    g = np.load(os.path.join(GOLDEN, "voxel_blocks.npz"))
    avl, cnt, local = stageio.block_structures(g["AllVoxels0"])
    for got, want in ((avl, g["avlBlocksList"]), (cnt, g["cntVoxelsLength"]), (local, g["AllVoxels"])):
        assert got.dtype == want.dtype and np.array_equal(got, want)
    avl, cnt, local = stageio.block_structures(np.zeros((0, 3), np.int16))
    assert avl.shape == (0, 3) and cnt.tolist() == [0] and local.shape == (0, 3)


def test_mat_artefacts_round_trip_like_the_reference_loaders(tmp_path):
    from scipy import io
    from caelo import stageio
    g = np.load(os.path.join(GOLDEN, "voxel_blocks.npz"))
    raw = str(tmp_path / "00" / "velodyne" / "000007.bin")
    rs = np.random.RandomState(0)
    kp = rs.uniform(-30, 30, (1024, 3)).astype(np.float32); feats = rs.uniform(-1, 1, (1024, 60)).astype(np.float32)
    ring = rs.uniform(-1, 1, (69, 1800, 5)).astype(np.float32); cnt = rs.randint(0, 3, (69, 1800)).astype(np.int32)
    # --- writers
    p_ring = stageio.save_spherical_ring(raw, ring, cnt)
    p_kp = stageio.save_keypts(raw, kp)
    p_vox = stageio.save_voxel_model(raw, g["AllVoxels0"], g["AllVoxels1"], g["AllVoxels2"])
    p_feat = stageio.save_features(raw, kp, feats)
    p_in = stageio.save_inliers(str(tmp_path / "00"), 6, 7, np.array([5, 9, 11]), np.array([0, 1, 2]))
    assert p_ring.endswith("00/SphericalRing/000007.bin.mat") and p_kp.endswith("00/KeyPts/000007.bin.mat")
    assert p_vox.endswith("00/VoxelModel/000007.bin.mat") and p_feat.endswith("00/Features/000007.bin.mat")
    assert p_in.endswith("00/InliersIdx/000006-000007.bin.mat")
    # --- read back the way the reference does (Match.py:28-72, SphericalRing.py:389-401): io.loadmat + key lookup
    m = io.loadmat(p_vox)
    assert np.array_equal(m["avlBlocksList"], g["avlBlocksList"]) and np.array_equal(m["cntVoxelsLength"].flatten(), g["cntVoxelsLength"])
    for k in ("AllVoxels", "AllVoxels0", "AllVoxels1", "AllVoxels2"):
        assert m[k].dtype == np.int16 and np.array_equal(m[k], g[k])
    m = io.loadmat(p_feat)
    assert np.array_equal(m["KeyPts"], kp) and np.array_equal(m["Features"], feats) and m["Weights"].shape == (1024, 1)
    m = io.loadmat(p_kp)
    assert np.array_equal(m["KeyPts"], kp) and set(("ExtendedKeyPts", "PlanarPts")) <= set(m)
    # --- our loaders
    r2, c2 = stageio.load_spherical_ring(raw)
    assert np.array_equal(r2, ring) and np.array_equal(c2, cnt)
    k2, a0, a1, a2 = stageio.load_voxel_model_and_keypts(raw)
    assert np.array_equal(k2, kp) and np.array_equal(a0, g["AllVoxels0"]) and np.array_equal(a2, g["AllVoxels2"])
    k3, f3, w3 = stageio.load_keypts_and_features(raw)
    assert np.array_equal(f3, feats) and np.array_equal(k3, kp) and w3.shape == (1024, 1)
    i0, i1 = stageio.load_inliers(str(tmp_path / "00"), 6, 7)
    assert i0.tolist() == [5, 9, 11] and i1.tolist() == [0, 1, 2]
    # --- KITTI scan
    pc = rs.uniform(-50, 50, (1000, 4)).astype(np.float32)
    os.makedirs(os.path.dirname(raw), exist_ok=True)
    pc.tofile(raw)
    assert np.array_equal(stageio.read_scan(raw), pc)


def test_sequence_golden_is_self_consistent(seq):
    assert seq["rel_rt"].shape == (19, 12) and seq["success"].all() and (seq["threshold"] == np.float32(0.4)).all()
    assert (seq["n_inliers"] > 100).all() and (seq["n_key"] == 1024).all()
    R = seq["rel_rt"][:, :9].reshape(-1, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-5          # rotations


@pytest.fixture(scope="module")
def ref_mat_files(tmp_path_factory):
    """tests/golden/mat_stage_files.npz: .mat files WRITTEN BY THE REFERENCE's own savemat statements
    (BatchPreprocess.py:54-64,:139-148, BatchVoxelization.py:42-62, PoseEstimation.py:292-295,:297-309 -- executed by
    tools/make_goldens.py) unpacked into the reference's directory layout."""
    g = np.load(os.path.join(GOLDEN, "mat_stage_files.npz"))
    seq = tmp_path_factory.mktemp("refmat") / "00"
    for i, name in enumerate(g["file_names"]):
        path = seq / str(name)
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_bytes(g["file_%d" % i].tobytes())
    (seq / "velodyne").mkdir()
    return g, str(seq)


def test_stageio_reads_files_written_by_the_reference(ref_mat_files):
    import hashlib
    from caelo import stageio
    g, seq = ref_mat_files
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    raw0, raw1 = (os.path.join(seq, "velodyne", "%06d.bin" % f) for f in (0, 1))
    ring, cnt = stageio.load_spherical_ring(raw0)
    assert ring.shape == (69, 1800, 5) and ring.dtype == np.float32 and sha(ring) == str(g["f0_ring_sha256"])
    assert cnt.shape == (69, 1800) and sha(cnt) == str(g["f0_counter_sha256"])
    kp, a0, a1, a2 = stageio.load_voxel_model_and_keypts(raw0)
    assert np.array_equal(kp, g["f0_keypts"]) and kp.dtype == np.float32
    assert [sha(a0), sha(a1), sha(a2)] == [str(v) for v in g["f0_voxel_sha256"]] and a0.dtype == np.int16
    # the three block structures the reference stored equal what block_structures derives from ITS AllVoxels0
    from scipy import io
    m = io.loadmat(stageio.mat_path(raw0, "VoxelModel"))
    avl, cntl, local = stageio.block_structures(a0)
    assert np.array_equal(m["avlBlocksList"], avl) and np.array_equal(m["cntVoxelsLength"].ravel(), cntl) and np.array_equal(m["AllVoxels"], local)
    k2, F, W = stageio.load_keypts_and_features(raw0)
    assert np.array_equal(k2, g["f0_keypts"]) and np.array_equal(F, g["f0_features"]) and W.shape == (len(k2), 1) and (W == 1).all()
    i0, i1 = stageio.load_inliers(seq, 0, 1)
    assert np.array_equal(i0, g["inliers_idx0"]) and np.array_equal(i1, g["inliers_idx1"]) and len(i0) > 100
    ke = io.loadmat(stageio.mat_path(raw1, "KeyPts"))
    assert np.array_equal(ke["KeyPts"], g["f1_keypts"]) and sha(np.asarray(ke["ExtendedKeyPts"], np.float32)) == str(g["f1_ext_sha256"])
    assert ke["PlanarPts"].size == 0                                    # SphericalRing.py:219,285: always empty
    assert bool(g["reference_loaders_read_stageio_files"])              # the other direction, asserted at generation time
    # our writers produce files with the same variables, dtypes and shapes as the reference's
    mine = stageio.save_voxel_model(os.path.join(seq, "mine", "velodyne", "000000.bin"), a0, a1, a2)
    m2 = io.loadmat(mine)
    for key in ("avlBlocksList", "cntVoxelsLength", "AllVoxels", "AllVoxels0", "AllVoxels1", "AllVoxels2"):
        assert m2[key].dtype == m[key].dtype and np.array_equal(m2[key], m[key]), key
