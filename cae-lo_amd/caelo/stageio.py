"""On-disk formats of the reference's stages: KITTI scans, calibration and pose
files, and the per-frame ``.mat`` artefacts the reference's drivers exchange -- so that this engine can
consume what the reference wrote and the reference (``Match.py`` loaders, ``RefinePoses.py``) can
consume what this engine writes.  Host side, NumPy + scipy.io only.

Directory convention (Match.py:29-31,47-49,67-68): for a raw scan ``<seq>/velodyne/000123.bin`` the
artefacts live in ``<seq>/<Folder>/000123.bin.mat``.
"""
import os

import numpy as np

BLOCK_SIZE = 64  # Voxel.py:24-27


# ---- KITTI scan / calibration / poses (8f-1) ---------------------------------------------------------
def read_scan(path):
    """``np.fromfile(..., float32).reshape(-1, 4)`` -- the KITTI velodyne layout every reference driver
    reads (BatchPreprocess.py:46-47)."""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


def read_calib_tr(path):
    """Velodyne -> camera transform ``Tr`` [3,4] f32.  The reference reads its own label-free
    ``calib_.txt`` with ``np.loadtxt`` and takes row 4 (PoseEstimation.py:202-203); KITTI's original
    ``calib.txt`` (``P0: ... Tr: ...``) is accepted as well."""
    with open(path) as f:
        lines = [ln.split() for ln in f if ln.strip()]
    if lines and lines[0][0].endswith(":"):
        row = [ln for ln in lines if ln[0] in ("Tr:", "Tr_velo_to_cam:")][0][1:]
    else:
        row = lines[4]
    return np.array(np.asarray(row, dtype=np.float64).reshape(3, 4), dtype=np.float32)


def write_poses(path, poses):
    """``poses_/NN.txt``: one 12-float row per frame, ``np.savetxt`` defaults (PoseEstimation.py:273-277)."""
    poses = np.array(poses, dtype=np.float32).reshape(len(poses), 12)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savetxt(path, poses)


def read_poses(path):
    return np.loadtxt(path, dtype=np.float64).reshape(-1, 12)


def _homogeneous(m34):
    """[3,4] -> [4,4] float32 rigid transform."""
    h = np.eye(4, dtype=np.float32)
    h[:3, :] = m34
    return h


def chain_poses(rel_rt, Tr=None):
    """Per-pair LiDAR motions [F-1,12] (R row-major | T) -> camera-frame KITTI poses [F,12], frame 0 = identity.

    What PoseEstimation.py:230-267 computes, written as homogeneous 4x4 algebra: a LiDAR-frame motion M becomes
    the camera-frame motion  C = Tr . M . Tr^-1  (calibration ``Tr`` [3,4], identity if None), and pose k+1 is
    pose k . C_k.  float32 throughout, as the reference's arrays are."""
    motions = np.asarray(rel_rt, dtype=np.float32).reshape(-1, 12)
    calib = _homogeneous(np.eye(3, 4) if Tr is None else np.asarray(Tr, dtype=np.float32).reshape(3, 4))
    calib_inv = np.eye(4, dtype=np.float32)
    calib_inv[:3, :3] = np.linalg.inv(calib[:3, :3])
    calib_inv[:3, 3] = -calib_inv[:3, :3] @ calib[:3, 3]
    out = np.empty((len(motions) + 1, 12), dtype=np.float32)
    pose = np.eye(4, dtype=np.float32)
    out[0] = pose[:3].ravel()
    for k, m in enumerate(motions):
        lidar = _homogeneous(np.c_[m[:9].reshape(3, 3), m[9:]])
        pose = pose @ (calib @ (lidar @ calib_inv))
        pose[3] = (0.0, 0.0, 0.0, 1.0)
        out[k + 1] = pose[:3].ravel()
    return out


# ---- .mat stage artefacts (8f-2) ------------------------------------------------------------------------
def mat_path(raw_file, folder):
    """<seq>/<folder>/<basename>.mat for a raw scan <seq>/velodyne/<basename> (Match.py:29-31)."""
    base = os.path.dirname(os.path.dirname(raw_file))
    return os.path.join(base, folder, os.path.basename(raw_file) + ".mat")


def _save(path, d):
    from scipy import io
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    io.savemat(path, d)
    return path


def _load(path):
    from scipy import io
    return io.loadmat(path)


def save_spherical_ring(raw_file, SphericalRing, GridCounter, folder="SphericalRing"):
    """BatchPreprocess.py:59-64."""
    return _save(mat_path(raw_file, folder), {"SphericalRing": np.asarray(SphericalRing), "GridCounter": np.asarray(GridCounter)})


def load_spherical_ring(raw_file, folder="SphericalRing"):
    """SphericalRing.py:389-401 -> (SphericalRing [69,1800,5] f32, GridCounter [69,1800])."""
    m = _load(mat_path(raw_file, folder))
    return m["SphericalRing"], m["GridCounter"]


def save_keypts(raw_file, KeyPts, ExtendedKeyPts=None, PlanarPts=None, folder="KeyPts"):
    """BatchPreprocess.py:140-148 (file name there: <frame:06d>.bin.mat = the raw scan's basename + '.mat')."""
    empty = np.zeros((0, 3), np.float32)
    return _save(mat_path(raw_file, folder), {"KeyPts": np.asarray(KeyPts), "PlanarPts": empty if PlanarPts is None else np.asarray(PlanarPts),
                                              "ExtendedKeyPts": empty if ExtendedKeyPts is None else np.asarray(ExtendedKeyPts)})


def block_structures(AllVoxels0):
    """(avlBlocksList i16 [B,3], cntVoxelsLength i32 [B+1], AllVoxels i16 [n0,3]) of Voxel.py:161-172 from
    AllVoxels0 alone: AllVoxels0 is the per-block concatenation, blocks in first-touch order, so the block of a
    row is ``AllVoxels0 // 64``, a new block starts wherever that changes, and AllVoxels is the in-block index."""
    a0 = np.asarray(AllVoxels0, dtype=np.int16).reshape(-1, 3)
    blk = a0 // BLOCK_SIZE
    if len(a0) == 0:
        return np.zeros((0, 3), np.int16), np.zeros(1, np.int32), np.zeros((0, 3), np.int16)
    start = np.r_[True, np.any(blk[1:] != blk[:-1], axis=1)]
    first = np.flatnonzero(start)
    return (blk[first].astype(np.int16), np.r_[first, len(a0)].astype(np.int32),
            (a0 - blk * BLOCK_SIZE).astype(np.int16))


def save_voxel_model(raw_file, AllVoxels0, AllVoxels1, AllVoxels2, folder="VoxelModel"):
    """BatchVoxelization.py:50-62: the six arrays LoadVoxelModel / LoadVoxelModelAndKeyPts read."""
    avl, cnt, local = block_structures(AllVoxels0)
    return _save(mat_path(raw_file, folder), {"avlBlocksList": avl, "cntVoxelsLength": cnt, "AllVoxels": local,
                                              "AllVoxels0": np.asarray(AllVoxels0, np.int16), "AllVoxels1": np.asarray(AllVoxels1, np.int16),
                                              "AllVoxels2": np.asarray(AllVoxels2, np.int16)})


def load_voxel_model_and_keypts(raw_file):
    """Match.py:46-63 -> (KeyPts, AllVoxels0, AllVoxels1, AllVoxels2)."""
    v = _load(mat_path(raw_file, "VoxelModel"))
    k = _load(mat_path(raw_file, "KeyPts"))
    return k["KeyPts"], v["AllVoxels0"], v["AllVoxels1"], v["AllVoxels2"]


def save_features(raw_file, KeyPts, Features, Weights=None, folder="Features"):
    """PoseEstimation.py:280-295."""
    KeyPts = np.asarray(KeyPts)
    W = np.ones((KeyPts.shape[0], 1), np.float32) if Weights is None else np.asarray(Weights)
    return _save(mat_path(raw_file, folder), {"KeyPts": KeyPts, "Features": np.asarray(Features), "Weights": W})


def load_keypts_and_features(raw_file, folder="Features"):
    """Match.py:66-72 -> (KeyPts, Features, Weights)."""
    m = _load(mat_path(raw_file, folder))
    return m["KeyPts"], m["Features"], m["Weights"]


def save_inliers(seq_dir, iFrame0, iFrame1, inliersIdx0, inliersIdx1, folder="InliersIdx"):
    """PoseEstimation.py:297-309: <seq>/InliersIdx/<i0:06d>-<i1:06d>.bin.mat."""
    path = os.path.join(seq_dir, folder, "%06d-%06d.bin.mat" % (iFrame0, iFrame1))
    return _save(path, {"iFrame0": iFrame0, "iFrame1": iFrame1, "inliersIdx0": np.asarray(inliersIdx0),
                        "inliersIdx1": np.asarray(inliersIdx1)})


def load_inliers(seq_dir, iFrame0, iFrame1, folder="InliersIdx"):
    m = _load(os.path.join(seq_dir, folder, "%06d-%06d.bin.mat" % (iFrame0, iFrame1)))
    return m["inliersIdx0"].ravel(), m["inliersIdx1"].ravel()
