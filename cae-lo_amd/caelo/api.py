"""Drop-in call surface: the reference's hot-path functions, same names / argument order / return
tuples / dtypes, served by the HIP engine.

    reference                                   here
    SphericalRing.ProjectPC2SphericalRing :72   ProjectPC2SphericalRing(PC)
    SphericalRing.GetKeyPtsByAE           :113  GetKeyPtsByAE(SphericalRing, GridCounter, RespondImg)
    SphericalRing.GetKeyPtsFromRawFileName:389  GetKeyPtsFromRawFileName(rawFileFullPath, RespondLayer)
    Voxel.Voxelization                    :100  Voxelization(PC)
    Voxel.GetPatchesList                  :177  GetPatchesList(Pts, AllVoxels0, AllVoxels1, AllVoxels2)
    Match.GetFeaturesFromPatches          :130  GetFeaturesFromPatches(PatchEncoder, PatchesList)
    Match.SolveRT / RANSAC4RT             :138/:162
    Match.SolveRelativePose               :241  SolveRelativePose(PC0, F0, W0, PC1, F1, W1)
    keras.models.load_model (Match.py:313,324)  load_model(h5_path) -> object with .predict

Arguments may be NumPy arrays (results come back as NumPy, like the reference) or torch tensors on
the GPU (results stay on the GPU).  Every array-level function runs on the GPU; there is no CPU
path.  Errors follow the reference: AssertionError / IndexError / ValueError for the conditions
listed in SURVEY 8b, RANSAC failure as a return value.
"""
import os

import warnings

import numpy as np
import torch

from . import engine as _eng
from .engine import default_engine, raise_status

nLines, ImgW, ImgH, CropWidth_SphericalRing = 64, 1800, 69, 8
Channels4AE = [0, 1, 2]


def _is_np(x):
    return isinstance(x, np.ndarray)


def _dev(x, dtype):
    e = default_engine()
    if isinstance(x, torch.Tensor):
        return x.to(device=e.device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype={torch.float32: np.float32, torch.int32: np.int32,
                                                            torch.int16: np.int16, torch.int64: np.int64,
                                                            torch.float64: np.float64}[dtype])).to(e.device)


def _out(t, as_np):
    return t.cpu().numpy() if as_np else t


# ---------------------------------------------------------------------------------------------
def ProjectPC2SphericalRing(PC):
    """SphericalRing.py:72-94 -> (Image_float [69,1800,5] f32, GridCounter [69,1800] i32)."""
    assert PC.shape[0] > 3 and PC.shape[1] == 4
    e = default_engine()
    ring, cnt, st = e.project(_dev(PC, torch.float32))
    raise_status(int(st.item()))
    return _out(ring, _is_np(PC)), _out(cnt, _is_np(PC))


class _Model:
    """What ``load_model`` returns: only ``.predict(ndarray) -> ndarray`` is part of the contract
    (SphericalRing.py:407, Match.py:131)."""

    def __init__(self, kind, path):
        self.kind, self.path = kind, path

    def predict(self, x):
        e = default_engine()
        as_np = _is_np(x)
        xd = _dev(x, torch.float32)
        if self.kind == "respond":
            assert xd.dim() == 4 and tuple(xd.shape[1:]) == (64, 1792, 3), "expected [B,64,1792,3]"
            out = torch.stack([e.respond(xd[b]) for b in range(xd.shape[0])])
        else:
            assert xd.dim() == 5 and tuple(xd.shape[1:]) == (16, 16, 16, 1), "expected [K,16,16,16,1]"
            out = e.encode(e.pack_patches(xd), group=1)
        return _out(out, as_np)


def load_model(path):
    """Replacement for keras.models.load_model on the two shipped .h5 files (Dirs.py:29-30)."""
    kind = default_engine().load_weights(path)
    return _Model(kind, path)


def GetKeyPtsByAE(SphericalRing, GridCounter, RespondImg):
    """SphericalRing.py:113-291 -> (KeyPts [K,3] f32, KeyPixels [K,2] i64, PlanarPts [0] f32).
    A 5-channel ring = demo mode (:414); a cropped 3-channel ring = batch mode
    (BatchPreprocess.py:97-98,131-136)."""
    e = default_engine()
    as_np = _is_np(SphericalRing)
    ring = _dev(SphericalRing, torch.float32)
    cnt = _dev(np.asarray(GridCounter, dtype=np.int32) if _is_np(GridCounter) else GridCounter, torch.int32)
    resp = _dev(RespondImg, torch.float32)
    kpts, kpix, nkey, st = e.keypoints(ring, cnt, resp)
    k = int(nkey.item())
    raise_status(int(st.item()))
    planar = np.array([], dtype=np.float32) if as_np else torch.empty(0, dtype=torch.float32, device=e.device)
    return _out(kpts[:k], as_np), _out(kpix[:k], as_np), planar


def ExtendKeyPtsInShpericalRing(SphericalRing, GridCounter, KeyPixels):
    """SphericalRing.py:294-317.  Returns ExtendedKeyPts [M,3] f32 and, like the reference, zeroes the 13 x 13
    windows of the caller's GridCounter in place (NumPy array or tensor).  Keypixels must come from GetKeyPtsByAE
    (rows [8,56), cols [8,1784)): windows that leave the image are clipped, not wrapped like NumPy's negative slices."""
    e = default_engine()
    as_np = _is_np(SphericalRing)
    ring = _dev(SphericalRing, torch.float32)
    cnt = _dev(GridCounter, torch.int32)
    kpix = _dev(KeyPixels, torch.int64)
    if kpix.shape[0] == 0:
        return _out(torch.empty((0, 3), dtype=torch.float32, device=e.device), as_np)
    ext, n_ext = e.extend_keypts(ring, cnt, kpix)
    if _is_np(GridCounter):
        GridCounter[...] = cnt.cpu().numpy().astype(GridCounter.dtype)      # the in-place side effect (:307)
    elif cnt.data_ptr() != GridCounter.data_ptr():
        GridCounter.copy_(cnt.to(GridCounter.dtype))
    return _out(ext[: int(n_ext.item())], as_np)


def RotateMat2EulerAngle_XYZ(R):
    """Transformations.py:181-186 (degrees)."""
    import math
    return np.array([math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], math.sqrt(R[2, 1] ** 2 + R[2, 2] ** 2)),
                     math.atan2(R[1, 0], R[0, 0])]) * (180.0 / math.pi)


def _icp_points(P, cols=3):
    P = np.ascontiguousarray(P)[:, 0:cols] if _is_np(P) else P[:, 0:cols]
    return _dev(P, torch.float32).contiguous()


def ICP(PC0, PC1, maxIterTimes=50, minIterTimes=20 - 1, inlierThreshold=0.5, smallShiftThreshold=0.05, decay_rate=0.9, ep=0.001):
    """MyICP.py:26-72: point-to-point ICP of PC1 onto PC0 (the commented-out alternative at RefinePoses.py:295).  The whole
    loop -- nearest neighbours, inlier gate, SolveRT, the update of PC1, the Euler-angle stop rule, the threshold decay --
    runs on the device (caelo_icp); one synchronisation at the end.  -> (R_star [3,3] f64, T_star [3,1] f64, isSuccess)."""
    e = default_engine()
    res = e.icp(_icp_points(PC0), _icp_points(PC1).clone(), threshold0=inlierThreshold, decay0=decay_rate,
                small_shift=smallShiftThreshold, ep=ep, max_iter=maxIterTimes, min_iter=minIterTimes, min_pairs=100, fail_only_first=0)
    r = e.icp_result(res)
    return np.array(r.R_star, np.float64).reshape(3, 3), np.array(r.T_star, np.float64).reshape(3, 1), bool(r.success)


def ICP_Pt2PtAndPt2Plane(PC0, PC1, PtsWithNorm0, PtsWithNorm1, maxIterTimes=50, minIterTimes=20 - 1, inlierThreshold0=0.5,
                         decay_rate0=0.9, inlierThreshold1=2.0, decay_rate1=0.5, smallShiftThreshold=0.1, ep=0.01, rng=None,
                         return_info=False):
    """MyICP.py:127-201 (caller RefinePoses.py:291-293): every iteration fits ONE rigid motion to the point pairs and the
    planar pairs (foot of the perpendicular, MyICP.py:88-114).  Device-resident loop (caelo_icp).  ``rng``: RandomState for
    the subsampling of more than 2000 planar points (:135-140; NumPy's global generator when None, like the reference).
    Empty planar sets raise ValueError like sklearn's fit at :94 -- which is what the reference's own artefacts lead to
    (GetKeyPtsByAE always returns an empty PlanarPts, SphericalRing.py:219,285)."""
    e = default_engine()
    PN0, PN1 = np.asarray(PtsWithNorm0), np.asarray(PtsWithNorm1)
    for a in (PN0, PN1):
        if a.ndim != 2 or a.shape[0] == 0 or a.shape[1] == 0:
            raise ValueError("Found array with 0 sample(s) (shape=%s) while a minimum of 1 is required." % (a.shape,))
    nMaxPts = 2000                                                                        # :135
    if PN1.shape[0] > nMaxPts:
        rs = np.random.mtrand._rand if rng is None else rng
        RandIdxes = np.array(rs.random_sample((nMaxPts,)) * PN1.shape[0], dtype=np.int32)   # :137-139
        PN1 = PN1[RandIdxes, :]
    res = e.icp(_icp_points(PC0), _icp_points(PC1).clone(), _icp_points(PN0, 6), _icp_points(PN1, 6).clone(),
                threshold0=inlierThreshold0, threshold1=inlierThreshold1, decay0=decay_rate0, decay1=decay_rate1,
                small_shift=smallShiftThreshold, ep=ep, max_iter=maxIterTimes, min_iter=minIterTimes, min_pairs=200, fail_only_first=1)
    r = e.icp_result(res)
    out = (np.array(r.R_star, np.float64).reshape(3, 3), np.array(r.T_star, np.float64).reshape(3, 1), bool(r.success))
    return out + (r,) if return_info else out


def RefinementCore(poses, ExtKeyPts0, PlanarPts0, ExtKeyPts1, PlanarPts1, iFrame0, iFrame1, relRs, relTs, inlierThreshold0, Tr, rng=None):
    """RefinePoses.py:273-334 (the two file reads at :276-277 replaced by their arrays, the module globals R_Tr ... by
    ``Tr``): re-register frame iFrame1 on iFrame0 with ICP_Pt2PtAndPt2Plane on the device, reject implausible changes,
    update the pose of iFrame1 and forward-update the following poses.  -> (flag, poses_, relRs, relTs)."""
    from . import refine
    return refine.RefinementCore(poses, ExtKeyPts0, PlanarPts0, ExtKeyPts1, PlanarPts1, iFrame0, iFrame1, relRs, relTs, inlierThreshold0, Tr,
                                 icp=ICP_Pt2PtAndPt2Plane, rng=rng)


def GetKeyPtsFromRawFileName(rawFileFullPath, RespondLayer):
    """SphericalRing.py:389-416: reads <seq>/SphericalRing/<name>.mat written by BatchPreprocess."""
    from scipy import io
    baseDir = os.path.dirname(os.path.dirname(rawFileFullPath))
    mat = io.loadmat(os.path.join(baseDir, "SphericalRing", os.path.basename(rawFileFullPath) + ".mat"))
    SphericalRing, GridCounter = mat["SphericalRing"], mat["GridCounter"]
    x = SphericalRing[0:nLines, 0:ImgW - CropWidth_SphericalRing, :][:, :, Channels4AE]
    RespondImg = np.squeeze(RespondLayer.predict(np.ascontiguousarray(x).reshape((1,) + x.shape)))
    return GetKeyPtsByAE(SphericalRing, GridCounter, RespondImg)


class _Blocks:
    """Member 0 of Voxelization's tuple (Voxel.py:59-72,:126-143): ``Blocks[ix][iy][iz]`` is ``[False]`` for an untouched
    64^3 block and ``[True, occupancy int8 [64,64,64], [local voxel indices], [global voxel indices]]`` for a touched
    one.  The reference deep-copies a 156 x 156 x 23 nested list per call; here the same indexing is served on demand
    from AllVoxels0 and its block structures (identical values, built when asked for)."""

    def __init__(self, avl, cnt, local, a0):
        self._dims = (156, 156, 23)                                   # nBlocksL, nBlocksW, nBlocksH (Voxel.py:42-44)
        self._index = {tuple(int(v) for v in b): i for i, b in enumerate(avl)}
        self._cnt, self._local, self._a0, self._cache = cnt, local, a0, {}

    def __len__(self):
        return self._dims[0]

    def block(self, ix, iy, iz):
        key = (ix, iy, iz)
        if key not in self._index:
            return [False]
        if key not in self._cache:
            i = self._index[key]
            lo, hi = int(self._cnt[i]), int(self._cnt[i + 1])
            occ = np.zeros((64, 64, 64), dtype=np.int8)
            loc = self._local[lo:hi]
            occ[loc[:, 0], loc[:, 1], loc[:, 2]] = 1
            self._cache[key] = [True, occ, loc.tolist(), self._a0[lo:hi].tolist()]
        return self._cache[key]

    class _Level:
        def __init__(self, owner, prefix):
            self._o, self._p = owner, prefix

        def __len__(self):
            return self._o._dims[len(self._p)]

        def __getitem__(self, i):
            n = self._o._dims[len(self._p)]
            if not -n <= i < n:
                raise IndexError("list index out of range")
            p = self._p + (i % n,)
            return self._o.block(*p) if len(p) == 3 else _Blocks._Level(self._o, p)

    def __getitem__(self, i):
        return _Blocks._Level(self, ())[i]


def Voxelization(PC):
    """Voxel.py:100-173 -> the reference's 9-tuple (Blocks, VoxelModel1, VoxelModel2, avlBlocksList, cntVoxelsLength,
    AllVoxels, AllVoxels0, AllVoxels1, AllVoxels2).  AllVoxels0/1/2 come from the device (values and first-touch
    order); the block structures (:161-172) and the dense int8 models (:106-107,:153-158) are derived from them on the
    host -- they are consumed only by code outside the hot path (BatchVoxelization.py:61-62, Match.py:28-43)."""
    from . import stageio
    e = default_engine()
    as_np = _is_np(PC)
    pc = _dev(PC, torch.float32)
    vmap, st = e.voxelize(pc)
    raise_status(int(st.item()) & ~_eng.ST_FEW_VOXELS)  # the reference only complains later, in GetPatchesList
    a0, a1, a2 = e.voxmap_export(vmap, pc.shape[0])
    h0, h1, h2 = (t.cpu().numpy() for t in (a0, a1, a2))
    avl, cnt, local = stageio.block_structures(h0)
    vm1 = np.zeros((1248, 1248, 184), dtype=np.int8)                  # nBlocks * 64 / 8 (Voxel.py:106)
    vm2 = np.zeros((312, 312, 46), dtype=np.int8)                     # nBlocks * 64 / 32 (:107)
    vm1[h1[:, 0], h1[:, 1], h1[:, 2]] = 1
    vm2[h2[:, 0], h2[:, 1], h2[:, 2]] = 1
    return (_Blocks(avl, cnt, local, h0), vm1, vm2, avl, cnt, local, _out(a0, as_np), _out(a1, as_np), _out(a2, as_np))


def GetPatchesBits(Pts, AllVoxels0, AllVoxels1, AllVoxels2):
    """GetPatchesList without the dense expansion: (bits [K,3,64] int64, flags [K,3] uint8) on device."""
    e = default_engine()
    vmap, st = e.voxmap_from_lists(_dev(AllVoxels0, torch.int16), _dev(AllVoxels1, torch.int16),
                                   _dev(AllVoxels2, torch.int16))
    bits, flags = e.patches(vmap, _dev(Pts, torch.float32), None, st)
    raise_status(int(st.item()))
    return bits, flags


def GetPatchesList(Pts, AllVoxels0, AllVoxels1, AllVoxels2):
    """Voxel.py:177-216 -> (Pts, [P0, P1, P2]) with P_s [K,16,16,16,1] f32 in {0,1}."""
    e = default_engine()
    as_np = _is_np(Pts)
    bits, flags = GetPatchesBits(Pts, AllVoxels0, AllVoxels1, AllVoxels2)
    n_tie = int(((flags & 2) != 0).sum().item())
    if n_tie:
        # the 496-nearest cut of Voxel.py:195-196 splits a class of equidistant voxels and the redo in the library's order (kd-tree from
        # 994 voxels on, np.argpartition below) did not happen: the kd build gave up on its work budget (kdorder.hip) -- canonical rule
        warnings.warn("GetPatchesList: %d patch(es) truncated inside a tie of equidistant voxels could not be redone in the library's "
                      "order; they may differ from the reference's in the cut class (flags & 2)" % n_tie, RuntimeWarning)
    out = [_out(e.unpack_patches(bits[:, s, :].contiguous()), as_np) for s in range(3)]
    return Pts, out


def GetFeaturesFromPatches(PatchEncoder, PatchesList):
    """Match.py:130-135."""
    f = [PatchEncoder.predict(p) for p in PatchesList]
    return np.c_[f[0], f[1], f[2]] if _is_np(f[0]) else torch.cat(f, dim=1)


def SolveRT(Pairs0, Pairs1):
    """Match.py:138-158 -> (R [3,3], T [3,1], isCredible)."""
    e = default_engine()
    as_np = _is_np(Pairs0)
    R, T, cred = e.solve_rt(_dev(Pairs0, torch.float32), _dev(Pairs1, torch.float32))
    return _out(R, as_np), _out(T, as_np), int(cred.item())


def _ransac(pc0, pc1, pair_idx, rng):
    """Shared by RANSAC4RT / SolveRelativePose.  ``rng``: RandomState or None (NumPy's global RNG,
    like the reference).  The stream is advanced by exactly the draws the reference's loop would
    have consumed (4 per iteration), whatever was pre-drawn for the GPU.

    The kernels score the 500 hypotheses of a level and leave a certificate (an upper bound per hypothesis on the count the
    reference's own float32 / BLAS arithmetic can reach); the host half (caelo.hostexact, csrc/certify.hip) replays
    Match.py:181-214 over the bounds and re-evaluates the deciding hypotheses through NumPy's BLAS / LAPACK entry points:
    the inlier mask, R_star / T_star and the refit are the reference's bits on this host."""
    from . import hostexact, _ffi
    e = default_engine()
    rs = np.random.mtrand._rand if rng is None else rng
    state = rs.get_state()
    draws = rs.random_sample(6000)
    cert = e.new_cert(1)
    res, mask = e.ransac(pc0, pc1, pair_idx, torch.from_numpy(draws).to(e.device), cert=cert[0])
    results, masks, _, status = e.certify(cert, [draws])
    n = int(pc1.shape[0])
    if status[0] == 2:   # more than 1024 pairs: no certificate -- the reference's loop on the host arrays, hypothesis by hypothesis
        p0 = pc0[:, :3][pair_idx].detach().cpu().numpy()
        p1 = pc1[:, :3].detach().cpu().numpy()
        r, m, _ = hostexact.ransac(p0, p1, draws)
        mask = torch.from_numpy(m.astype(np.uint8)).to(e.device)
    else:
        assert status[0] == 0
        r = results[0]
        mask = torch.from_numpy(masks[0, :n].copy()).to(e.device)
    r = _ffi.PoseResult.from_buffer_copy(r.tobytes())   # (attribute access like Engine.pose_result)
    used = 4 * (r.best_trial // 500 * 500 + r.iterations if r.success else 1500)
    rs.set_state(state)
    rs.random_sample(used)
    return r, mask


def RANSAC4RT(Pairs0, Pairs1, Weights0=None, Weights1=None, rng=None):
    """Match.py:162-218 -> (R, T, isSuccess, inlierIdx bool[N], residualThreshold)."""
    e = default_engine()
    as_np = _is_np(Pairs0)
    p0, p1 = _dev(Pairs0, torch.float32), _dev(Pairs1, torch.float32)
    idx = torch.arange(p1.shape[0], device=e.device, dtype=torch.int64)
    r, mask = _ransac(p0, p1, idx, rng)
    R = np.array(r.R_ransac, dtype=np.float32).reshape(3, 3)
    T = np.array(r.T_ransac, dtype=np.float32).reshape(3, 1)
    if r.best_trial < 0:   # no hypothesis was ever accepted: the reference returns its float64 initial values (:177-178)
        R, T = np.eye(3, dtype=np.float64), np.zeros((3, 1), dtype=np.float64)
    m = mask.bool()
    thr = {0: 0.4, 1: 0.8, 2: 1.6}[int(round(np.log2(r.threshold / 0.4)))]
    if as_np:
        return R, T, bool(r.success), m.cpu().numpy(), thr
    return torch.from_numpy(R).to(e.device), torch.from_numpy(T).to(e.device), bool(r.success), m, thr


def SolveRelativePose(OriPC0, OriCodes0, Weights0, OriPC1, OriCodes1, Weights1, rng=None):
    """Match.py:241-283 -> (R, T, isSuccess, inliersIdx0, inliersIdx1, residualThreshold)."""
    e = default_engine()
    as_np = _is_np(OriPC0)
    pc0, pc1 = _dev(OriPC0, torch.float32), _dev(OriPC1, torch.float32)
    f0, f1 = _dev(OriCodes0, torch.float32), _dev(OriCodes1, torch.float32)
    pair_idx = e.match(f0, f1)
    r, mask = _ransac(pc0, pc1, pair_idx, rng)
    m = mask.bool()
    i1 = torch.nonzero(m).flatten()
    i0 = pair_idx[i1]
    thr = {0: 0.4, 1: 0.8, 2: 1.6}[int(round(np.log2(r.threshold / 0.4)))]
    R = np.array(r.R, dtype=np.float32).reshape(3, 3)
    T = np.array(r.T, dtype=np.float32).reshape(3, 1)
    if as_np:
        return R, T, bool(r.success), i0.cpu().numpy(), i1.cpu().numpy(), thr
    return torch.from_numpy(R).to(e.device), torch.from_numpy(T).to(e.device), bool(r.success), i0, i1, thr
