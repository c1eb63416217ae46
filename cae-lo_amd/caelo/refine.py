"""Host side of the pose refinement that follows the odometry: the 3 x 4 pose algebra of
Transformations.py and the control flow of RefinePoses.RefinementCore.  The registration itself (ICP_Pt2PtAndPt2Plane)
runs on the device (caelo_icp, csrc/icp.hip) and is passed in."""
import math

import numpy as np


def RotateMat2EulerAngle_XYZ(R):
    """Transformations.py:181-186 (degrees)."""
    return np.array([math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], math.sqrt(R[2, 1] ** 2 + R[2, 2] ** 2)),
                     math.atan2(R[1, 0], R[0, 0])]) * (180.0 / math.pi)


def GetRtFromOnePose(pose):
    """Transformations.py:164-168: a KITTI 12-float row -> (R [3,3], T [3,1])."""
    m = np.asarray(pose).reshape(3, 4)
    return m[:, 0:3], m[:, 3].reshape(3, 1)


def GetRelRtBetween2Poses(pose0, pose1):
    """Transformations.py:106-113: motion from pose0 to pose1 in the frame of pose0."""
    Ra, Ta = GetRtFromOnePose(pose0)
    Ra_inv = np.linalg.inv(Ra)
    Rb, Tb = GetRtFromOnePose(pose1)
    return Ra_inv @ Rb, Ra_inv @ Tb - Ra_inv @ Ta


def GetLidarRelRtBetween2Poses(pose0, pose1, R_Tr, T_Tr, R_Tr_inv, T_Tr_inv):
    """Transformations.py:118-125: the same motion expressed in the LiDAR frame (camera poses conjugated by Tr)."""
    Ra, Ta = GetRtFromOnePose(pose0)
    Ra_inv = np.linalg.inv(Ra)
    Rb, Tb = GetRtFromOnePose(pose1)
    R = R_Tr_inv @ (Ra_inv @ (Rb @ R_Tr))
    T = R_Tr_inv @ (Ra_inv @ (Rb @ T_Tr + Tb) - Ra_inv @ Ta) + T_Tr_inv
    return R, T


def ForwardUpdatePoses(poses, frameNum, newPose, relRs, relTs):
    """RefinePoses.py:120-145: frame ``frameNum`` takes ``newPose``; the poses after it are re-chained from the stored
    relative motions; the relative motion into frameNum is recomputed.  Inputs are not modified."""
    out_poses, out_Rs, out_Ts = np.array(poses), np.array(relRs), np.array(relTs)
    out_poses[frameNum, :] = newPose
    dR, dT = GetRelRtBetween2Poses(out_poses[frameNum - 1, :], newPose)
    out_Rs[frameNum - 1, :, :] = dR
    out_Ts[frameNum - 1, :] = dT.reshape(3,)
    for k in range(frameNum + 1, out_poses.shape[0]):
        Rp, Tp = GetRtFromOnePose(out_poses[k - 1])
        out_poses[k, :] = np.c_[Rp @ out_Rs[k - 1], Rp @ out_Ts[k - 1].reshape(3, 1) + Tp].reshape(12)
    return out_poses, out_Rs, out_Ts


def RefinementCore(poses, ExtKeyPts0, PlanarPts0, ExtKeyPts1, PlanarPts1, iFrame0, iFrame1, relRs, relTs, inlierThreshold0, Tr, icp, rng=None):
    """RefinePoses.py:273-334.  flag: -1 registration failed, 0 rejected (the relative pose would change by more than
    10 degrees or 5 m), 1 pose of iFrame1 refined and the later poses forward-updated."""
    untouched = np.array(poses)
    R_Tr, T_Tr = GetRtFromOnePose(np.asarray(Tr))
    R_Tr_inv = np.linalg.inv(R_Tr)
    T_Tr_inv = -(R_Tr_inv @ T_Tr)
    pose_a, pose_b = poses[iFrame0, :], poses[iFrame1, :]
    odoR, odoT = GetLidarRelRtBetween2Poses(pose_a, pose_b, R_Tr, T_Tr, R_Tr_inv, T_Tr_inv)                 # :283
    moved = np.array((odoR @ np.asarray(ExtKeyPts1).T + odoT).T, dtype=np.float32)                          # :284
    planar_moved = np.array(PlanarPts1)
    planar_moved[:, 0:3] = np.array((odoR @ np.asarray(PlanarPts1)[:, 0:3].T + odoT).T, dtype=np.float32)   # :286-287
    R_icp, T_icp, ok = icp(ExtKeyPts0, moved, PlanarPts0, planar_moved, maxIterTimes=50, minIterTimes=20 - 1,
                           inlierThreshold0=inlierThreshold0, decay_rate0=0.9, inlierThreshold1=5.0, decay_rate1=0.9,
                           smallShiftThreshold=0.1, ep=0.001, rng=rng)                                      # :290-293
    if not ok:
        return -1, untouched, relRs, relTs                                                                  # :297-298
    newR, newT = R_icp @ odoR, R_icp @ odoT + T_icp                                                         # :300-301
    d_euler = np.linalg.norm(RotateMat2EulerAngle_XYZ(odoR) - RotateMat2EulerAngle_XYZ(newR))              # :304-306
    if d_euler > 10 or np.linalg.norm(odoT - newT) > 5:                                                     # :307-309
        return 0, untouched, relRs, relTs
    Ra, Ta = GetRtFromOnePose(pose_a)
    camR = R_Tr @ (newR @ R_Tr_inv)                                                                         # :315
    camT = R_Tr @ (newR @ T_Tr_inv + newT) + T_Tr                                                           # :316
    refined = np.c_[Ra @ camR, Ra @ camT + Ta].reshape(12)                                                  # :317-321
    out = ForwardUpdatePoses(poses, iFrame1, refined, relRs, relTs)                                         # :326
    return (1,) + out
