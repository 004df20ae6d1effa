"""Flat bundle-adjustment problems from the reference's own data sets (fixtures under tests/golden/, packed by
tests/golden/gen_dataset_fixtures.py) -- test and bench plumbing, nothing here computes a residual.

Conventions of the data (data/vslam_set2/README.md): frame pose = map <- robot as `tx ty tz qx qy qz qw`, robot x forward /
y left / z up, camera = robot rotated into the optical frame (extrinsics q_xyzw = (-.5, .5, -.5, .5), t = 0), pixels from
K = `fx fy cx cy`.  The ORB-SLAM2 tracks of the TUM sequence (BASELINE config #1) carry camera poses directly (optical
frame, extrinsics = identity) and no feature estimates: those are triangulated here from the given poses.
Values of config/base7_vis_feat_only.json: reprojection sigma 2 px, Huber 1, at least 5 observations per feature.
"""
import os

import numpy as np
from scipy.spatial.transform import Rotation as Rot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")
EXT_OPTICAL = np.array([-0.5, 0.5, -0.5, 0.5, 0.0, 0.0, 0.0])      # robot <- camera, q_xyzw then t
EXT_IDENTITY = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
VIS_FEAT_ONLY = dict(reproj_sigma=2.0, reproj_huber=1.0, min_obs=5)


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def last_sighting_wins(d):
    """A frame file may list a feature id several times (the TUM tracks do); the reference stores sightings in a map keyed
    by (feature, frame, camera) and assigns (orb_output_low_level_feature_reader.cpp:184-186, :60-61), so the last line wins."""
    key = d["obs_frame"].astype(np.int64) * (int(d["obs_feature"].max()) + 1) + d["obs_feature"]
    _, first_rev = np.unique(key[::-1], return_index=True)
    keep = np.sort(len(key) - 1 - first_rev)
    out = dict(d)
    for k in ("obs_frame", "obs_feature", "obs_pixel"):
        out[k] = d[k][keep]
    return out


def pose_blocks(poses_tq):
    """`tx ty tz qx qy qz qw` rows -> the optimiser's pose blocks `[t(3), axis-angle(3)]`."""
    q = poses_tq[:, 3:7] / np.linalg.norm(poses_tq[:, 3:7], axis=1, keepdims=True)
    return np.concatenate([poses_tq[:, 0:3], Rot.from_quat(q).as_rotvec()], axis=1)


def _cam_from_world(poses, ext):
    Rw = Rot.from_rotvec(poses[:, 3:6]).as_matrix()
    Re = Rot.from_quat(ext[0:4]).as_matrix()
    Rc = np.einsum("ji,nkj->nik", Re, Rw)                            # Re^T Rw^T
    tc = -np.einsum("nij,nj->ni", Rc, poses[:, 0:3]) - Re.T @ ext[4:7]
    return Rc, tc


def triangulate_tracks(poses, ext, K, obs_frame, obs_feature, obs_pixel, min_obs, max_rms_px=20.0, depth_range=(0.2, 60.0)):
    """Linear (DLT) triangulation of every track with at least `min_obs` observations; tracks whose point lands
    outside `depth_range` in any view or re-projects worse than `max_rms_px` are dropped.  Returns (ids, xyz)."""
    Rc, tc = _cam_from_world(poses, ext)
    order = np.argsort(obs_feature, kind="stable")
    of, oi = obs_frame[order], obs_feature[order]
    u = (obs_pixel[order].astype(np.float64) - K[2:4]) / K[0:2]
    ids, start, cnt = np.unique(oi, return_index=True, return_counts=True)
    keep_ids, xyz = [], []
    for i, s, c in zip(ids, start, cnt):
        if c < min_obs:
            continue
        f = of[s:s + c]
        A = np.concatenate([np.concatenate([u[s:s + c, k:k + 1] * Rc[f, 2] - Rc[f, k], (u[s:s + c, k] * tc[f, 2] - tc[f, k])[:, None]], axis=1)
                            for k in (0, 1)])
        X = np.linalg.svd(A)[2][-1]
        if abs(X[3]) < 1e-12:
            continue
        X = X[:3] / X[3]
        pc = np.einsum("nij,j->ni", Rc[f], X) + tc[f]
        if pc[:, 2].min() < depth_range[0] or pc[:, 2].max() > depth_range[1]:
            continue
        e = (pc[:, 0:2] / pc[:, 2:3] - u[s:s + c]) * K[0:2]
        if np.sqrt((e * e).sum(1).mean()) > max_rms_px:
            continue
        keep_ids.append(int(i)); xyz.append(X)
    return np.array(keep_ids, dtype=np.int64), np.array(xyz).reshape(-1, 3)


def problem_from_dataset(d, ext=EXT_OPTICAL, sigma=VIS_FEAT_ONLY["reproj_sigma"], huber=VIS_FEAT_ONLY["reproj_huber"],
                         min_obs=2, const_poses=2, features=None):
    """Reprojection-only problem in the layout synth.upload() pushes through the C ABI.  `features` = (ids, xyz) overrides
    the data set's own feature estimates (ground truth for the simulated sets).  Features seen fewer than `min_obs`
    times are left out, as orb_output_low_level_feature_reader.cpp:66-71 (single sightings) and
    object_pose_graph_optimizer.h:826-861 (min_low_level_feature_observations) do; frames keep their order."""
    d = last_sighting_wins(d)
    poses = pose_blocks(d["poses_tq"])
    ids, xyz = features if features is not None else (d["feature_ids"].astype(np.int64), d["feature_xyz"])
    count = dict(zip(*np.unique(d["obs_feature"], return_counts=True)))
    used = np.array([count.get(int(i), 0) >= min_obs for i in ids], dtype=bool)
    ids, xyz = ids[used], xyz[used]
    index = {int(i): k for k, i in enumerate(ids)}
    sel = np.array([int(i) in index for i in d["obs_feature"]], dtype=bool)
    rp_point = np.array([index[int(i)] for i in d["obs_feature"][sel]], dtype=np.uint32)
    rp_pose = d["obs_frame"][sel].astype(np.uint32)
    rp_pixel = d["obs_pixel"][sel].astype(np.float64)
    order = np.lexsort((rp_pose, rp_point))                          # by (point, pose), the order the kernels prefer
    P = len(poses)
    prob = dict(K=np.asarray(d["K"], dtype=np.float64).reshape(1, 4), ext=np.asarray(ext, dtype=np.float64).reshape(1, 7),
                poses=poses.copy(), gt_poses=poses.copy(), pose_const=np.zeros(P, np.uint8),
                points=xyz.copy(), gt_points=xyz.copy(), point_const=np.zeros(len(xyz), np.uint8), feature_ids=ids,
                objects=np.zeros((0, 7)), object_const=np.zeros(0, np.uint8),
                rp_pose=rp_pose[order], rp_point=rp_point[order], rp_cam=np.zeros(int(sel.sum()), np.uint16),
                rp_pixel=rp_pixel[order], rp_sigma=float(sigma), rp_huber=float(huber),
                bb_obj=np.zeros(0, np.uint32), bb_pose=np.zeros(0, np.uint32))
    prob["pose_const"][:const_poses] = 1
    return prob


def tum_problem(d=None, min_obs=VIS_FEAT_ONLY["min_obs"], max_frames=None):
    """BASELINE config #1: visual-only BA over the ORB-SLAM2 tracks of TUM fr2/pioneer_360, features triangulated from the
    poses in the files.  `max_frames` keeps the first frames only (a window)."""
    d = last_sighting_wins(d if d is not None else load_fixture("tum_fr2_360_tracks"))
    if max_frames is not None:
        keep = d["obs_frame"] < max_frames
        d.update(poses_tq=d["poses_tq"][:max_frames], frame_ids=d["frame_ids"][:max_frames], obs_frame=d["obs_frame"][keep],
                 obs_feature=d["obs_feature"][keep], obs_pixel=d["obs_pixel"][keep])
    poses = pose_blocks(d["poses_tq"])
    feats = triangulate_tracks(poses, EXT_IDENTITY, d["K"], d["obs_frame"], d["obs_feature"], d["obs_pixel"], min_obs)
    return problem_from_dataset(d, ext=EXT_IDENTITY, min_obs=min_obs, const_poses=1, features=feats)
