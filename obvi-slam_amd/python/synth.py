"""Synthetic bundle-adjustment problems, as flat arrays for the C ABI.

Input generation for bench.py and the tests -- no solver code.  Patterned on the reference's own
synthetic fixture description (data/vslam_superset1/README.md: steps along a smooth path, points
within 20 m, landmark noise) with the values of config/base7a_2_fallback.json:
  camera  K = (525, 525, 319.5, 239.5), 640x480 (data/TUM_fr2_pioneer_360.../camera_matrix.txt),
          extrinsics q_xyzw = (-.5, .5, -.5, .5), t = 0 (optical z-forward on an x-forward robot)
  poses   0.2 m keyframe spacing on a closed planar loop, heading along the tangent, small
          roll/pitch so the axis-angle is generic; initial guess = integrated noisy odometry
          with per-step sigma 0.025 |delta|
  points  seen by a contiguous run of 5 + Geom(mean 5) keyframes, 1 px pixel noise, 5 % gross
          outliers (+-50 px), initial point = truth + N(0, 0.1 m)
  objects six shape classes (chair, bench, roadblock, treetrunk, lamppost, trashcan), box =
          projected ellipsoid + N(0, 30 px) per edge, covariance diag 900, >= 10 observations
The projection used to synthesise measurements is an independent numpy restatement of
vslam_math_util.h:347-394 and ellipsoid_utils.h:160-273 (it doubles as a cross-check of the
oracle in tests/test_golden.py).
RNG: numpy Generator(MT19937(20241008 + config)).
"""
import numpy as np
from scipy.spatial.transform import Rotation as Rot

K_DEFAULT = np.array([525.0, 525.0, 319.5, 239.5])
EXT_DEFAULT = np.array([-0.5, 0.5, -0.5, 0.5, 0.0, 0.0, 0.0])
IMG_W, IMG_H = 640.0, 480.0
DIM_REG = float(np.float32(1e-3))  # kDimensionRegularizationConstant is a `float` (ellipsoid_utils.h:22)

# shape priors: SURVEY 5.6 (mean dims), diagonal covariances
SHAPE_CLASSES = {
    "chair": ((0.62, 0.62, 0.975), (0.05, 0.05, 0.08)),
    "bench": ((1.0, 2.5, 1.5), (0.3, 0.6, 0.3)),
    "roadblock": ((0.29, 0.29, 0.48), (0.03, 0.03, 0.05)),
    "treetrunk": ((0.4, 0.4, 2.0), (0.1, 0.1, 1.0)),
    "lamppost": ((0.3, 0.3, 4.0), (0.1, 0.1, 1.5)),
    "trashcan": ((1.0, 1.0, 1.5), (0.15, 0.15, 0.2)),
}

# residual parameters of config/base7a_2_fallback.json
RESIDUAL_PARAMS = dict(reproj_sigma=1.5, reproj_huber=1.0, bbox_huber=0.5, shape_huber=10.0,
                       invalid_ellipsoid_error=1000.0, ltm_huber=1.0, relpose_huber=1.0,
                       odom_cov_mult=0.025, bbox_var=900.0)


# ------------------------------------------------------------------------------------------
# independent numpy restatement of the two projection functions
# ------------------------------------------------------------------------------------------
def quat_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def project_points(poses, pts, K=K_DEFAULT, ext=EXT_DEFAULT):
    """poses [n,6] (t, aa), pts [n,3] -> pixel [n,2], depth [n]  (vslam_math_util.h:347-394)."""
    R = Rot.from_rotvec(poses[:, 3:6]).as_matrix()                       # world<-robot
    pr = np.einsum("nji,nj->ni", R, pts - poses[:, 0:3])                 # R^T (X - t)
    Re = quat_to_R(ext[0:4])
    pc = (pr - ext[4:7]) @ Re                                            # R_e^T (p_r - t_e)
    u = pc[:, 0] / pc[:, 2]
    v = pc[:, 1] / pc[:, 2]
    return np.stack([K[0] * u + K[2], K[1] * v + K[3]], axis=1), pc[:, 2]


def project_ellipsoids(ell, poses, K=K_DEFAULT, ext=EXT_DEFAULT):
    """ell [n,7] (or [n,9]: the unconstrained block), poses [n,6] -> pixel corners [n,4] (minx,maxx,miny,maxy), valid [n], depth [n].
    Dual-quadric projection of ellipsoid_utils.h:160-273."""
    n = len(ell)
    R = Rot.from_rotvec(poses[:, 3:6]).as_matrix()
    Re = quat_to_R(ext[0:4])
    Rcw = np.einsum("ji,nkj->nik", Re, R)                                # R_e^T R^T
    tcw = -np.einsum("nij,nj->ni", Rcw, poses[:, 0:3]) - Re.T @ ext[4:7]
    if ell.shape[1] == 9:                                                 # unconstrained block (x y z ax ay az dx dy dz): VectorToAxisAngle, identity at or below 1e-8
        ang = np.linalg.norm(ell[:, 3:6], axis=1)
        Ro = Rot.from_rotvec(np.where((ang > 1e-8)[:, None], ell[:, 3:6], 0.0)).as_matrix()
    else:
        cy, sy = np.cos(ell[:, 3]), np.sin(ell[:, 3])
        Ro = np.zeros((n, 3, 3))
        Ro[:, 0, 0], Ro[:, 0, 1], Ro[:, 1, 0], Ro[:, 1, 1], Ro[:, 2, 2] = cy, -sy, sy, cy, 1.0
    M = np.zeros((n, 3, 4))
    M[:, :, 0:3] = Rcw @ Ro
    M[:, :, 3] = np.einsum("nij,nj->ni", Rcw, ell[:, 0:3]) + tcw
    d = np.concatenate([(ell[:, -3:] / 2.0) ** 2 + DIM_REG, -np.ones((n, 1))], axis=1)
    Q = np.einsum("nik,nk,njk->nij", M, d, M)
    xin = Q[:, 0, 2] ** 2 - Q[:, 0, 0] * Q[:, 2, 2]
    yin = Q[:, 1, 2] ** 2 - Q[:, 1, 1] * Q[:, 2, 2]
    valid = (xin > 0) & (yin > 0)
    xs, ys = np.sqrt(np.where(valid, xin, 1.0)), np.sqrt(np.where(valid, yin, 1.0))
    c = np.stack([(Q[:, 0, 2] + xs), (Q[:, 0, 2] - xs), (Q[:, 1, 2] + ys), (Q[:, 1, 2] - ys)], axis=1) / Q[:, 2, 2][:, None]
    px = np.stack([K[0] * c[:, 0] + K[2], K[0] * c[:, 1] + K[2], K[1] * c[:, 2] + K[3], K[1] * c[:, 3] + K[3]], axis=1)
    return px, valid, M[:, 2, 3]


# ------------------------------------------------------------------------------------------
def _trajectory(P, rng, spacing=0.2):
    radius = max(P * spacing / (2 * np.pi), 10.0)
    phi = np.arange(P) * spacing / radius
    pos = np.stack([radius * np.cos(phi), radius * np.sin(phi), 0.05 * np.sin(3 * phi)], axis=1)
    yaw = phi + np.pi / 2
    pitch = 0.02 * np.sin(5 * phi + 0.3)
    roll = 0.015 * np.cos(4 * phi)
    R = Rot.from_euler("ZYX", np.stack([yaw, pitch, roll], axis=1))
    return pos, R


def _noisy_odometry(pos, R, rng, mult):
    """initial guess: pose_0 = truth; pose_{i+1} = pose_i * (delta_i with noise sigma = mult*|delta|)."""
    P = len(pos)
    Rm = R.as_matrix()
    est_t = np.zeros((P, 3))
    est_R = np.zeros((P, 3, 3))
    est_t[0], est_R[0] = pos[0], Rm[0]
    dt = np.einsum("nji,nj->ni", Rm[:-1], pos[1:] - pos[:-1])
    dR = np.einsum("nji,njk->nik", Rm[:-1], Rm[1:])
    daa = Rot.from_matrix(dR).as_rotvec()
    nt = rng.normal(size=(P - 1, 3)) * (mult * np.abs(dt) + mult * np.linalg.norm(daa, axis=1)[:, None] + 1e-4)
    na = rng.normal(size=(P - 1, 3)) * (mult * np.abs(daa) + mult * np.linalg.norm(dt, axis=1)[:, None] * 0.1 + 1e-4)
    dt_n = dt + nt
    dR_n = Rot.from_rotvec(daa + na).as_matrix()
    for i in range(P - 1):
        est_t[i + 1] = est_t[i] + est_R[i] @ dt_n[i]
        est_R[i + 1] = est_R[i] @ dR_n[i]
    odom_aa = Rot.from_matrix(dR_n).as_rotvec()
    return est_t, Rot.from_matrix(est_R), dt_n, odom_aa


def odom_cov(t, aa, m_tt, m_tr, m_rt, m_rr):
    """generateOdomCov (relative_pose_factor_utils.h:17-36): diag variances, sigma floor 1e-3."""
    ang = np.linalg.norm(aa, axis=1)
    sd = np.zeros((len(t), 6))
    sd[:, 0:3] = np.abs(t) * m_tt + (np.abs(ang) * m_rt)[:, None]
    sd[:, 3:6] = np.abs(aa) * m_rr + (np.linalg.norm(t, axis=1) * m_tr)[:, None]
    var = np.maximum(sd ** 2, 1e-6)
    cov = np.zeros((len(t), 6, 6))
    idx = np.arange(6)
    cov[:, idx, idx] = var
    return cov.reshape(-1, 36)


def make_problem(P, L, O=0, seed=20241008, outlier_frac=0.05, const_poses=1, pixel_noise=1.0,
                 point_noise=0.1, with_relpose=True, min_obj_obs=10, object_classes=None, bbox_noise=30.0, stereo=False, object_seed=None, min_parallax_deg=0.0):
    """Returns a dict of flat arrays accepted by upload(); 'gt_*' hold the ground truth.
    min_parallax_deg > 0 (the "w" problems, make_well_posed): only features whose rays from the first and the last frame of their track meet at that angle or
    more are kept -- the camera looks along the direction of travel, so a feature near the optical axis 15 m ahead has no observable depth and LM walks
    it to infinity at its own pace (the 600-m features of profiles/r05_end_state_config3.txt).  0 = the generator of rounds 1-5, same random stream."""
    rng = np.random.Generator(np.random.MT19937(seed))
    rp = RESIDUAL_PARAMS
    pos, R = _trajectory(P, rng)
    gt_poses = np.concatenate([pos, R.as_rotvec()], axis=1)
    est_t, est_R, odom_t, odom_aa = _noisy_odometry(pos, R, rng, rp["odom_cov_mult"])
    poses = np.concatenate([est_t, est_R.as_rotvec()], axis=1)
    Rm = R.as_matrix()
    Re = quat_to_R(EXT_DEFAULT[0:4])

    # ---- points ---------------------------------------------------------------------------
    gt_points = np.zeros((0, 3))
    obs_pose, obs_point, obs_pix = [], [], []
    n_have = 0
    guard = 0
    while n_have < L and guard < 200:
        guard += 1
        nb = int((L - n_have) * 1.6) + 64
        run = np.minimum(5 + rng.geometric(1.0 / 6.0, size=nb) - 1, max(5, min(P, 40)))
        run = np.minimum(run, P)
        first = rng.integers(0, np.maximum(P - run + 1, 1))
        mid = np.minimum(first + run // 2, P - 1)
        depth = rng.uniform(np.maximum(2.0, 0.2 * run * 0.6 + 1.5), 20.0)
        ax = rng.uniform(-0.42, 0.42, size=nb)
        ay = rng.uniform(-0.32, 0.32, size=nb)
        pc = np.stack([np.tan(ax) * depth, np.tan(ay) * depth, depth], axis=1)
        pr = pc @ Re.T + EXT_DEFAULT[4:7]
        X = np.einsum("nij,nj->ni", Rm[mid], pr) + pos[mid]
        # all (point, frame) pairs of each run
        offs = np.arange(run.max())
        fr = first[:, None] + offs[None, :]
        ok = offs[None, :] < run[:, None]
        pi_, fi_ = np.nonzero(ok)
        frames = fr[pi_, fi_]
        pix, z = project_points(gt_poses[frames], X[pi_])
        vis = (z > 0.5) & (pix[:, 0] > 2) & (pix[:, 0] < IMG_W - 2) & (pix[:, 1] > 2) & (pix[:, 1] < IMG_H - 2)
        cnt = np.bincount(pi_[vis], minlength=nb)
        keep_pt = cnt >= 5
        if min_parallax_deg > 0.0:
            ra, rb = X - pos[first], X - pos[np.minimum(first + run - 1, P - 1)]
            cosang = (ra * rb).sum(axis=1) / np.maximum(np.linalg.norm(ra, axis=1) * np.linalg.norm(rb, axis=1), 1e-12)
            keep_pt &= np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0))) >= min_parallax_deg
        keep_pt &= np.cumsum(keep_pt) <= (L - n_have)
        new_id = np.cumsum(keep_pt) - 1 + n_have
        sel = vis & keep_pt[pi_]
        obs_pose.append(frames[sel]); obs_point.append(new_id[pi_[sel]]); obs_pix.append(pix[sel])
        gt_points = np.concatenate([gt_points, X[keep_pt]], axis=0)
        n_have = len(gt_points)
    obs_pose = np.concatenate(obs_pose).astype(np.uint32)
    obs_point = np.concatenate(obs_point).astype(np.uint32)
    obs_pix = np.concatenate(obs_pix)
    obs_cam = np.zeros(len(obs_pose), np.uint16)
    K_all, ext_all = K_DEFAULT[None, :].copy(), EXT_DEFAULT[None, :].copy()
    if stereo:
        # second camera 0.12 m to the right of the first (robot -y), same orientation and intrinsics
        ext2 = EXT_DEFAULT.copy(); ext2[5] = -0.12
        pix2, z2 = project_points(gt_poses[obs_pose], gt_points[obs_point], K_DEFAULT, ext2)
        vis2 = (z2 > 0.5) & (pix2[:, 0] > 2) & (pix2[:, 0] < IMG_W - 2) & (pix2[:, 1] > 2) & (pix2[:, 1] < IMG_H - 2)
        obs_pose = np.concatenate([obs_pose, obs_pose[vis2]]); obs_point = np.concatenate([obs_point, obs_point[vis2]])
        obs_pix = np.concatenate([obs_pix, pix2[vis2]]); obs_cam = np.concatenate([obs_cam, np.ones(int(vis2.sum()), np.uint16)])
        K_all = np.stack([K_DEFAULT, K_DEFAULT]); ext_all = np.stack([EXT_DEFAULT, ext2])
    n_r = len(obs_pose)
    obs_pix = obs_pix + rng.normal(size=(n_r, 2)) * pixel_noise
    is_out = rng.uniform(size=n_r) < outlier_frac
    obs_pix[is_out] += rng.uniform(-50, 50, size=(int(is_out.sum()), 2))
    points = gt_points + rng.normal(size=gt_points.shape) * point_noise
    # sort observations by (point, pose): CSC-by-point order, the layout the kernels prefer
    order = np.lexsort((obs_cam, obs_pose, obs_point))
    obs_pose, obs_point, obs_pix, is_out, obs_cam = obs_pose[order], obs_point[order], obs_pix[order], is_out[order], obs_cam[order]

    prob = dict(K=K_all, ext=ext_all,
                poses=poses, gt_poses=gt_poses, pose_const=np.zeros(P, np.uint8),
                points=points, gt_points=gt_points, point_const=np.zeros(len(points), np.uint8),
                rp_pose=obs_pose, rp_point=obs_point, rp_cam=obs_cam, rp_pixel=obs_pix,
                rp_sigma=rp["reproj_sigma"], rp_huber=rp["reproj_huber"], rp_is_outlier=is_out)
    prob["pose_const"][:const_poses] = 1
    import os as _os
    if _os.environ.get("OBVI_SYNTH_SORT_POINTS"):
        # experiment (round 6): feature ids in the order of first sighting, as a front end that numbers features when it first sees them hands them over
        # (the generator's ids are random with respect to the trajectory)
        firstf = np.full(len(points), P, np.int64); np.minimum.at(firstf, obs_point.astype(np.int64), obs_pose.astype(np.int64))
        rank = np.empty(len(points), np.int64); rank[np.argsort(firstf, kind="stable")] = np.arange(len(points))
        inv = np.argsort(rank)
        q_point = rank[obs_point.astype(np.int64)].astype(np.uint32)
        order2 = np.lexsort((obs_cam, obs_pose, q_point))
        prob.update(points=points[inv], gt_points=gt_points[inv], rp_point=q_point[order2], rp_pose=obs_pose[order2], rp_cam=obs_cam[order2], rp_pixel=obs_pix[order2], rp_is_outlier=is_out[order2])

    # ---- objects --------------------------------------------------------------------------
    objects = np.zeros((0, 7)); gt_objects = np.zeros((0, 7))
    bb_obj = np.zeros(0, np.uint32); bb_pose = np.zeros(0, np.uint32); bb_corners = np.zeros((0, 4))
    sp_mean = np.zeros((0, 3)); sp_cov = np.zeros((0, 9)); obj_class = []
    if O > 0:
        # object_seed: place the objects (and draw their initial estimates) from a separate stream so that several
        # windows over the same place -- different `seed`, same `object_seed` -- share one object set (config #4)
        rng_meas = rng
        if object_seed is not None:
            rng = np.random.Generator(np.random.MT19937(object_seed))
        names = list(object_classes) if object_classes else list(SHAPE_CLASSES.keys())
        cand_o, cand_corners, cand_pose, cand_gt, cand_cls = [], [], [], [], []
        tries = 0
        n_obj = 0
        while n_obj < O and tries < 50:
            tries += 1
            nb = (O - n_obj) * 3 + 8
            cls = rng.integers(0, len(names), size=nb)
            dims = np.array([SHAPE_CLASSES[names[c]][0] for c in cls])
            anchor = rng.integers(0, P, size=nb)
            # place ahead of the anchor keyframe (so it is in view from it and from the frames
            # before it), 3..15 m off the path
            ahead = rng.uniform(6.0, 18.0, size=nb)
            lateral = rng.uniform(0.15, 0.45, size=nb) * ahead * rng.choice([-1.0, 1.0], size=nb)
            hdg = np.arctan2(Rm[anchor, 1, 0], Rm[anchor, 0, 0])
            fwd = np.stack([np.cos(hdg), np.sin(hdg), np.zeros(nb)], axis=1)
            left = np.stack([-np.sin(hdg), np.cos(hdg), np.zeros(nb)], axis=1)
            centre = pos[anchor] + fwd * ahead[:, None] + left * lateral[:, None]
            centre[:, 2] = dims[:, 2] / 2.0 - 0.3
            yaw = rng.uniform(-np.pi, np.pi, size=nb)
            ell = np.concatenate([centre, yaw[:, None], dims], axis=1)
            span = min(P, 400)
            offs = np.arange(-span, 40)
            fr = np.clip(anchor[:, None] + offs[None, :], 0, P - 1)
            oi, fi = np.nonzero(np.ones_like(fr, dtype=bool))
            frames = fr[oi, fi]
            px, valid, depth = project_ellipsoids(ell[oi], gt_poses[frames])
            inimg = valid & (depth > 1.5) & (depth < 30.0) & (px.min(axis=1) > 5) & (px[:, 0:2].max(axis=1) < IMG_W - 5) & (px[:, 2:4].max(axis=1) < IMG_H - 5)
            # de-duplicate clipped frames
            key = oi.astype(np.int64) * P + frames
            _, uniq_idx = np.unique(key, return_index=True)
            m = np.zeros(len(key), bool); m[uniq_idx] = True
            inimg &= m
            cnt = np.bincount(oi[inimg], minlength=nb)
            keep = cnt >= min_obj_obs
            keep &= np.cumsum(keep) <= (O - n_obj)
            new_id = np.cumsum(keep) - 1 + n_obj
            sel = inimg & keep[oi]
            cand_o.append(new_id[oi[sel]]); cand_pose.append(frames[sel]); cand_corners.append(px[sel])
            cand_gt.append(ell[keep]); cand_cls.append(cls[keep])
            n_obj += int(keep.sum())
        if n_obj == 0:
            raise ValueError("no object with >= %d in-image observations could be placed" % min_obj_obs)
        gt_objects = np.concatenate(cand_gt, axis=0)
        cls_all = np.concatenate(cand_cls)
        bb_obj = np.concatenate(cand_o).astype(np.uint32)
        bb_pose = np.concatenate(cand_pose).astype(np.uint32)
        bb_corners = np.concatenate(cand_corners)
        # the functor's corner order is (min_x, max_x, min_y, max_y) in pixels
        bb_corners = np.stack([bb_corners[:, 0:2].min(axis=1), bb_corners[:, 0:2].max(axis=1),
                               bb_corners[:, 2:4].min(axis=1), bb_corners[:, 2:4].max(axis=1)], axis=1)
        bb_corners = bb_corners + rng_meas.normal(size=bb_corners.shape) * bbox_noise
        order = np.lexsort((bb_pose, bb_obj))
        bb_obj, bb_pose, bb_corners = bb_obj[order], bb_pose[order], bb_corners[order]
        objects = gt_objects.copy()
        objects[:, 0:3] += rng.normal(size=(len(objects), 3)) * 0.3
        objects[:, 3] += rng.normal(size=len(objects)) * 0.2
        objects[:, 4:7] *= 1.0 + rng.normal(size=(len(objects), 3)) * 0.1
        sp_mean = np.array([SHAPE_CLASSES[names[c]][0] for c in cls_all]).reshape(-1, 3)
        sp_cov = np.zeros((len(objects), 3, 3))
        sd = np.array([SHAPE_CLASSES[names[c]][1] for c in cls_all]).reshape(-1, 3)
        sp_cov[:, np.arange(3), np.arange(3)] = sd ** 2
        sp_cov = sp_cov.reshape(-1, 9)
        obj_class = [names[c] for c in cls_all]
    n_b = len(bb_obj)
    bb_cov = np.zeros((n_b, 4, 4))
    bb_cov[:, np.arange(4), np.arange(4)] = rp["bbox_var"]
    prob.update(objects=objects, gt_objects=gt_objects, object_const=np.zeros(len(objects), np.uint8),
                bb_obj=bb_obj, bb_pose=bb_pose, bb_cam=np.zeros(n_b, np.uint16), bb_corners=bb_corners,
                bb_cov=bb_cov.reshape(-1, 16), bb_huber=rp["bbox_huber"], bb_invalid=rp["invalid_ellipsoid_error"],
                sp_obj=np.arange(len(objects), dtype=np.uint32), sp_mean=sp_mean, sp_cov=sp_cov, sp_huber=rp["shape_huber"], obj_class=obj_class)

    # ---- consecutive-frame odometry factors ----------------------------------------------
    if with_relpose and P > 1:
        m = rp["odom_cov_mult"]
        prob.update(rl_a=np.arange(P - 1, dtype=np.uint32), rl_b=np.arange(1, P, dtype=np.uint32),
                    rl_t=odom_t, rl_aa=odom_aa, rl_cov=odom_cov(odom_t, odom_aa, m, m, m, m), rl_huber=rp["relpose_huber"])
    return prob


def make_well_posed(prob, min_depth=0.5):
    """The "w" variant of a problem (round 6, VERDICT r5 item 2: an end state that two fp64 implementations can be held to).  Same structure, same measurements;
    what changes is the START, the way a SLAM front end would have produced it: a feature / an object is initialised relative to the ESTIMATED pose of the
    frame it is first anchored to (middle frame of a feature's track, first observing frame of an object) instead of at ground truth in a world frame the
    drifted trajectory has long left -- so no sighting starts at near-zero or negative depth (`min_depth` metres in front of EVERY observing camera, pushed
    along the ray of its anchor camera otherwise) -- and the caller uploads the odometry factors of all consecutive frames (upload(relpose=True)) with
    `const_poses` >= 1 poses constant, which fixes the scale gauge everywhere along the trajectory.  Deterministic: no random draw."""
    q = dict(prob)
    P = len(prob["poses"])
    Rg = Rot.from_rotvec(prob["gt_poses"][:, 3:6]).as_matrix(); tg = prob["gt_poses"][:, :3]
    Re_ = Rot.from_rotvec(prob["poses"][:, 3:6]).as_matrix(); te = prob["poses"][:, :3]
    # ---- features: anchor = the middle observation of the track (the observations are sorted by (point, pose))
    L = len(prob["points"])
    rp_point, rp_pose = prob["rp_point"].astype(np.int64), prob["rp_pose"].astype(np.int64)
    ptr = np.searchsorted(rp_point, np.arange(L + 1))
    has = ptr[1:] > ptr[:-1]
    mid = np.where(has, rp_pose[np.minimum((ptr[:-1] + ptr[1:]) // 2, len(rp_pose) - 1)], 0)
    Xg = prob["points"]                                     # ground truth + the generator's noise, in the ground-truth world
    local = np.einsum("nji,nj->ni", Rg[mid], Xg - tg[mid])  # ... as seen from the anchor's true pose
    X = np.einsum("nij,nj->ni", Re_[mid], local) + te[mid]  # ... re-attached to the anchor's estimated pose
    X = np.where(has[:, None], X, Xg)
    for _ in range(8):                                      # push along the anchor ray until every observing camera sees it >= min_depth ahead
        _, z = project_points(prob["poses"][rp_pose], X[rp_point], prob["K"][0], prob["ext"][0])
        zmin = np.full(L, np.inf); np.minimum.at(zmin, rp_point, z)
        bad = has & (zmin < min_depth)
        if not bad.any():
            break
        ray = X[bad] - te[mid[bad]]
        X[bad] = X[bad] + ray / np.maximum(np.linalg.norm(ray, axis=1, keepdims=True), 1e-9) * (min_depth - zmin[bad] + 0.25)[:, None]
    q["points"] = X
    # ---- objects: anchor = the first observing frame
    O = len(prob["objects"])
    if O:
        first = np.full(O, P, np.int64); np.minimum.at(first, prob["bb_obj"].astype(np.int64), prob["bb_pose"].astype(np.int64))
        first = np.where(first >= P, 0, first)
        obj = prob["objects"].copy()
        c_local = np.einsum("nji,nj->ni", Rg[first], obj[:, :3] - tg[first])
        obj[:, :3] = np.einsum("nij,nj->ni", Re_[first], c_local) + te[first]
        # yaw rides along with the heading error of the anchor (rotation about z of R_est R_gt^T)
        dR = np.einsum("nij,nkj->nik", Re_[first], Rg[first])
        obj[:, 3] += np.arctan2(dR[:, 1, 0], dR[:, 0, 0])
        q["objects"] = obj
    return q


def config3w(P=2000, L=300000, O=200, seed=20241008 + 3):
    """"Config 3w": BASELINE config #3's sizes as a well-posed problem (tests/golden/gen_config3w_end_state.py has the recipe in words).
    Upload with the odometry factors (upload(relpose=True), the default)."""
    return make_well_posed(make_problem(P=P, L=L, O=O, seed=seed, const_poses=5, min_obj_obs=10, object_classes=("bench",), min_parallax_deg=3.0, stereo=True))


def nine_dof(prob, tilt=0.0, seed=0):
    """The same problem with the 9-parameter ellipsoid block (x y z ax ay az dx dy dz) of vslam_obj_opt_types_refactor.h:15-21: the objects' rotation becomes
    the rotation vector (0, 0, yaw) composed with a random tilt of up to `tilt` radians about a horizontal axis (tilt = 0: exactly the yaw-only objects, for which a
    9-block handle must reproduce a 7-block handle's numbers); the boxes are NOT re-projected (they are measurements).  LTM priors, if any, get the matching 9-vector
    mean and a 9x9 covariance (the yaw's variance on all three rotation entries, cross terms to the other parameters kept for az)."""
    q = dict(prob)
    rng = np.random.Generator(np.random.MT19937(seed))

    def nine(obj):
        n = len(obj)
        aa = np.zeros((n, 3)); aa[:, 2] = obj[:, 3]
        if tilt > 0.0 and n:
            phi = rng.uniform(0, 2 * np.pi, n); mag = rng.uniform(0.2 * tilt, tilt, n)
            t = Rot.from_rotvec(np.stack([mag * np.cos(phi), mag * np.sin(phi), np.zeros(n)], axis=1))
            aa = (Rot.from_rotvec(aa) * t).as_rotvec()
        return np.concatenate([obj[:, :3], aa, obj[:, 4:7]], axis=1)
    q["objects"] = nine(prob["objects"])
    if "gt_objects" in prob:
        q["gt_objects"] = nine(prob["gt_objects"])
    if "lt_obj" in prob and len(prob["lt_obj"]):
        idx = [0, 1, 2, 5, 6, 7, 8]                          # where the 7-block's entries (x y z yaw dx dy dz) sit in the 9-block
        m7, c7 = np.asarray(prob["lt_mean"]).reshape(-1, 7), np.asarray(prob["lt_cov"]).reshape(-1, 7, 7)
        m9 = np.zeros((len(m7), 9)); m9[:, idx] = m7
        c9 = np.zeros((len(m7), 9, 9))
        for a, ia in enumerate(idx):
            for b, ib in enumerate(idx):
                c9[:, ia, ib] = c7[:, a, b]
        c9[:, 3, 3] = c7[:, 3, 3]; c9[:, 4, 4] = c7[:, 3, 3]
        q["lt_mean"], q["lt_cov"] = m9, c9.reshape(-1, 81)
    q["object_block_size"] = 9
    return q


def upload(ba, prob, relpose=True, objects=True, reproj=True):
    """Push a flat problem through the C ABI (works for any object with the BundleAdjuster API)."""
    ba.set_cameras(prob["K"], prob["ext"])
    ba.set_poses(prob["poses"], prob["pose_const"])
    ba.set_points(prob["points"], prob["point_const"])
    ba.set_objects(prob["objects"], prob["object_const"])
    if reproj:
        ba.set_reproj(prob["rp_pose"], prob["rp_point"], prob["rp_cam"], prob["rp_pixel"], prob["rp_sigma"], prob["rp_huber"])
    if objects and len(prob["objects"]):
        ba.set_bbox(prob["bb_obj"], prob["bb_pose"], prob["bb_cam"], prob["bb_corners"], prob["bb_cov"], prob["bb_huber"], prob["bb_invalid"])
        ba.set_shape_priors(prob["sp_obj"], prob["sp_mean"], prob["sp_cov"], prob["sp_huber"])
        if "lt_obj" in prob:
            ba.set_ltm_priors(prob["lt_obj"], prob["lt_mean"], prob["lt_cov"], prob["lt_huber"])
    if relpose and "rl_a" in prob:
        ba.set_relpose(prob["rl_a"], prob["rl_b"], prob["rl_t"], prob["rl_aa"], prob["rl_cov"], prob["rl_huber"])


def problem_stats(prob):
    return dict(P=len(prob["poses"]), L=len(prob["points"]), O=len(prob["objects"]), N_r=len(prob["rp_pose"]),
                N_b=len(prob["bb_obj"]), N_rel=len(prob.get("rl_a", [])))


def dump_flat(prob, path, max_it=3, nonmono=True, ftol=0.0, gtol=0.0, ptol=0.0, radius=100.0, max_radius=1e4):
    """The flat problem + solver parameters as one little-endian file for the optional Ceres harness
    (oracle/ceres_harness/ceres_harness.cpp): magic, then arrays as (u64 count, payload) and scalars as f64, in upload() order."""
    import struct

    def arr(f, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype).ravel()
        f.write(struct.pack("<Q", a.size)); f.write(a.tobytes())

    def scal(f, v):
        f.write(struct.pack("<d", float(v)))
    e = lambda k, shape: prob[k] if k in prob else np.zeros(shape)   # noqa: E731
    with open(path, "wb") as f:
        f.write(b"OBVIFLT1")
        arr(f, prob["K"], "<f8"); arr(f, prob["ext"], "<f8")
        arr(f, prob["poses"], "<f8"); arr(f, prob["pose_const"], "u1"); arr(f, prob["points"], "<f8"); arr(f, prob["point_const"], "u1")
        arr(f, prob["objects"], "<f8"); arr(f, prob["object_const"], "u1")
        arr(f, prob["rp_pose"], "<u4"); arr(f, prob["rp_point"], "<u4"); arr(f, prob["rp_cam"], "<u4"); arr(f, prob["rp_pixel"], "<f8")
        scal(f, prob["rp_sigma"]); scal(f, prob["rp_huber"])
        arr(f, prob["bb_obj"], "<u4"); arr(f, prob["bb_pose"], "<u4"); arr(f, prob["bb_cam"], "<u4"); arr(f, prob["bb_corners"], "<f8"); arr(f, prob["bb_cov"], "<f8")
        scal(f, prob["bb_huber"]); scal(f, prob["bb_invalid"])
        arr(f, prob["sp_obj"], "<u4"); arr(f, prob["sp_mean"], "<f8"); arr(f, prob["sp_cov"], "<f8"); scal(f, prob["sp_huber"])
        arr(f, e("lt_obj", 0), "<u4"); arr(f, e("lt_mean", (0, 7)), "<f8"); arr(f, e("lt_cov", (0, 49)), "<f8"); scal(f, prob.get("lt_huber", 1.0))
        arr(f, e("rl_a", 0), "<u4"); arr(f, e("rl_b", 0), "<u4"); arr(f, e("rl_t", (0, 3)), "<f8"); arr(f, e("rl_aa", (0, 3)), "<f8"); arr(f, e("rl_cov", (0, 36)), "<f8")
        scal(f, prob.get("rl_huber", 1.0))
        for v in (max_it, int(nonmono), ftol, gtol, ptol, radius, max_radius):
            scal(f, v)


def make_sessions(n_sessions, P, L, O, seed0, object_seed, **kw):
    """BASELINE configs[4] (config #5), concurrent form: `n_sessions` sessions over one place -- own trajectory noise, features and
    sightings each (seed0 + s), ONE object map (object_seed: same objects, same initial estimates in every session).  The object-only
    factors (shape priors) of the shared map are carried by session 0 alone, as obvi_ba_set_shared_objects asks."""
    out = []
    for s in range(n_sessions):
        q = make_problem(P=P, L=L, O=O, seed=seed0 + s, object_seed=object_seed, **kw)
        if s != 0:
            for k in ("sp_obj", "sp_mean", "sp_cov"):
                q[k] = q[k][:0]
        out.append(q)
    for q in out[1:]:
        assert np.array_equal(q["objects"], out[0]["objects"]), "sessions of one map must start from the same object estimates"
    return out


def join_problems(probs):
    """The joint problem of sessions that share one object map: poses, features and every factor of every session in ONE problem (pose and
    feature indices shifted session by session, objects unchanged) -- what the sharded solve of the sessions must equal."""
    j = dict(probs[0])
    p_off = np.cumsum([0] + [len(q["poses"]) for q in probs])
    l_off = np.cumsum([0] + [len(q["points"]) for q in probs])

    def cat(key, offs=None, dtype=None):
        parts = []
        for s, q in enumerate(probs):
            a = np.asarray(q[key])
            parts.append(a if offs is None else (a.astype(np.int64) + offs[s]))
        a = np.concatenate(parts, axis=0)
        return a.astype(dtype) if dtype is not None else a
    j.update(poses=cat("poses"), pose_const=cat("pose_const"), points=cat("points"), point_const=cat("point_const"),
             rp_pose=cat("rp_pose", p_off, np.uint32), rp_point=cat("rp_point", l_off, np.uint32), rp_cam=cat("rp_cam"), rp_pixel=cat("rp_pixel"),
             bb_obj=cat("bb_obj"), bb_pose=cat("bb_pose", p_off, np.uint32), bb_cam=cat("bb_cam"), bb_corners=cat("bb_corners"), bb_cov=cat("bb_cov"),
             sp_obj=cat("sp_obj"), sp_mean=cat("sp_mean"), sp_cov=cat("sp_cov"))
    if np.ndim(probs[0]["rp_sigma"]) > 0:
        j["rp_sigma"] = cat("rp_sigma")
    for k in ("gt_poses", "gt_points", "rp_is_outlier"):
        if k in probs[0]:
            j[k] = cat(k)
    if "rl_a" in probs[0]:
        j.update(rl_a=cat("rl_a", p_off, np.uint32), rl_b=cat("rl_b", p_off, np.uint32), rl_t=cat("rl_t"), rl_aa=cat("rl_aa"), rl_cov=cat("rl_cov"))
    if any("lt_obj" in q for q in probs):
        have = [q for q in probs if "lt_obj" in q]
        j.update(lt_obj=np.concatenate([q["lt_obj"] for q in have]), lt_mean=np.concatenate([q["lt_mean"] for q in have]),
                 lt_cov=np.concatenate([q["lt_cov"] for q in have]), lt_huber=have[0]["lt_huber"])
    j["session_pose_offsets"], j["session_point_offsets"] = p_off, l_off
    return j
