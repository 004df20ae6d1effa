"""Visual-feature front-end gating (include/obvi_frontend.h): the epipolar-consistency votes and the
minimum-parallax test of visual_feature_front_end.h.  CPU: the oracle's restatement against an independent numpy statement
(fundamental-matrix line distance; scipy rotations).  GPU: the HIP kernels against the oracle on the same batches."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

import helpers
import obvi_ba
import synth


def _affine(pose):
    T = np.eye(4); T[:3, :3] = Rot.from_rotvec(pose[3:6]).as_matrix(); T[:3, 3] = pose[:3]
    return T


def _cam_to_robot(ext):
    T = np.eye(4); T[:3, :3] = synth.quat_to_R(ext[:4]); T[:3, 3] = ext[4:7]
    return T


def _Kmat(k):
    return np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1.0]])


def numpy_epipolar_error(K1, K2, ext1, ext2, px1, px2, pose1, pose2):
    """distance-vector from px2 to the epipolar line of px1 in image 2, through the fundamental matrix (independent of the
    reference's construction, which walks along the line through the epipole and the transferred point)"""
    c1_c2 = np.linalg.inv(_affine(pose2) @ _cam_to_robot(ext2)) @ _affine(pose1) @ _cam_to_robot(ext1)
    R, t = c1_c2[:3, :3], c1_c2[:3, 3]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(_Kmat(K2)).T @ tx @ R @ np.linalg.inv(_Kmat(K1))
    l = F @ np.array([px1[0], px1[1], 1.0])
    d = (l[0] * px2[0] + l[1] * px2[1] + l[2]) / (l[0] ** 2 + l[1] ** 2)
    return -d * l[:2]                      # foot of the perpendicular minus the point


def make_batch(seed, n_cand=300, stereo=True, same_frame_other_camera=True):
    """candidates = observations of a frame; references = the same feature in the previous frames (both cameras)"""
    prob = synth.make_problem(P=40, L=600, O=0, seed=seed, stereo=stereo, outlier_frac=0.15)
    rng = np.random.default_rng(seed)
    order = np.lexsort((prob["rp_cam"], prob["rp_pose"], prob["rp_point"]))
    pose, point, cam, pix = prob["rp_pose"][order], prob["rp_point"][order], prob["rp_cam"][order], prob["rp_pixel"][order]
    cand = rng.choice(np.flatnonzero(pose >= 6), n_cand, replace=False)
    cp, cc, cx, ptr, rp, rc, rx, rf, rs = [], [], [], [0], [], [], [], [], []
    for i in cand:
        same = np.flatnonzero((point == point[i]) & (pose > pose[i] - 5) & (pose <= pose[i]))     # frame > candidate - 5 (:502-507), candidate's own frame included
        if not same_frame_other_camera:
            # the other camera of a rectified pair in the SAME frame has its epipole at infinity (pure sideways baseline, z = 0 up to
            # round-off): the reference's construction divides by that z, so the vector is inf/NaN or round-off noise -- not comparable
            same = same[(pose[same] < pose[i]) | (cam[same] == cam[i])]
        same = same[np.lexsort((cam[same], pose[same]))]
        cp.append(pose[i]); cc.append(cam[i]); cx.append(pix[i])
        for k in same:
            rp.append(pose[k]); rc.append(cam[k]); rx.append(pix[k]); rf.append(pose[k]); rs.append(int(pose[k] == pose[i] and cam[k] == cam[i]))
        ptr.append(len(rp))
    return prob, dict(cand_pose=np.array(cp), cand_cam=np.array(cc), cand_pixel=np.array(cx), ref_ptr=np.array(ptr), ref_pose=np.array(rp), ref_cam=np.array(rc),
                      ref_pixel=np.array(rx).reshape(-1, 2), ref_frame=np.array(rf), ref_skip=np.array(rs))


def test_epipolar_error_vector_against_the_fundamental_matrix():
    prob, b = make_batch(3)
    o = helpers.oracle_ba()
    err = o.epipolar_errors(prob["K"], prob["ext"], prob["poses"], b["cand_pose"], b["cand_cam"], b["cand_pixel"], b["ref_ptr"], b["ref_pose"], b["ref_cam"], b["ref_pixel"])
    checked = 0
    for i in range(len(b["cand_pose"])):
        for k in range(b["ref_ptr"][i], b["ref_ptr"][i + 1]):
            if b["ref_pose"][k] == b["cand_pose"][i]:
                continue                                        # same frame: stereo baseline only or identical view; covered by the GPU == oracle test
            want = numpy_epipolar_error(prob["K"][b["ref_cam"][k]], prob["K"][b["cand_cam"][i]], prob["ext"][b["ref_cam"][k]], prob["ext"][b["cand_cam"][i]],
                                        b["ref_pixel"][k], b["cand_pixel"][i], prob["poses"][b["ref_pose"][k]], prob["poses"][b["cand_pose"][i]])
            assert np.abs(err[k] - want).max() <= 1e-7 * max(1.0, np.abs(want).max())
            checked += 1
    assert checked > 500
    # a noise-free correspondence lies on its epipolar line: projections of ground-truth points from ground-truth poses
    gt = dict(prob); gt["poses"] = prob["gt_poses"]
    px_a, _ = synth.project_points(prob["gt_poses"][[3] * 50], prob["gt_points"][:50], prob["K"][0], prob["ext"][0])
    px_b, _ = synth.project_points(prob["gt_poses"][[7] * 50], prob["gt_points"][:50], prob["K"][0], prob["ext"][0])
    e0 = o.epipolar_errors(prob["K"], prob["ext"], prob["gt_poses"], np.full(50, 7), np.zeros(50), px_b, np.arange(51), np.full(50, 3), np.zeros(50), px_a)
    assert np.abs(e0).max() < 1e-8


def test_vote_counting_rules():
    """:511-602 on a hand-made batch: early return looks at the earliest reference frame only; a reference that shouldBeTheSame as the
    candidate does not vote; no voters -> 0/0 -> not an inlier; strict '>' against the majority percentage."""
    prob, b = make_batch(5, n_cand=50)
    o = helpers.oracle_ba()
    args = (prob["K"], prob["ext"], prob["poses"], b["cand_pose"], b["cand_cam"], b["cand_pixel"], b["ref_ptr"], b["ref_pose"], b["ref_cam"], b["ref_pixel"], b["ref_frame"], b["ref_skip"])
    err = o.epipolar_errors(*args[:10])
    norm = np.hypot(err[:, 0], err[:, 1])
    for early in (True, False):
        for thresh, maj in ((8.0, 0.5), (1.0, 0.5), (8.0, 0.99), (3.0, 0.0)):
            votes, voters, inl = o.epipolar_votes(*args, params=obvi_ba.EpipolarParams(thresh, maj, early))
            for i in range(len(votes)):
                ks = np.arange(b["ref_ptr"][i], b["ref_ptr"][i + 1])
                if early and len(ks):
                    ks = ks[b["ref_frame"][ks] == b["ref_frame"][ks[0]]]
                ks = ks[b["ref_skip"][ks] == 0]
                assert voters[i] == len(ks) and votes[i] == int((norm[ks] < thresh).sum())
                assert inl[i] == int(len(ks) > 0 and votes[i] / len(ks) > maj)
    # a candidate whose only reference is itself: no voters
    v, n, inl = o.epipolar_votes(prob["K"], prob["ext"], prob["poses"], [10], [0], [[100.0, 100.0]], [0, 1], [10], [0], [[100.0, 100.0]], [10], [1])
    assert (v[0], n[0], inl[0]) == (0, 0, 0)


def _parallax_batch(seed, n_feat=200):
    rng = np.random.default_rng(seed)
    frame_ptr, has_pose, poses, obs_ptr, pix = [0], [], [], [0], []
    for f in range(n_feat):
        nfr = int(rng.integers(0, 6))
        base = rng.normal(size=6) * np.array([2, 2, 0.2, 0.05, 0.05, 1.0])
        p0 = rng.uniform(50, 400, size=2)
        step_t, step_r, step_px = rng.choice([0.0, 0.02, 0.3]), rng.choice([0.0, 0.01, 0.2]), rng.choice([0.0, 1.0, 4.99, 5.0, 12.0])
        for k in range(nfr):
            has_pose.append(int(rng.uniform() > 0.15))
            p = base.copy(); p[0] += k * step_t; p[5] += k * step_r
            poses.append(p)
            for c in range(int(rng.integers(1, 3))):
                pix.append(p0 + np.array([k * step_px, 0.0]) + (0.0 if c == 0 else 40.0))
            obs_ptr.append(len(pix))
        frame_ptr.append(len(has_pose))
    return np.array(frame_ptr), np.array(has_pose), np.array(poses).reshape(-1, 6), np.array(obs_ptr), np.array(pix).reshape(-1, 2)


def numpy_parallax(frame_ptr, has_pose, poses, obs_ptr, pix, prm):
    out = []
    for f in range(len(frame_ptr) - 1):
        ks = range(frame_ptr[f], frame_ptr[f + 1])
        ok = False
        for a, i in enumerate(ks):
            for j in list(ks)[a + 1:]:
                pose_req = pixel_req = False
                if prm.enforce_pose and has_pose[i] and has_pose[j]:
                    rel = np.linalg.inv(_affine(poses[i])) @ _affine(poses[j])
                    ang = np.linalg.norm(Rot.from_matrix(rel[:3, :3]).as_rotvec())
                    pose_req = np.linalg.norm(rel[:3, 3]) >= prm.min_transl or ang >= prm.min_orient
                if prm.enforce_pixel:
                    pa, pb = pix[obs_ptr[i]:obs_ptr[i + 1]], pix[obs_ptr[j]:obs_ptr[j + 1]]
                    pixel_req = bool((np.linalg.norm(pa[:, None, :] - pb[None, :, :], axis=2) >= prm.min_pixel).any())
                req = pose_req if (prm.enforce_pose and not prm.enforce_pixel) else pixel_req if (prm.enforce_pixel and not prm.enforce_pose) else (pose_req and pixel_req) if prm.enforce_pose else True
                ok = ok or req
        out.append(int(ok))
    return np.array(out, np.uint8)


@pytest.mark.parametrize("enforce_pixel,enforce_pose", [(True, False), (False, True), (True, True), (False, False)])
def test_parallax_test_against_numpy(enforce_pixel, enforce_pose):
    batch = _parallax_batch(11)
    prm = obvi_ba.ParallaxParams(5.0, 0.1, 0.05, enforce_pixel, enforce_pose)
    got = helpers.oracle_ba().parallax(*batch, params=prm)
    assert np.array_equal(got, numpy_parallax(*batch, prm)) and 0 < got.sum() < len(got)


@pytest.mark.gpu
def test_gating_kernels_match_the_oracle():
    g, o = helpers.product_ba(), helpers.oracle_ba()
    for seed, stereo in ((3, True), (8, False)):
        prob, b = make_batch(seed, n_cand=2000, stereo=stereo, same_frame_other_camera=False)
        args = (prob["K"], prob["ext"], prob["poses"], b["cand_pose"], b["cand_cam"], b["cand_pixel"], b["ref_ptr"], b["ref_pose"], b["ref_cam"], b["ref_pixel"])
        eg, eo = g.epipolar_errors(*args), o.epipolar_errors(*args)
        # (a reference that IS the candidate has a zero baseline: its epipole is 0/0 or round-off/round-off; it never votes, :551-553)
        real = b["ref_skip"] == 0
        assert np.isfinite(eo[real]).all() and np.isfinite(eg[real]).all()
        assert np.abs(eg[real] - eo[real]).max() <= 1e-9 * max(1.0, np.abs(eo[real]).max())
        for early in (True, False):
            for thresh in (8.0, 2.0):
                prm = obvi_ba.EpipolarParams(thresh, 0.5, early)
                rg, ro = g.epipolar_votes(*args, b["ref_frame"], b["ref_skip"], params=prm), o.epipolar_votes(*args, b["ref_frame"], b["ref_skip"], params=prm)
                # a vote may differ only where an error norm sits within round-off of the threshold
                norm = np.where(real, np.hypot(eo[:, 0], eo[:, 1]), np.inf)
                near = np.zeros(len(rg[0]), bool)
                for i in range(len(near)):
                    near[i] = (np.abs(norm[b["ref_ptr"][i]:b["ref_ptr"][i + 1]] - thresh) < 1e-9).any()
                for a_, b_ in zip(rg, ro):
                    assert np.array_equal(a_[~near], b_[~near])
                assert 0.3 < ro[2].mean() < 1.0                # the 15 % gross outliers (and their victims) are voted out, the rest stays
    for ep, eo_ in ((True, False), (False, True), (True, True), (False, False)):
        batch = _parallax_batch(17, n_feat=3000)
        prm = obvi_ba.ParallaxParams(5.0, 0.1, 0.05, ep, eo_)
        assert np.array_equal(g.parallax(*batch, params=prm), o.parallax(*batch, params=prm))
    # argument checking: indices out of range are refused (-4), nothing is launched
    prob, b = make_batch(3, n_cand=10)
    with pytest.raises(obvi_ba.ObviError, match="status -4"):
        g.epipolar_votes(prob["K"], prob["ext"], prob["poses"][:5], b["cand_pose"], b["cand_cam"], b["cand_pixel"], b["ref_ptr"], b["ref_pose"], b["ref_cam"], b["ref_pixel"], b["ref_frame"], b["ref_skip"])
