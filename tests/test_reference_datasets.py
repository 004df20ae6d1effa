"""CPU: the oracle against the known answers the reference's own data sets hold.

data/vslam_set2, 4, 5, 6, 7 are simulated sequences whose pixels are the exact projections of the ground-truth features
from the ground-truth frame poses (data/vslam_set2/README.md), written with 6 decimals.  The restated reprojection model
-- quaternion -> axis-angle pose block, robot <- camera extrinsics, rectified pixel, multiplier f / sigma -- must
therefore give a zero residual for every observation at ground truth, and bundle adjustment started away from ground
truth must come back to it.  Fixtures: tests/golden/*.npz, packed by tests/golden/gen_dataset_fixtures.py.
"""
import numpy as np
import pytest

import dataset_io
import helpers
import synth

# worst projection error the 6-decimal text of each set allows (pixels); set 6 turns the camera, its quaternions' rounding shows
GT_PIXEL_TOL = {"vslam_set2": 5e-5, "vslam_set4": 3e-4, "vslam_set5": 4e-4, "vslam_set6": 2e-3, "vslam_set7": 2e-4}


@pytest.mark.parametrize("name", sorted(GT_PIXEL_TOL))
def test_residual_vanishes_at_the_data_sets_ground_truth(name):
    d = dataset_io.load_fixture(name)
    prob = dataset_io.problem_from_dataset(d, min_obs=1)
    assert len(prob["rp_pose"]) == len(d["obs_frame"])                 # every observation has a ground-truth feature
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    cost, res, sq = ba.evaluate(False, True)
    pixel_err = np.abs(res) * prob["rp_sigma"]                         # r = (f / sigma) (u - u_obs)  ->  pixels
    assert pixel_err.max() < GT_PIXEL_TOL[name], pixel_err.max()
    assert cost < 0.5 * len(res) * (GT_PIXEL_TOL[name] / prob["rp_sigma"]) ** 2


def perturbed(prob, seed, pose_sigma=(0.05, 0.01), point_sigma=0.2):
    rng = np.random.Generator(np.random.MT19937(seed))
    out = dict(prob)
    free = prob["pose_const"] == 0
    poses = prob["poses"].copy()
    poses[free, 0:3] += rng.normal(size=(int(free.sum()), 3)) * pose_sigma[0]
    poses[free, 3:6] += rng.normal(size=(int(free.sum()), 3)) * pose_sigma[1]
    out["poses"] = poses
    out["points"] = prob["points"] + rng.normal(size=prob["points"].shape) * point_sigma
    return out


# how far the minimum may sit from ground truth (m / rad): the rounding of the text, amplified by the 0.5 m baseline of the
# two fixed poses that sets the scale (set 6 rounds its quaternions as well)
GT_POSE_TOL = {"vslam_set2": 1e-3, "vslam_set4": 1e-4, "vslam_set5": 1e-2, "vslam_set6": 1e-2, "vslam_set7": 1e-6}


@pytest.mark.parametrize("name", sorted(GT_POSE_TOL))
def test_bundle_adjustment_returns_to_ground_truth(name):
    """Two poses fixed (gauge and scale); everything else perturbed; noiseless pixels -> the minimum is ground truth."""
    d = dataset_io.load_fixture(name)
    gt = dataset_io.problem_from_dataset(d, min_obs=3, const_poses=2)
    prob = perturbed(gt, seed=7)
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=60, ftol=1e-14, gtol=1e-14, ptol=1e-14))
    assert s.is_solution_usable
    assert s.initial_cost > 1.0 and s.final_cost < 1e-4 * s.initial_cost, (s.initial_cost, s.final_cost)
    poses = ba.get_poses()
    assert np.abs(poses - gt["poses"]).max() < GT_POSE_TOL[name], np.abs(poses - gt["poses"]).max()
    # well-observed features come back too (a feature seen under a few degrees of parallax keeps a depth error)
    seen = np.bincount(prob["rp_point"], minlength=len(prob["points"]))
    err = np.linalg.norm(ba.get_points() - gt["points"], axis=1)
    well = seen >= min(8, len(prob["poses"]))
    assert well.any() and np.median(err[well]) < 10 * GT_POSE_TOL[name] + 1e-3, np.median(err[well])


def test_tum_tracks_plumbing_config_1():
    """BASELINE config #1 on the CPU oracle: ORB-SLAM2 tracks of TUM fr2/pioneer_360 with the values of
    config/base7_vis_feat_only.json (sigma 2 px, Huber 1, >= 5 observations), features triangulated from the file poses."""
    d = dataset_io.load_fixture("tum_fr2_360_tracks")
    assert len(d["frame_ids"]) == 380 and len(d["obs_frame"]) == 78251 and len(np.unique(d["obs_feature"])) == 7419
    prob = dataset_io.tum_problem(d, max_frames=120)
    stats = synth.problem_stats(prob)
    assert stats["P"] == 120 and stats["L"] > 200 and stats["N_r"] > 5 * stats["L"] * 0.99
    assert np.bincount(prob["rp_point"]).min() >= 5
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=15))
    assert s.is_solution_usable and s.final_cost < 0.5 * s.initial_cost, (s.initial_cost, s.final_cost)
    cost, res, sq = ba.evaluate(True, True)
    assert abs(cost - s.final_cost) <= 1e-9 * s.final_cost
