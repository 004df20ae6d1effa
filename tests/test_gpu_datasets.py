"""GPU: the HIP path on the reference's own data sets (fixtures under tests/golden/, see test_reference_datasets.py for
the same checks on the oracle).  Known answers: zero residual at the simulated sets' ground truth, bundle adjustment
returns to it; real data (BASELINE config #1, ORB-SLAM2 tracks of TUM fr2/pioneer_360): same LM trajectory as the oracle."""
import numpy as np
import pytest

import dataset_io
import helpers
import synth
from test_reference_datasets import GT_PIXEL_TOL, GT_POSE_TOL, perturbed

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(GT_PIXEL_TOL))
def test_residual_vanishes_at_the_data_sets_ground_truth(name):
    prob = dataset_io.problem_from_dataset(dataset_io.load_fixture(name), min_obs=1)
    g = helpers.product_ba(); synth.upload(g, prob)
    cost, res, sq = g.evaluate(False, True)
    assert (np.abs(res) * prob["rp_sigma"]).max() < GT_PIXEL_TOL[name]
    o = helpers.oracle_ba(); synth.upload(o, prob)
    # residuals here are differences of O(1) rectified coordinates times f / sigma: compare absolutely at that scale
    assert np.abs(res - o.evaluate(False, True)[1]).max() < 1e-12 * prob["K"][0, 0] / prob["rp_sigma"]


@pytest.mark.parametrize("name", sorted(GT_POSE_TOL))
def test_bundle_adjustment_returns_to_ground_truth(name):
    gt = dataset_io.problem_from_dataset(dataset_io.load_fixture(name), min_obs=3, const_poses=2)
    prob = perturbed(gt, seed=7)
    prm = helpers.ba_params(max_it=60, ftol=1e-14, gtol=1e-14, ptol=1e-14)
    g = helpers.product_ba(); synth.upload(g, prob)
    sg = g.solve(prm)
    assert sg.is_solution_usable and sg.final_cost < 1e-4 * sg.initial_cost
    assert np.abs(g.get_poses() - gt["poses"]).max() < GT_POSE_TOL[name]
    o = helpers.oracle_ba(); synth.upload(o, prob)
    so = o.solve(prm)
    for a, b in zip(o.iterations(), g.iterations()):                     # down to the rounding level: the same trajectory
        if a.cost <= 1e-9 * so.initial_cost:
            break
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-8 * a.cost + 1e-12 * so.initial_cost
    # at the minimum the cost is rounding noise of the text (1e-8 of the start): both land in the same place
    assert np.abs(g.get_poses() - o.get_poses()).max() < 1e-6
    assert abs(sg.final_cost - so.final_cost) <= 1e-6 * so.initial_cost


def test_tum_tracks_same_trajectory_as_the_oracle():
    """BASELINE config #1 (380 frames, tracks with >= 5 sightings, repeated ids in a frame, frames without any
    sighting): real, ragged data through both paths."""
    prob = dataset_io.tum_problem()
    stats = synth.problem_stats(prob)
    assert stats["P"] == 380 and stats["N_r"] > 30000
    prm = helpers.ba_params(max_it=12)
    o = helpers.oracle_ba(); synth.upload(o, prob)
    g = helpers.product_ba(); synth.upload(g, prob)
    ro, rg = o.evaluate(False, True), g.evaluate(False, True)
    assert abs(rg[0] - ro[0]) <= 1e-12 * ro[0] and helpers.rel_err(rg[1], ro[1]) < 1e-12
    so, sg = o.solve(prm), g.solve(prm)
    assert sg.num_iterations == so.num_iterations and sg.termination_type == so.termination_type
    for a, b in zip(o.iterations(), g.iterations()):
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-8 * a.cost
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost and sg.final_cost < 0.5 * sg.initial_cost
    assert np.abs(g.get_poses() - o.get_poses()).max() < 1e-7
    assert sg.num_parameters_reduced == so.num_parameters_reduced
