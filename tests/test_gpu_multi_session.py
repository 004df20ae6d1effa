"""GPU (-m gpu): BASELINE configs[4] at its stated shape -- 16 sessions of 500 keyframes / 50 000 features over one place with 200 mapped
objects, chained through the long-term map (the reference chains its sessions the same way, one after the other:
ltm_trajectory_sequence_executor.py:45-92; session s needs the map of session s - 1, so the chain does not shard: one GPU).

Session s: the objects start from the map (estimates + marginal 7x7 covariances as IndependentObjectMapFactor priors,
long_term_map_factor_creator.h:265-322), two-phase bundle adjustment, then the new map is extracted on the device
(obvi_ba_object_covariances: the blocks ceres::Covariance gives long_term_object_map_extraction.cpp:419-433).

Too large for the oracle (the two-session chain against the oracle is tests/test_gpu_parity.py::test_multi_session_chain_through_the_long_term_map);
here the properties the chain must have: every session maps the objects it saw, the map's error against the synthetic truth falls as sessions
accumulate, the reported standard deviations fall with it and stay calibrated (the error is of the size the covariance says)."""
import numpy as np
import pytest

import helpers
import obvi_ba
import synth

pytestmark = pytest.mark.gpu


def test_sixteen_sessions_chained_through_the_long_term_map():
    n_sessions, n_objects = 16, 200
    prm = obvi_ba.SolverParams(max_num_iterations=50, allow_non_monotonic_steps=True, function_tolerance=1e-4, gradient_tolerance=1e-10,
                               parameter_tolerance=1e-8, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
    g = helpers.product_ba()
    ltm, err_med, sd_med, ratio_med, mapped = None, [], [], [], []
    known = np.zeros(n_objects, bool)
    for s in range(n_sessions):
        prob = synth.make_problem(P=500, L=50000, O=n_objects, seed=1000 + s, object_seed=77, const_poses=1, min_obj_obs=10, object_classes=("bench",))
        assert len(prob["objects"]) == n_objects and len(prob["rp_pose"]) > 400000
        if ltm is not None:
            prob["objects"][ltm[0]] = ltm[1]                          # a mapped object starts from the map
            prob.update(lt_obj=ltm[0].astype(np.uint32), lt_mean=ltm[1], lt_cov=ltm[2].reshape(-1, 49), lt_huber=1.0)
        synth.upload(g, prob)
        s1 = g.solve(prm)                                             # phase I
        mask, nex = g.select_outliers(0, 0.1)                         # phase II without the worst 10 % of the visual factors
        assert nex > 0.05 * len(mask)
        g.set_active_mask(0, mask)
        s2 = g.solve(prm)
        assert s1.termination_type != obvi_ba.FAILURE and s2.termination_type != obvi_ba.FAILURE and s2.final_cost < s1.initial_cost
        g.set_active_mask(0, np.ones_like(mask))                      # the extraction problem holds every factor again
        ids = np.arange(n_objects, dtype=np.uint32)
        cov = g.object_covariances(ids)
        est = g.get_objects()
        seen = np.abs(cov).max(axis=(1, 2)) > 0
        assert np.all(seen[known])                                    # an object of the map stays in the map (its prior keeps it variable)
        known |= seen
        assert np.all(np.linalg.eigvalsh(cov[seen]) > 0)              # every extracted block is a covariance
        err = np.linalg.norm(est[seen, :3] - prob["gt_objects"][seen, :3], axis=1)
        sd = np.sqrt(np.einsum("oii->oi", cov[seen])[:, :3].sum(axis=1))         # sigma of the centre (root of the trace of the position block)
        err_med.append(np.median(err)); sd_med.append(np.median(sd)); ratio_med.append(np.median(err / sd)); mapped.append(int(seen.sum()))
        ltm = (ids[seen], est[seen], cov[seen])
    print("objects mapped", mapped, "\ncentre error median", np.round(err_med, 3), "\nsigma median", np.round(sd_med, 3), "\nerror / sigma median", np.round(ratio_med, 2))
    assert mapped[-1] >= 0.9 * n_objects and all(b >= a for a, b in zip(mapped, mapped[1:]))
    # the map gets better and knows it: late sessions against the first, and no session far above its predecessor
    assert np.mean(err_med[-4:]) < 0.5 * err_med[0] and np.mean(sd_med[-4:]) < 0.5 * sd_med[0]
    assert all(b < 1.25 * a for a, b in zip(sd_med, sd_med[1:]))
    # calibrated: the typical error is of the size of the reported sigma, in every session
    assert all(0.2 < r < 3.0 for r in ratio_med), ratio_med
