"""GPU (-m gpu): the library's internal feature numbering (round 6; csrc/ba_handle.h, upload.cpp obvi_ba_set_reproj).  From 2^18 observations on -- here from the
first, OBVI_POINT_RENUMBER_MIN=1 -- the kernels index features in the order of their first observing pose, so that a wavefront's observations belong to neighbouring
poses; the caller's numbering stops at the ABI.  Everything a caller can say or ask about a feature by index must be unaffected: values in and out, constness flags,
parameter priors, column norms, snapshots, values handed over later, a second structure on the same handle -- against the oracle, on a problem whose feature ids
are shuffled with respect to the trajectory."""
import numpy as np
import pytest

import helpers
import obvi_ba
import synth

pytestmark = pytest.mark.gpu


def shuffled(prob, seed=3):
    """the same problem with the features renumbered at random (observations re-sorted by the new ids)"""
    q = dict(prob)
    L = len(prob["points"])
    new_of_old = np.random.default_rng(seed).permutation(L)
    old_of_new = np.argsort(new_of_old)
    q["points"], q["gt_points"], q["point_const"] = prob["points"][old_of_new], prob["gt_points"][old_of_new], prob["point_const"][old_of_new]
    rp = new_of_old[prob["rp_point"]].astype(np.uint32)
    order = np.lexsort((prob["rp_cam"], prob["rp_pose"], rp))
    for k in ("rp_pose", "rp_cam", "rp_pixel", "rp_is_outlier"):
        q[k] = prob[k][order]
    q["rp_point"] = rp[order]
    return q


@pytest.fixture()
def renumber(monkeypatch):
    monkeypatch.setenv("OBVI_POINT_RENUMBER_MIN", "1")


def test_every_feature_accessor_speaks_the_callers_numbering(renumber):
    prob = shuffled(synth.make_problem(P=40, L=1500, O=3, seed=2, object_classes=("bench",), bbox_noise=5.0, min_obj_obs=5))
    prob["point_const"] = (np.random.default_rng(0).random(len(prob["points"])) < 0.05).astype(np.uint8)
    o, g = helpers.oracle_ba(), helpers.product_ba()
    for ba in (o, g):
        synth.upload(ba, prob)
    assert np.array_equal(g.get_points(), prob["points"])                      # out as they went in
    co, ro, so = o.evaluate(True); cg, rg, sg = g.evaluate(True)
    assert abs(cg - co) <= 1e-12 * co and np.abs(rg - ro).max() <= 1e-12 * np.abs(ro).max()
    r_o, J0o, J1o = o.debug_linearize(0); r_g, J0g, J1g = g.debug_linearize(0)
    assert helpers.rel_err(J0g, J0o) < 1e-12 and helpers.rel_err(J1g, J1o) < 1e-12
    no, ng = o.column_sqnorms(), g.column_sqnorms()
    assert np.array_equal(ng[1] < 0, no[1] < 0) and helpers.rel_err(ng[1], no[1]) < 1e-11          # -1 exactly at the caller's constant / unobserved features
    # values handed over later, feature by feature
    moved = prob["points"] + np.random.default_rng(1).normal(size=prob["points"].shape) * 0.01
    for ba in (o, g):
        ba.update_points(moved)
    assert np.array_equal(g.get_points(), moved)
    assert abs(g.evaluate(True, False)[0] - o.evaluate(True, False)[0]) <= 1e-12 * o.evaluate(True, False)[0]
    for ba in (o, g):
        ba.update_state(points=prob["points"])
    assert np.array_equal(g.get_state()[1], prob["points"])
    # snapshot / solve / restore, then the solve itself against the oracle
    prm = helpers.ba_params(max_it=15)
    g.snapshot(); g.solve(prm); g.restore()
    assert np.array_equal(g.get_points(), prob["points"])
    so_, sg_ = o.solve(prm), g.solve(prm)
    assert sg_.num_iterations == so_.num_iterations and abs(sg_.final_cost - so_.final_cost) <= 1e-8 * so_.final_cost
    assert np.abs(g.get_points() - o.get_points()).max() < 1e-6 and np.abs(g.get_poses() - o.get_poses()).max() < 1e-8
    const = prob["point_const"] == 1
    assert np.array_equal(g.get_points()[const], prob["points"][const])        # the caller's constant features did not move
    # constness flags and parameter priors by the caller's index
    flags = np.zeros(len(prob["points"]), np.uint8); flags[::7] = 1
    for ba in (o, g):
        ba.set_poses(prob["poses"], prob["pose_const"]); ba.set_objects(prob["objects"], prob["object_const"]); ba.set_points(prob["points"], prob["point_const"])
        ba.set_const_flags(point_const=flags)
    so_, sg_ = o.solve(helpers.ba_params(max_it=4)), g.solve(helpers.ba_params(max_it=4))
    assert abs(sg_.final_cost - so_.final_cost) <= 1e-8 * so_.final_cost and np.array_equal(g.get_points()[::7], prob["points"][::7])
    kinds, blocks, params = [1, 1, 0], [5, 1200, 3], [2, 0, 4]
    for ba in (o, g):
        ba.set_parameter_priors(kinds, blocks, params, [0.0, 0.0, 0.0], [0.1, 0.2, 0.3])
    no, ng = o.column_sqnorms(), g.column_sqnorms()
    assert helpers.rel_err(ng[1], no[1]) < 1e-11
    co_, cg_ = o.object_covariances(np.arange(3)), g.object_covariances(np.arange(3))
    assert np.abs(cg_ - co_).max() <= 1e-7 * np.abs(co_).max()


def test_a_second_structure_on_the_same_handle_and_a_reset(renumber):
    a = shuffled(synth.make_problem(P=30, L=800, O=0, seed=5), seed=1)
    b = shuffled(synth.make_problem(P=30, L=800, O=0, seed=6), seed=2)          # the same feature count, other tracks: another internal order
    g, o = helpers.product_ba(), helpers.oracle_ba()
    synth.upload(g, a)
    g.solve(helpers.ba_params(max_it=3))
    kept = g.get_points()
    g.set_reproj(b["rp_pose"], b["rp_point"], b["rp_cam"], b["rp_pixel"], b["rp_sigma"], b["rp_huber"])     # values stay, structure changes
    assert np.array_equal(g.get_points(), kept)
    bb = dict(b); bb["points"] = kept; bb["poses"] = g.get_poses()
    bb.update({k: a[k] for k in a if k.startswith("rl_")})                       # (the handle still holds a's odometry factors)
    synth.upload(o, bb)
    assert abs(g.evaluate(True, False)[0] - o.evaluate(True, False)[0]) <= 1e-12 * o.evaluate(True, False)[0]
    g.reset()
    synth.upload(g, a); synth.upload(o, a)
    so_, sg_ = o.solve(helpers.ba_params(max_it=6)), g.solve(helpers.ba_params(max_it=6))
    assert sg_.num_iterations == so_.num_iterations and abs(sg_.final_cost - so_.final_cost) <= 1e-8 * so_.final_cost
    assert np.abs(g.get_points() - o.get_points()).max() < 1e-6


def test_renumbered_and_callers_order_give_the_same_solve(monkeypatch):
    prob = shuffled(synth.make_problem(P=60, L=3000, O=4, seed=8, object_classes=("bench",), min_obj_obs=6))
    out = []
    for min_obs in ("0", "1"):
        monkeypatch.setenv("OBVI_POINT_RENUMBER_MIN", min_obs)
        g = helpers.product_ba(deterministic=True)
        synth.upload(g, prob)
        s = g.solve(helpers.ba_params(max_it=10))
        out.append((s, g.get_points(), g.get_poses()))
    (sa, xa, pa), (sb, xb, pb) = out
    assert sa.num_iterations == sb.num_iterations and abs(sa.final_cost - sb.final_cost) <= 1e-10 * sa.final_cost     # another summation order, nothing else
    assert np.abs(xa - xb).max() < 1e-7 and np.abs(pa - pb).max() < 1e-9


def test_deterministic_mode_stays_bit_identical_under_the_internal_numbering(renumber):
    prob = shuffled(synth.make_problem(P=50, L=2500, O=3, seed=11, object_classes=("bench",), min_obj_obs=6))
    runs = []
    for _ in range(2):
        g = helpers.product_ba(deterministic=True)
        synth.upload(g, prob)
        s = g.solve(helpers.ba_params(max_it=10))
        runs.append(([i.cost for i in g.iterations()], g.get_points(), g.get_poses(), g.get_objects()))
    assert runs[0][0] == runs[1][0] and all(np.array_equal(a, b) for a, b in zip(runs[0][1:], runs[1][1:]))
