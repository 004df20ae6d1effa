"""GPU (-m gpu): the 9-parameter ellipsoid block (obvi_ba_options.object_block_size = 9; north_star "9-DoF ellipsoid parameter blocks", SURVEY 8(b) `object 7|9`,
vslam_obj_opt_types_refactor.h:15-21, ellipsoid_utils.h:217-229 `#else`) through the C ABI against the CPU oracle: linearisation of every object factor, reduced
system, LM trajectory (default and deterministic handle, the 16-lane and the scratch-and-gather schedule of the small factors), covariance blocks 9 x 9, two-phase
selection, shared objects across two handles (90 doubles per object in the first collective).  The oracle's 9-block restatement is pinned in tests/test_golden.py
(numpy + surface sampling; the reference cannot compile the branch) and tests/test_oracle_solver.py (dense normal equations).
Tolerances as in test_gpu_parity.py."""
import threading

import numpy as np
import pytest
import torch  # noqa: F401  (loaded before libobvi_ba.so: torch brings its own HIP runtime, and the one that is loaded first is the one that finds the device)

import helpers
import obvi_ba
import synth
from helpers import rel_err

pytestmark = pytest.mark.gpu


def with_ltm(prob, seed=1):
    O = len(prob["objects"])
    A = np.random.default_rng(seed).normal(size=(O, 7, 7))
    prob.update(lt_obj=np.arange(O, dtype=np.uint32), lt_mean=prob["gt_objects"] + 0.05, lt_cov=(A @ A.transpose(0, 2, 1) + 7 * np.eye(7)).reshape(O, 49) * 0.01, lt_huber=1.0)
    return prob


@pytest.fixture(scope="module")
def small9():
    return synth.nine_dof(with_ltm(synth.make_problem(P=30, L=400, O=3, seed=1, object_classes=("bench",), bbox_noise=5.0, min_obj_obs=5)), tilt=0.3, seed=2)


def pair(prob, **opts):
    o, g = helpers.oracle_ba(object_block_size=9), helpers.product_ba(object_block_size=9, **opts)
    for ba in (o, g):
        synth.upload(ba, prob)
    return o, g


def test_evaluate_and_linearisation(small9):
    o, g = pair(small9)
    assert g.get_objects().shape == (3, 9) and np.array_equal(g.get_objects(), small9["objects"])
    for loss in (True, False):
        co, ro, so = o.evaluate(loss); cg, rg, sg = g.evaluate(loss)
        assert abs(cg - co) <= 1e-12 * co and np.abs(rg - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max()) and np.abs(sg - so).max() <= 1e-12 * max(1.0, np.abs(so).max())
    for t in (2, 3, 4):
        ro, J0o, J1o = o.debug_linearize(t); rg, J0g, J1g = g.debug_linearize(t)
        assert J0g.shape[2] == 9 and np.abs(rg - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max()), t
        assert rel_err(J0g, J0o) < 1e-12, t
        if J1o is not None:
            assert rel_err(J1g, J1o) < 1e-12, t


@pytest.mark.parametrize("knob", ["default", "deterministic", "scratch_and_gather"])
def test_reduced_system_and_lm_trajectory(small9, knob, monkeypatch):
    if knob == "scratch_and_gather":
        monkeypatch.setenv("OBVI_SMALL_LANES_BELOW", "0")           # the big-problem schedule of the small factors: per-factor slots (81 doubles) + k_bbox_gather
    o, g = pair(small9, deterministic=(knob == "deterministic"))
    for radius in (100.0, 0.5):
        So, bo = o.debug_reduced_system(radius); Sg, bg = g.debug_reduced_system(radius)
        assert So.shape == Sg.shape and So.shape[0] == 6 * 29 + 9 * 3
        assert rel_err(Sg, So) < 1e-11 and rel_err(bg, bo) < 1e-10
    prm = helpers.ba_params(max_it=40)
    so, sg = o.solve(prm), g.solve(prm)
    assert sg.termination_type == so.termination_type and sg.num_iterations == so.num_iterations and sg.num_iterations > 4
    for a, b in zip(o.iterations(), g.iterations()):
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-8 * a.cost
    assert np.abs(g.get_poses() - o.get_poses()).max() < 1e-8 and np.abs(g.get_objects() - o.get_objects()).max() < 1e-6
    assert sg.num_parameters_reduced == so.num_parameters_reduced and sg.reduced_system_size == so.reduced_system_size == 6 * 29 + 9 * 3


def test_covariance_blocks_are_nine_by_nine(small9):
    o, g = pair(small9)
    prm = helpers.ba_params(max_it=10)
    o.solve(prm); g.solve(prm)
    a, b = np.array([0, 1, 2, 0, 2]), np.array([0, 1, 2, 1, 0])
    co, cg = o.object_covariances(a, b), g.object_covariances(a, b)
    assert cg.shape == (5, 9, 9)
    for i in range(5):
        assert np.abs(cg[i] - co[i]).max() <= 1e-7 * np.abs(co[i]).max(), i
    no, ng = o.column_sqnorms(), g.column_sqnorms()
    assert ng[2].shape == (3, 9) and rel_err(ng[2], no[2]) < 1e-11


def test_two_phase_window_with_nine_parameter_objects():
    prob = synth.nine_dof(synth.make_problem(P=50, L=3000, O=6, seed=5, object_classes=("bench",), min_obj_obs=8), tilt=0.2, seed=3)
    o, g = pair(prob)
    legs = {}
    import end_state
    for name, ba in (("oracle", o), ("hip", g)):
        legs[name] = end_state.run_two_phase(ba, prob, obvi_ba, synth, block=end_state.LOCAL_BA, polish_iterations=0, upload=False)
    c = end_state.compare(legs["hip"], legs["oracle"])
    print(c["phase_1"], c["phase_2"], c["state_after_phase_2"])
    assert c["same_excluded_sets"] and c["phase_1"]["same_lm_sequence"] and c["phase_1"]["final_cost_rel"] < 1e-8
    # Phase II of this window is a chaotic run for any two fp64 implementations (a tilted ellipsoid's rotation about its long axis is barely observed; the run stops on a
    # relative cost change of 1e-4): measured over the round 23 against 18 iterations with costs 1.9e-4 apart, 12 against 30, and 18 = 18 with the same accept / reject
    # sequence and still 2.2e-4 apart.  Both start it from the same state with the same factors (asserted above); what is held here is the valley.  The LM trajectory of
    # the 9-parameter block itself is held to 1e-8 over 40 iterations in test_reduced_system_and_lm_trajectory.
    assert c["phase_2"]["final_cost_rel"] <= 1e-2, c["phase_2"]


def test_upright_nine_blocks_reproduce_the_seven_block_handle():
    """The 7 path's numbers must not move, and a 9-block handle given upright objects evaluates the 7-block handle's problem: same cost and residuals to the
    last digits, the reduced system with the (ax, ay) rows struck out equal to the 7-block's."""
    p7 = synth.make_problem(P=30, L=400, O=3, seed=1, object_classes=("bench",), bbox_noise=5.0, min_obj_obs=5)
    p9 = synth.nine_dof(p7, tilt=0.0)
    g7, g9 = helpers.product_ba(), helpers.product_ba(object_block_size=9)
    synth.upload(g7, p7); synth.upload(g9, p9)
    c7, r7, _ = g7.evaluate(True, True); c9, r9, _ = g9.evaluate(True, True)
    assert abs(c9 - c7) <= 1e-13 * c7 and np.abs(r9 - r7).max() < 1e-11
    S7, b7 = g7.debug_reduced_system(100.0); S9, b9 = g9.debug_reduced_system(100.0)
    keep = list(range(6 * 29)) + [6 * 29 + 9 * o + k for o in range(3) for k in (0, 1, 2, 5, 6, 7, 8)]
    assert rel_err(S9[np.ix_(keep, keep)], S7) < 1e-11 and rel_err(b9[keep], b7) < 1e-10


def test_object_block_size_is_checked():
    with pytest.raises(obvi_ba.ObviError):
        helpers.product_ba(object_block_size=8)
    g = helpers.product_ba(object_block_size=9)
    with pytest.raises((obvi_ba.ObviError, ValueError)):
        g.set_objects(np.zeros((2, 7)))


def test_two_handles_sharing_nine_parameter_objects_land_on_the_oracles_joint_solve():
    """The multi-GPU exchange with the larger block: 90 doubles per shared object in the first collective, a 9-row block per object in the shared tail."""
    import dist_util  # noqa: F401
    from test_gpu_shared_objects import EmulatedAllReduce, split_problem
    scene = synth.make_problem(P=60, L=900, O=3, seed=33, min_obj_obs=6, object_classes=("bench",), bbox_noise=5.0)
    wins, joint, keep_pts = split_problem(scene, 30)
    joint9 = synth.nine_dof(joint, tilt=0.25, seed=9)
    prm = helpers.ba_params(max_it=12)
    orc = helpers.oracle_ba(object_block_size=9); synth.upload(orc, joint9); sorc = orc.solve(prm)
    emu = EmulatedAllReduce(2)
    handles, out = [], [None, None]
    for rank, (q, pts, rng) in enumerate(wins):
        q9 = dict(q); q9["objects"] = joint9["objects"]
        ba = helpers.product_ba(object_block_size=9)
        synth.upload(ba, q9)
        ba.set_shared_objects(np.ones(3, np.uint8), rank, 2)
        ba.set_allreduce(emu.hook(rank))
        handles.append(ba)

    def run(rank):
        out[rank] = handles[rank].solve(prm)
    torch.cuda.synchronize()                     # (torch's CUDA state comes up on the main thread, not lazily inside a hook on a worker thread)
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert all(o is not None for o in out)
    assert any(n == 90 * 3 for n, _ in emu.log)
    for rank in range(2):
        assert out[rank].num_iterations == sorc.num_iterations and abs(out[rank].final_cost - sorc.final_cost) <= 1e-8 * sorc.final_cost
        assert np.abs(handles[rank].get_objects() - orc.get_objects()).max() < 2e-5          # measured 3.4e-6 (a tilted ellipsoid's rotation about its long axis is weakly observed)


def test_deterministic_reruns_and_the_object_only_configurations(small9):
    """The 9-parameter block in the other shapes the path takes: two deterministic solves are the same bits; the pending-object refinement (every pose constant, no
    features: pending_object_estimator.cpp:11-151) and the pose-graph + objects stage (no visual factors: pose_graph_plus_objects_optimizer.h:23-353) follow the oracle."""
    runs = []
    for _ in range(2):
        g = helpers.product_ba(object_block_size=9, deterministic=True)
        synth.upload(g, small9)
        s = g.solve(helpers.ba_params(max_it=12))
        runs.append((s.final_cost, [i.cost for i in g.iterations()], g.get_objects(), g.get_poses(), g.object_covariances(np.arange(3))))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert all(np.array_equal(a, b) for a, b in zip(runs[0][2:], runs[1][2:]))
    # pending objects: poses constant, bounding boxes + shape priors only
    pend = dict(small9); pend["pose_const"] = np.ones(len(small9["poses"]), np.uint8)
    o, g = helpers.oracle_ba(object_block_size=9), helpers.product_ba(object_block_size=9)
    for ba in (o, g):
        synth.upload(ba, pend, relpose=False, reproj=False)
    so, sg = o.solve(helpers.ba_params(max_it=30)), g.solve(helpers.ba_params(max_it=30))
    assert sg.num_iterations == so.num_iterations and abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.abs(g.get_objects() - o.get_objects()).max() < 1e-6 and sg.reduced_system_size == 27
    # pose graph + objects: relative-pose factors, bounding boxes, priors; no features
    o, g = helpers.oracle_ba(object_block_size=9), helpers.product_ba(object_block_size=9)
    for ba in (o, g):
        synth.upload(ba, small9, reproj=False)
    so, sg = o.solve(helpers.ba_params(max_it=25)), g.solve(helpers.ba_params(max_it=25))
    assert sg.num_iterations == so.num_iterations and abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.abs(g.get_poses() - o.get_poses()).max() < 1e-7 and np.abs(g.get_objects() - o.get_objects()).max() < 1e-5
