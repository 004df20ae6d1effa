"""GPU (-m gpu): every optimisation of a sliding-window session, HIP path against the CPU oracle FROM THE SAME START.

Two whole sessions (one per backend) cannot be compared exactly: round-off of different summation orders is amplified by each loosely
converged LM run and handed on to the next window, until a tolerance test or the 10 % outlier cut falls the other way in one of them
(tests/test_gpu_deterministic.py measures where).  Here the host mirror's driver is built against BOTH backends (tests/lockstep_shim.*):
every call of the C ABI goes to libobvi_ba.so and to the oracle, the two answers are compared and logged, and after every solve the
oracle takes over the HIP path's result -- so each of the session's 162 optimisations (local BAs in two phases, pose-graph stages,
global BAs, the final BA, the long-term-map extraction) is an independent parity statement: same LM iteration count, same accept /
reject sequence, same termination, costs and parameter blocks to the stated tolerances, identical outlier masks.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_host_mirror import HOST, scene  # noqa: F401  (fixture: the 80-frame scene)

pytestmark = pytest.mark.gpu
LOCKSTEP_DRIVER = os.path.join(helpers.ROOT, "tests", "run_offline_ba_lockstep")


@pytest.fixture(scope="module")
def lockstep_driver():
    tests = os.path.join(helpers.ROOT, "tests")
    srcs = [os.path.join(HOST, "run_offline_ba.cpp"), os.path.join(tests, "lockstep_shim.cpp"), os.path.join(tests, "lockstep_shim.h")]
    deps = srcs + [helpers.ensure_oracle(), helpers.PRODUCT_LIB] + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if not os.path.exists(LOCKSTEP_DRIVER) or os.path.getmtime(LOCKSTEP_DRIVER) < max(os.path.getmtime(d) for d in deps):
        obj = os.path.join(tests, "lockstep_shim.o")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-c", "-o", obj, srcs[1]])          # the shim itself sees the real names
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(helpers.ROOT, "include"), "-I" + HOST, "-include", srcs[2], "-o", LOCKSTEP_DRIVER, srcs[0], obj,
                               "-L" + os.path.dirname(helpers.PRODUCT_LIB), "-lobvi_ba", "-L" + os.path.join(helpers.ROOT, "oracle"), "-lobvi_oracle",
                               "-Wl,-rpath,$ORIGIN/../obvi-slam_amd/csrc", "-Wl,-rpath,$ORIGIN/../oracle", "-ldl"])
    return LOCKSTEP_DRIVER


@pytest.mark.parametrize("deterministic", [False, True])
def test_every_optimisation_of_a_session_follows_the_oracle(lockstep_driver, scene, tmp_path, deterministic):  # noqa: F811
    prob, path, _ = scene
    out, log = str(tmp_path / "out.json"), str(tmp_path / "lockstep.jsonl")
    cmd = [lockstep_driver, path, out, "--window", "20", "--gba-frequency", "25", "--ltm"] + (["--deterministic"] if deterministic else [])
    subprocess.check_call(cmd, timeout=1800, env=dict(os.environ, OBVI_LOCKSTEP_LOG=log))
    recs = [json.loads(ln) for ln in open(log)]
    assert not [r for r in recs if r["call"] == "status"]                                        # no call succeeded on one backend and failed on the other
    solves = [r for r in recs if r["call"] == "solve"]
    assert len(solves) >= len(json.load(open(out))["records"]) >= 150                            # (+ the solves of the long-term-map extraction's rank repair)
    busy = [r for r in solves if r["initial_cost"] > 1e-3]
    # the same LM run: iteration count, accept / reject sequence, termination, reduced program
    def same_run(r):
        return r["iterations_hip"] == r["iterations_oracle"] and r["termination_hip"] == r["termination_oracle"] and r["same_accept_sequence"] == 1 and r["params_reduced_equal"] == 1
    good, bad = [r for r in busy if same_run(r)], [r for r in busy if not same_run(r)]
    keys = ("initial_cost_rel", "final_cost_rel", "max_iteration_cost_rel", "pose_diff", "point_diff", "object_diff")
    worst = {k: max(r[k] for r in good) for k in keys}
    med = {k: float(np.median([r[k] for r in good])) for k in keys}
    print("lock-step session (%s): %d optimisations, %d with work, %d of them the oracle's LM run step for step; those: worst %s median %s; the other %d: %s"
          % ("deterministic" if deterministic else "default", len(solves), len(busy), len(good), {k: "%.1e" % v for k, v in worst.items()}, {k: "%.1e" % v for k, v in med.items()}, len(bad),
             [(r["iterations_hip"], r["iterations_oracle"], "%.1e" % r["final_cost_rel"], "%.1e" % r["pose_diff"]) for r in bad]))
    if os.path.isdir(os.path.join(helpers.ROOT, "gpurun_out")):
        import shutil
        shutil.copy(log, os.path.join(helpers.ROOT, "gpurun_out", "lockstep_%s.jsonl" % ("det" if deterministic else "default")))
    assert all(r["params_reduced_equal"] == 1 for r in solves) and max(r["initial_cost_rel"] for r in busy) <= 1e-11     # the same problem, the same objective at the same point: everywhere
    # (1) Where the two LM runs are the same run: the stated end-state tolerances.
    assert med["final_cost_rel"] <= 1e-10 and med["pose_diff"] <= 1e-10 and med["point_diff"] <= 1e-9      # measured: 5e-13, 1.4e-13, 3.5e-12
    # END state of the worst such run over ten sessions (1 551 runs): cost 2.3e-5, poses 1.6e-5.  An INTERMEDIATE iterate may be further off than the end state (a
    # 9-iteration window: 1.6e-4 at one iterate, 9e-7 at the end; the tail over those 1 551 runs: 1.6e-4, 8.8e-5, 8.3e-5, 6.1e-5, ...): it gets a sanity bound only
    assert worst["final_cost_rel"] <= 2e-4 and worst["pose_diff"] <= 1e-4 and worst["max_iteration_cost_rel"] <= 1e-3
    # objects: in units of the oracle's OWN 7x7 covariance block of the object (sqrt(d^T Sigma^-1 d), yaw included; tests/lockstep_shim.cpp whitened_object_diff) -- the
    # absolute difference (median 5e-8 m, but 0.26 m for an object a window sees from a few frames along one viewing ray) says little: what that bar of 1.0 could hide is
    # a defect in a well-constrained block, and the whitened difference cannot
    wh = [r["object_diff_whitened"] for r in good if r["objects"] > 0 and r["object_diff_whitened"] >= 0.0]
    print("whitened object differences over %d runs with objects (%d without a covariance): median %.2e, 90%% %.2e, worst %.2e; absolute: median %.2e worst %.2e"
          % (len(wh), sum(1 for r in good if r["objects"] > 0 and r["object_diff_whitened"] < 0.0), np.median(wh), np.quantile(wh, 0.9), max(wh), med["object_diff"], worst["object_diff"]))
    # What the bar can be: two runs of a LOOSELY converged window (function tolerance 1e-3 / 1e-4) end a fraction of a sigma apart in their weakest object -- the fp64
    # oracle against its own extended-precision build on six windows of this size: 5e-7 ... 0.54 sigma (yaw taken modulo pi; without that, thousands of radians of drift
    # in the yaw of ellipsoids with equal horizontal axes).  So: typically far below a sigma, never beyond one -- a block that is wrong by more than the oracle's own
    # uncertainty about it fails.
    # Measured (two sessions, 132-133 runs with objects): median 4e-5 / 6e-4 sigma, 90 % 0.06 / 0.09, worst 0.20 / 0.22; half of the runs below 1e-3.
    assert len(wh) >= 50 and med["object_diff"] <= 1e-6 and np.median(wh) <= 1e-2 and np.quantile(wh, 0.9) <= 0.5 and max(wh) <= 1.0
    # (2) Where they are not, the END STATE says why, run by run (round 5; before: a bar on the share of such runs, taken from its distribution over sixty sessions):
    #   (a) the stopping rule.  A run of the reference's blocks ends when |cost change| <= function_tolerance * cost; decided in the last bits, two runs that agree to
    #       1e-12 up to there stop k = 1 ... 4 iterations apart, and their final costs then differ by about k x that tolerance -- measured 0.8 ... 1.7 x k x the tolerance
    #       of the solve (1e-4 for phase II of a local BA, 1e-6 for the final BA), poses within 3e-3 m.  tests/test_gpu_end_state.py shows the same against the extended-precision arbiter: the fp64
    #       oracle itself stops one iteration from the arbiter there, 1e-4 apart.
    #   (b) long runs on a problem with a direction the data barely constrains (pose-graph stages and global BAs of 25 ... 250 iterations, non-monotonic steps): the
    #       trajectories of ANY two fp64 runs separate (profiles/r05_end_state_config3.txt: HIP default, HIP deterministic, the oracle and the oracle on one thread
    #       fewer end pairwise as far apart as HIP and the oracle do here); both end at equally good points: costs within 1e-2, poses within 5e-2.
    #   Anything else -- a short run that differs by more than its own tolerance explains -- fails.
    def stopping_rule(r):
        # k iterations apart: each of the iterations the longer run went on for lowered the cost by more than the tolerance (or it would have stopped) and, this
        # close to the end, by not much more -- measured 0.8 ... 1.7 x k x tolerance over the sessions of rounds 4-5; allowed 3 (k + 1) x.  (k itself: up to 5 on the
        # local BAs; a final BA at tolerance 1e-6 was seen to go on for 7 more -- 16 against 23 iterations, costs 8.5e-6 apart = 1.2 x k x tolerance; the bound that
        # binds is the cost one, which grows with k)
        k = abs(r["iterations_hip"] - r["iterations_oracle"])
        return k <= 10 and max(r["iterations_hip"], r["iterations_oracle"]) <= 40 and \
            r["final_cost_rel"] <= 3.0 * (k + 1) * max(r["function_tolerance"], 1e-8) and r["pose_diff"] <= 5e-3   # (profiles/r05_lockstep_distribution.txt, 40 sessions: k <= 7, ratio median 0.51, max 1.99)
    def long_run(r):
        return max(r["iterations_hip"], r["iterations_oracle"]) >= 25 and r["final_cost_rel"] <= 2e-2 and r["pose_diff"] <= 5e-2
    unexplained = [r for r in bad if not (stopping_rule(r) or long_run(r))]
    assert not unexplained, unexplained
    assert len(good) >= 0.75 * len(busy)          # sanity only (measured 147 ... 158 of 165): what matters is (1) and (2), every run is in one of them
    # identical outlier selections (two-phase cut), evaluations and covariance blocks
    sel = [r for r in recs if r["call"] == "select_outliers"]
    assert len(sel) >= 100 and all(r["masks_differ"] == 0 and r["excluded_hip"] == r["excluded_oracle"] for r in sel)
    ev = [r for r in recs if r["call"] == "evaluate"]   # (none when the runner cuts the outliers on the device: the evaluation then happens inside select_outliers)
    assert not ev or (max(r["cost_rel"] for r in ev) <= 1e-11 and max(r["sqnorm_rel"] for r in ev) <= 1e-11)
    cov = [r for r in recs if r["call"] == "object_covariances"]
    assert cov and all(r["status_hip"] == r["status_oracle"] for r in cov) and max(r["block_rel"] for r in cov) <= 1e-6
