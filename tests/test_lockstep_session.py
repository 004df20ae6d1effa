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
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(helpers.ROOT, "include"), "-I" + HOST, "-include", srcs[2], "-o", LOCKSTEP_DRIVER, srcs[0], obj,
                               "-L" + os.path.dirname(helpers.PRODUCT_LIB), "-lobvi_ba", "-L" + os.path.join(helpers.ROOT, "oracle"), "-lobvi_oracle",
                               "-Wl,-rpath,$ORIGIN/../obvi-slam_amd/csrc", "-Wl,-rpath,$ORIGIN/../oracle"])
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
    assert len(solves) == len(json.load(open(out))["records"]) >= 150
    busy = [r for r in solves if r["initial_cost"] > 1e-3]
    # the same LM run in every optimisation: iteration count, accept / reject sequence, termination, reduced program
    bad = [r for r in solves if not (r["iterations_hip"] == r["iterations_oracle"] and r["termination_hip"] == r["termination_oracle"] and r["same_accept_sequence"] == 1 and r["params_reduced_equal"] == 1)]
    worst = {k: max(r[k] for r in busy) for k in ("initial_cost_rel", "final_cost_rel", "max_iteration_cost_rel", "pose_diff", "point_diff", "object_diff")}
    print("lock-step session (%s): %d optimisations, %d with a different LM run; worst %s" % ("deterministic" if deterministic else "default", len(solves), len(bad), {k: "%.1e" % v for k, v in worst.items()}))
    assert not bad, bad[:3]
    assert worst["initial_cost_rel"] <= 1e-11                                                    # the same objective at the same point
    assert worst["final_cost_rel"] <= 1e-7 and worst["max_iteration_cost_rel"] <= 1e-7           # stated tolerance of the LM end state: 1e-8 on well-conditioned windows
    assert worst["pose_diff"] <= 1e-6 and worst["object_diff"] <= 1e-5 and worst["point_diff"] <= 1e-4   # m / rad; a far feature moves along its ray for nothing
    # identical outlier selections (two-phase cut), evaluations and covariance blocks
    sel = [r for r in recs if r["call"] == "select_outliers"]
    assert len(sel) >= 100 and all(r["masks_differ"] == 0 and r["excluded_hip"] == r["excluded_oracle"] for r in sel)
    ev = [r for r in recs if r["call"] == "evaluate"]
    assert ev and max(r["cost_rel"] for r in ev) <= 1e-11 and max(r["sqnorm_rel"] for r in ev) <= 1e-11
    cov = [r for r in recs if r["call"] == "object_covariances"]
    assert cov and all(r["status_hip"] == r["status_oracle"] for r in cov) and max(r["block_rel"] for r in cov) <= 1e-6
