"""CPU: the oracle (and the product's host-compiled factor arithmetic) against the committed golden vectors."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731


@pytest.fixture(scope="module")
def oracle():
    lib = C.CDLL(helpers.ensure_oracle())
    lib.oracle_ellipsoid_corners.restype = C.c_int
    return lib


@pytest.fixture(scope="module")
def hostmath():
    so = os.path.join(helpers.ROOT, "tests", "libhostmath.so")
    src = os.path.join(helpers.ROOT, "tests", "hostmath_shim.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.hostmath_bbox.restype = C.c_int
    return lib


def _reproj(lib, fn, c):
    a = {k: np.array(c[k], dtype=np.float64) for k in ("pose", "point", "K", "ext", "pixel")}
    r, Jp, Jl = np.zeros(2), np.zeros(12), np.zeros(6)
    getattr(lib, fn)(dp(a["pose"]), dp(a["point"]), dp(a["K"]), dp(a["ext"]), dp(a["pixel"]), C.c_double(c["sigma"]), dp(r), dp(Jp), dp(Jl))
    return r, Jp.reshape(2, 6), Jl.reshape(2, 3)


def test_reference_tuple_reprojection(oracle, hostmath):
    t = json.load(open(os.path.join(GOLD, "reference_tuples.json")))["reprojection"]
    c = dict(pose=t["pose_t_aa"], point=t["point"], K=t["K"], ext=t["ext_qxyzw_t"], pixel=t["pixel"], sigma=t["sigma"])
    for lib, fn in ((oracle, "oracle_reproj"), (hostmath, "hostmath_reproj")):
        r, _, _ = _reproj(lib, fn, c)
        assert helpers.rel_err(r, t["residual"]) < t["residual_rel_tol"]


# kinds of tests/golden/a2_analytic_jacobian.json whose camera depth lies within a few thousand ulps of the clamp: there the depth itself
# (a difference of O(1) numbers) carries a relative rounding error of 1e-16 / |z|, and so does everything divided by it
A2_TOL = {"depth_1e-06": 1e-8, "depth_0.001": 1e-11, "depth_5e-16": 1e-11, "depth_-1e-12": 1e-11, "depth_0": 1e-11}
A2_SKIP = ("depth_2e-15", "depth_1e-12")   # z = 2e-15 / 1e-12 known to 5 % / 1e-4 only: which side of 1e-15 it falls on is rounding


def a2_cases():
    return json.load(open(os.path.join(GOLD, "a2_analytic_jacobian.json")))["cases"]


def test_analytic_jacobian_functor_against_the_reference_outputs(oracle, hostmath):
    """SURVEY row a2 (== a1): ReprojectionCostFunctorAnalyticJacobian::Evaluate.  The vectors are outputs of the reference's own function
    body (tests/golden/gen_a2_vectors.py): residual and both Jacobians to 1e-12 relative, for the oracle's restatement of
    reprojection_cost_functor_analytic_jacobian.h:63-70, 160, 289-292 and for the product's arithmetic (host build of csrc/ba_math.h with
    reprojection_variant = OBVI_REPROJECTION_ANALYTIC): generic cases, rotation vectors down to exactly zero (no constant branch), points behind
    the camera / on the image plane (clamped depth, gated derivative)."""
    seen = set()
    for c in a2_cases():
        if c["kind"] in A2_SKIP:
            continue
        seen.add(c["kind"].split("_")[0])
        tol = A2_TOL.get(c["kind"], 1e-12)
        cc = dict(pose=c["pose_t_aa"], point=c["point"], K=c["K"], ext=c["ext_qxyzw_t"], pixel=c["pixel"], sigma=c["sigma"])
        for lib, fn in ((oracle, "oracle_reproj_analytic"), (hostmath, "hostmath_reproj_analytic")):
            r, Jp, Jl = _reproj(lib, fn, cc)
            assert helpers.rel_err(r, c["residual"]) < tol, (fn, c["kind"])
            assert helpers.rel_err(Jp.ravel(), c["J_pose_2x6"]) < tol, (fn, c["kind"])
            assert helpers.rel_err(Jl.ravel(), c["J_point_2x3"]) < tol, (fn, c["kind"])
    assert seen >= {"survey", "generic", "small", "depth", "exact"}


def test_analytic_and_production_functors_differ_where_the_reference_says(oracle, hostmath):
    """a2 vs a3 on the same inputs: equal (1e-12) away from the two places the functors differ; below |aa| = 1e-8 the production functor's rotation is
    a constant (zero d r / d aa, vslam_math_util.h:363-369) and the analytic one's is not; behind the camera the production functor
    divides by the negative depth and the analytic one by 1e-15."""
    for c in a2_cases():
        cc = dict(pose=c["pose_t_aa"], point=c["point"], K=c["K"], ext=c["ext_qxyzw_t"], pixel=c["pixel"], sigma=c["sigma"])
        for lib, pre in ((oracle, "oracle_reproj"), (hostmath, "hostmath_reproj")):
            r3, Jp3, Jl3 = _reproj(lib, pre, cc)
            r2, Jp2, Jl2 = _reproj(lib, pre + "_analytic", cc)
            aa = np.linalg.norm(c["pose_t_aa"][3:])
            if c["kind"] in ("generic", "survey_tuple") or (c["kind"].startswith("small_angle") and aa > 1e-8):
                tol = 1e-12 if aa >= 1e-2 or pre == "hostmath_reproj" else 1e-9   # the dual-number path through aa/|aa| loses digits at tiny angles
                assert helpers.rel_err(r2, r3) < 1e-12 and helpers.rel_err(Jp2, Jp3) < tol and helpers.rel_err(Jl2, Jl3) < 1e-12, (pre, c["kind"])
            elif c["kind"].startswith("small_angle"):
                assert helpers.rel_err(r2, r3) < 1e-6 and np.all(Jp3[:, 3:] == 0.0) and np.abs(Jp2[:, 3:]).max() > 1.0, (pre, c["kind"])
            elif c["kind"] in ("depth_-5", "depth_-0.001", "exact_depth_-2"):
                assert np.abs(r2).max() > 1e12 and np.abs(r3).max() < 1e9, (pre, c["kind"])


def test_reference_tuple_bbox(oracle):
    t = json.load(open(os.path.join(GOLD, "reference_tuples.json")))["ellipsoid_bbox"]
    c = np.zeros(4)
    ok = oracle.oracle_ellipsoid_corners(dp(np.array(t["ellipsoid"])), dp(np.array(t["pose_t_aa"])), dp(np.array(t["K"])), dp(np.array(t["ext_qxyzw_t"])), dp(c))
    assert ok == 1
    assert np.abs(c - np.array(t["rectified_corners"])).max() < t["corners_abs_tol"]


def test_reprojection_vs_numpy_and_mpmath(oracle, hostmath):
    """Residuals to 1e-12 relative; Jacobians: the closed-form (product) path to 1e-12 everywhere, the
    dual-number (oracle == how the reference differentiates) path to 1e-12 for |aa| >= 1e-2 -- autodiff
    through aa/|aa| loses digits for tiny angles (measured 3e-11 at |aa| ~ 1e-6)."""
    cases = json.load(open(os.path.join(GOLD, "reproj_numpy.json")))
    assert len(cases) >= 64
    for c in cases:
        for lib, fn, jtol in ((oracle, "oracle_reproj", 2e-11), (hostmath, "hostmath_reproj", 1e-12)):
            r, Jp, Jl = _reproj(lib, fn, c)
            scale = max(1.0, np.abs(c["residual"]).max())
            assert np.abs(r - np.array(c["residual"])).max() / scale < 1e-12
            if c.get("zero_rotation_jacobian"):
                assert np.all(Jp[:, 3:] == 0.0)      # small-angle branch is a constant rotation: zero derivative
                continue
            assert helpers.rel_err(Jp, c["J_pose"]) < jtol
            assert helpers.rel_err(Jl, c["J_point"]) < 1e-12


def test_product_vs_oracle_jacobian_generic_angles(oracle, hostmath):
    cases = json.load(open(os.path.join(GOLD, "reproj_numpy.json")))
    for c in cases:
        if c.get("zero_rotation_jacobian") or np.linalg.norm(c["pose"][3:]) < 1e-2:
            continue
        _, Jo, Lo = _reproj(oracle, "oracle_reproj", c)
        _, Jh, Lh = _reproj(hostmath, "hostmath_reproj", c)
        assert helpers.rel_err(Jh, Jo) < 1e-12 and helpers.rel_err(Lh, Lo) < 1e-12


def test_bbox_vs_numpy(oracle, hostmath):
    cases = json.load(open(os.path.join(GOLD, "bbox_numpy.json")))
    n_invalid = 0
    for c in cases:
        ell, pose, K, ext = (np.array(c[k], dtype=np.float64) for k in ("ellipsoid", "pose", "K", "ext"))
        out = np.zeros(4)
        ok = oracle.oracle_ellipsoid_corners(dp(ell), dp(pose), dp(K), dp(ext), dp(out))
        rect = np.zeros(4) if c.get("invalid") else np.array(c["rectified_corners"])
        si = np.ascontiguousarray(np.diag([K[0], K[0], K[1], K[1]]) / 30.0)
        r, Je, Jp = np.zeros(4), np.zeros(28), np.zeros(24)
        okh = hostmath.hostmath_bbox(dp(ell), dp(pose), dp(K), dp(ext), dp(rect), dp(si), C.c_double(1000.0), dp(r), dp(Je), dp(Jp))
        if c.get("invalid"):
            n_invalid += 1
            assert ok == 0 and okh == 0
            assert np.all(r == 1000.0) and np.all(Je == 0.0) and np.all(Jp == 0.0)   # bounding_box_factor.h:81-96
            continue
        assert ok == 1 and okh == 1
        assert np.abs(out - rect).max() < 1e-12
        assert np.abs(r).max() < 1e-9            # predicted == observed -> zero residual
    assert n_invalid == 1


def test_huber(oracle):
    for c in json.load(open(os.path.join(GOLD, "huber.json"))):
        rho = np.zeros(3)
        oracle.oracle_huber(C.c_double(c["s"]), C.c_double(c["a"]), dp(rho))
        assert np.allclose(rho, c["rho"], rtol=1e-15, atol=0)


def test_spd_inverse_sqrt(oracle):
    rng = np.random.default_rng(5)
    oracle.oracle_spd_inverse_sqrt.restype = C.c_int
    for n in (3, 4, 6, 7):
        A = rng.normal(size=(n, n)); cov = A @ A.T + n * np.eye(n)
        out = np.zeros((n, n))
        assert oracle.oracle_spd_inverse_sqrt(dp(np.ascontiguousarray(cov)), C.c_int(n), dp(out)) == 1
        w, V = np.linalg.eigh(cov)
        ref = V @ np.diag(w ** -0.5) @ V.T
        assert helpers.rel_err(out, ref) < 1e-12
        assert helpers.rel_err(out @ out @ cov, np.eye(n)) < 1e-11
    bad = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert oracle.oracle_spd_inverse_sqrt(dp(bad), C.c_int(2), dp(np.zeros((2, 2)))) == 0


def test_mini_ba_regression():
    """The oracle reproduces its committed LM trajectory (guards the checker against accidental change)."""
    import synth
    g = json.load(open(os.path.join(GOLD, "mini_ba.json")))
    prob = {k: np.array(v) if isinstance(v, list) else v for k, v in g["problem"].items()}
    o = helpers.oracle_ba()
    synth.upload(o, prob)
    assert abs(o.evaluate(True)[0] - g["cost_robust"]) <= 1e-12 * g["cost_robust"]
    assert abs(o.evaluate(False)[0] - g["cost_raw"]) <= 1e-12 * g["cost_raw"]
    s = o.solve(helpers.ba_params(**g["solver"]))
    its = o.iterations()
    assert s.num_iterations == g["num_iterations"] and len(its) == len(g["trace"])
    for it, ref in zip(its, g["trace"]):
        assert abs(it.cost - ref["cost"]) <= 1e-9 * ref["cost"]
    assert np.abs(o.get_poses() - np.array(g["final_poses"])).max() < 1e-8
