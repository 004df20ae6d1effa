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


def test_nine_parameter_ellipsoid_block(oracle, hostmath):
    """The unconstrained ellipsoid block (x y z ax ay az dx dy dz; vslam_obj_opt_types_refactor.h:15-21, ellipsoid_utils.h:217-229 `#else`).  The reference cannot
    compile that branch, so its pins are: (1) upright, it IS the reference's golden tuple #1; (2) tilted, an independent numpy restatement
    (synth.project_ellipsoids with scipy's rotation vector) and brute-force sampling of the ellipsoid's surface; (3) the product's Jacobian (host build of
    ba_math.h, 15 directions and the one-direction form of the lane-parallel kernels) against finite differences of that restatement, and -- upright --
    against the 7-block's columns; (4) VectorToAxisAngle's constant branch at or below 1e-8 (vslam_math_util.h:31-42): identity, zero derivative."""
    import synth
    from scipy.spatial.transform import Rotation as Rot
    hostmath.hostmath_bbox9.restype = C.c_int
    t = json.load(open(os.path.join(GOLD, "reference_tuples.json")))["ellipsoid_bbox"]
    e7 = np.array(t["ellipsoid"]); K, ext, pose = np.array(t["K"]), np.array(t["ext_qxyzw_t"]), np.array(t["pose_t_aa"])
    e9 = np.array([e7[0], e7[1], e7[2], 0.0, 0.0, e7[3], e7[4], e7[5], e7[6]])
    c7, c9 = np.zeros(4), np.zeros(4)
    assert oracle.oracle_ellipsoid_corners(dp(e7), dp(pose), dp(K), dp(ext), dp(c7)) == 1 and oracle.oracle_ellipsoid_corners9(dp(e9), dp(pose), dp(K), dp(ext), dp(c9)) == 1
    assert np.abs(c9 - c7).max() < 1e-14 and np.abs(c9 - np.array(t["rectified_corners"])).max() < t["corners_abs_tol"]          # (1)

    def rectified(px):
        return np.array([(px[0] - K[2]) / K[0], (px[1] - K[2]) / K[0], (px[2] - K[3]) / K[1], (px[3] - K[3]) / K[1]])
    rng = np.random.default_rng(7)
    si = np.ascontiguousarray(np.diag([K[0], K[0], K[1], K[1]]) / 30.0)
    for case in range(40):
        tilt = Rot.from_rotvec(rng.normal(size=3) * (0.6 if case % 2 else 0.05))
        aa = (Rot.from_rotvec([0, 0, e7[3] + rng.normal() * 0.5]) * tilt).as_rotvec()
        ell = np.concatenate([e7[:3] + rng.normal(size=3) * 0.3, aa, e7[4:] * (1 + rng.normal(size=3) * 0.2)])
        out = np.zeros(4)
        assert oracle.oracle_ellipsoid_corners9(dp(ell), dp(pose), dp(K), dp(ext), dp(out)) == 1
        px, valid, _ = synth.project_ellipsoids(ell[None], pose[None], K, ext)
        want = rectified(px[0])
        assert valid[0] and np.abs(out - want).max() < 1e-12                                                               # (2) numpy restatement
        if case < 4:                                                                                                       # (2) the surface itself
            u, v = np.meshgrid(np.linspace(0, np.pi, 700), np.linspace(0, 2 * np.pi, 1400))
            semi = np.sqrt((ell[6:] / 2) ** 2 + synth.DIM_REG)
            pts = np.stack([semi[0] * np.sin(u) * np.cos(v), semi[1] * np.sin(u) * np.sin(v), semi[2] * np.cos(u)], axis=-1).reshape(-1, 3)
            pw = pts @ Rot.from_rotvec(aa).as_matrix().T + ell[:3]
            pix, z = synth.project_points(np.repeat(pose[None], len(pw), 0), pw, K, ext)
            assert (z > 0).all()
            box = rectified(np.array([pix[:, 0].min(), pix[:, 0].max(), pix[:, 1].min(), pix[:, 1].max()]))
            assert np.abs(np.sort(out[:2]) - box[:2]).max() < 2e-5 and np.abs(np.sort(out[2:]) - box[2:]).max() < 2e-5
        r, Je, Jp = np.zeros(4), np.zeros(36), np.zeros(24)
        assert hostmath.hostmath_bbox9(dp(ell), dp(pose), dp(K), dp(ext), dp(want.copy()), dp(si), C.c_double(1000.0), dp(r), dp(Je), dp(Jp)) == 1
        assert np.abs(r).max() < 1e-9                                                                                      # product == restatement
        Je = Je.reshape(4, 9)
        for k in range(9):                                                                                                 # (3) finite differences
            h = 1e-6
            ep, em = ell.copy(), ell.copy(); ep[k] += h; em[k] -= h
            fd = si @ (rectified(synth.project_ellipsoids(ep[None], pose[None], K, ext)[0][0]) - rectified(synth.project_ellipsoids(em[None], pose[None], K, ext)[0][0])) / (2 * h)
            assert np.abs(fd - Je[:, k]).max() < 2e-6 * max(1.0, np.abs(fd).max()), (case, k)
    # (3) upright: the 7-block's columns
    r7, Je7, Jp7 = np.zeros(4), np.zeros(28), np.zeros(24); r9, Je9, Jp9 = np.zeros(4), np.zeros(36), np.zeros(24)
    rect = rectified(synth.project_ellipsoids(e7[None], pose[None], K, ext)[0][0]) + 0.01
    assert hostmath.hostmath_bbox(dp(e7), dp(pose), dp(K), dp(ext), dp(rect), dp(si), C.c_double(1000.0), dp(r7), dp(Je7), dp(Jp7)) == 1
    assert hostmath.hostmath_bbox9(dp(e9), dp(pose), dp(K), dp(ext), dp(rect), dp(si), C.c_double(1000.0), dp(r9), dp(Je9), dp(Jp9)) == 1
    assert np.abs(r9 - r7).max() < 1e-12 and np.abs(Jp9 - Jp7).max() < 1e-11
    assert np.abs(Je9.reshape(4, 9)[:, [0, 1, 2, 5, 6, 7, 8]] - Je7.reshape(4, 7)).max() < 1e-11
    # (4) the constant branch
    tiny = np.array([e7[0], e7[1], e7[2], 3e-9, -2e-9, 5e-9, e7[4], e7[5], e7[6]]); zero = tiny.copy(); zero[3:6] = 0.0
    ct, cz = np.zeros(4), np.zeros(4)
    assert oracle.oracle_ellipsoid_corners9(dp(tiny), dp(pose), dp(K), dp(ext), dp(ct)) == 1 and oracle.oracle_ellipsoid_corners9(dp(zero), dp(pose), dp(K), dp(ext), dp(cz)) == 1
    assert np.array_equal(ct, cz)
    assert hostmath.hostmath_bbox9(dp(tiny), dp(pose), dp(K), dp(ext), dp(rect), dp(si), C.c_double(1000.0), dp(r9), dp(Je9), dp(Jp9)) == 1
    assert np.all(Je9.reshape(4, 9)[:, 3:6] == 0.0)


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
