#!/usr/bin/env python
"""Generates the committed golden fixtures under tests/golden/.

Run from the repo root:  python tests/golden/gen_fixtures.py
Nothing here reads /root/reference.  Sources of truth:
  * reproj_numpy.json   residuals from the independent numpy/scipy restatement in
                        obvi-slam_amd/python/synth.py (project_points); Jacobians by 40-digit mpmath
                        differentiation of a third, mpmath restatement of the same formula.
  * bbox_numpy.json     rectified corners from synth.project_ellipsoids, cross-checked here by
                        brute-force sampling of the ellipsoid surface (max/min of projected points).
  * huber.json          rho(s), rho'(s) of ceres::HuberLoss [Ceres-doc] from the closed form.
  * mini_ba.json        a 6-pose / 40-point / 2-object problem with the ORACLE's own cost trace
                        (regression fixture: it pins the oracle against accidental change and gives the
                        HIP path a committed LM trajectory to reproduce; it is not an independent truth).
"""
import json
import os
import sys

import numpy as np
import mpmath as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from scipy.spatial.transform import Rotation as Rot  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 40


def mp_reproj(pose, X, K, ext, pix, sigma):
    """r(pose, X) in mpmath, returns function of the 9 free variables."""
    def f(*v):
        t = mp.matrix(v[0:3]); a = mp.matrix(v[3:6]); Xw = mp.matrix(v[6:9])
        th = mp.sqrt(a[0] ** 2 + a[1] ** 2 + a[2] ** 2)
        u = a / th
        c, s = mp.cos(th), mp.sin(th)
        Kx = mp.matrix([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        R = mp.eye(3) * c + (1 - c) * (u * u.T) + s * Kx
        pr = R.T * (Xw - t)
        q = [mp.mpf(x) for x in ext[:4]]
        n = mp.sqrt(sum(x * x for x in q)); x, y, z, w = [e / n for e in q]
        Re = mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        pc = Re.T * (pr - mp.matrix([mp.mpf(e) for e in ext[4:]]))
        return pc, None
    def r(i):
        def g(*v):
            pc, _ = f(*v)
            return (pc[i] / pc[2] - (mp.mpf(pix[i]) - mp.mpf(K[2 + i])) / mp.mpf(K[i])) * mp.mpf(K[i]) / mp.mpf(sigma)
        return g
    return r(0), r(1)


def gen_reproj(rng, n=64):
    K = synth.K_DEFAULT
    cases = []
    for i in range(n):
        ext = np.concatenate([rng.normal(size=4), rng.normal(size=3) * 0.2])
        scale = [1.0, 0.3, 1e-2, 2.5][i % 4]
        pose = np.concatenate([rng.normal(size=3) * 3, rng.normal(size=3) * scale])
        R = Rot.from_rotvec(pose[3:]).as_matrix()
        Re = synth.quat_to_R(ext[:4])
        depth = [rng.uniform(1, 20), 0.05][i % 16 == 15]          # every 16th case: z -> 0+
        pc = np.array([rng.uniform(-1, 1) * depth, rng.uniform(-1, 1) * depth, depth])
        X = R @ (Re @ pc + ext[4:]) + pose[:3]
        pix = rng.uniform(0, 600, size=2)
        sigma = float(rng.uniform(0.5, 3.0))
        px, _ = synth.project_points(pose[None], X[None], K, ext)
        r = (px[0] - pix) / sigma
        r0, r1 = mp_reproj(pose, X, K, ext, pix, sigma)
        v = tuple(mp.mpf(float(e)) for e in np.concatenate([pose, X]))
        J = np.zeros((2, 9))
        for a, fn in enumerate((r0, r1)):
            for k in range(9):
                J[a, k] = float(mp.diff(fn, v, tuple(1 if j == k else 0 for j in range(9))))
        r_mp = np.array([float(r0(*v)), float(r1(*v))])
        assert np.allclose(r, r_mp, rtol=1e-9, atol=1e-9), (r, r_mp)
        cases.append(dict(K=K.tolist(), ext=ext.tolist(), pose=pose.tolist(), point=X.tolist(), pixel=pix.tolist(), sigma=sigma,
                          residual=r_mp.tolist(), J_pose=J[:, :6].tolist(), J_point=J[:, 6:].tolist()))
    # the aa == 0 branch (vslam_math_util.h:363-369): identity rotation, zero derivative w.r.t. aa
    pose = np.array([1.0, 2.0, 3.0, 0.0, 0.0, 0.0]); X = np.array([8.0, 2.5, 3.3]); pix = np.array([300.0, 200.0])
    ext = synth.EXT_DEFAULT
    px, _ = synth.project_points(pose[None], X[None], K, ext)
    cases.append(dict(K=K.tolist(), ext=ext.tolist(), pose=pose.tolist(), point=X.tolist(), pixel=pix.tolist(), sigma=1.5,
                      residual=((px[0] - pix) / 1.5).tolist(), J_pose=None, J_point=None, zero_rotation_jacobian=True))
    return cases


def brute_force_bbox(ell, pose, K, ext, n=400):
    """Project a dense sampling of the ellipsoid surface with semi-axes sqrt((d/2)^2 + c)."""
    u, v = np.meshgrid(np.linspace(0, 2 * np.pi, n), np.linspace(0, np.pi, n))
    ax = np.sqrt((ell[4:7] / 2) ** 2 + synth.DIM_REG)
    pts = np.stack([ax[0] * np.cos(u) * np.sin(v), ax[1] * np.sin(u) * np.sin(v), ax[2] * np.cos(v)], axis=-1).reshape(-1, 3)
    Rz = Rot.from_euler("z", ell[3]).as_matrix()
    pw = pts @ Rz.T + ell[0:3]
    px, z = synth.project_points(np.repeat(pose[None], len(pw), 0), pw, K, ext)
    return np.array([px[:, 0].min(), px[:, 0].max(), px[:, 1].min(), px[:, 1].max()])


def gen_bbox(rng, n=24):
    K, ext = synth.K_DEFAULT, synth.EXT_DEFAULT
    cases = []
    while len(cases) < n:
        pose = np.concatenate([rng.normal(size=3), [0, 0, rng.uniform(-3, 3)]]) + np.concatenate([np.zeros(3), rng.normal(size=3) * 0.05])
        R = Rot.from_rotvec(pose[3:]).as_matrix()
        fwd = R @ np.array([1.0, 0, 0]); left = R @ np.array([0, 1.0, 0])
        d = rng.uniform(4, 15)
        centre = pose[:3] + fwd * d + left * rng.uniform(-0.3, 0.3) * d + np.array([0, 0, rng.uniform(-0.5, 0.5)])
        ell = np.concatenate([centre, [rng.uniform(-np.pi, np.pi)], rng.uniform(0.3, 2.5, size=3)])
        px, valid, depth = synth.project_ellipsoids(ell[None], pose[None], K, ext)
        if not valid[0] or depth[0] < 2:
            continue
        c = px[0]
        box = np.array([min(c[0], c[1]), max(c[0], c[1]), min(c[2], c[3]), max(c[2], c[3])])
        bf = brute_force_bbox(ell, pose, K, ext)
        assert np.abs(box - bf).max() < 0.05, (box, bf)       # sampling resolution
        rect = np.array([(c[0] - K[2]) / K[0], (c[1] - K[2]) / K[0], (c[2] - K[3]) / K[1], (c[3] - K[3]) / K[1]])
        cases.append(dict(K=K.tolist(), ext=ext.tolist(), pose=pose.tolist(), ellipsoid=ell.tolist(), rectified_corners=rect.tolist(),
                          brute_force_pixel_box=bf.tolist()))
    # invalid case: camera inside the ellipsoid -> both radicands <= 0 -> functor returns false
    pose = np.array([0.0, 0, 0, 0, 0, 0.3]); ell = np.array([0.2, 0.0, 0.0, 0.1, 4.0, 4.0, 4.0])
    px, valid, _ = synth.project_ellipsoids(ell[None], pose[None], K, ext)
    assert not valid[0]
    cases.append(dict(K=K.tolist(), ext=ext.tolist(), pose=pose.tolist(), ellipsoid=ell.tolist(), rectified_corners=None, invalid=True))
    return cases


def gen_huber():
    cases = []
    for a in (0.5, 1.0, 10.0):
        for s in (0.0, 0.1 * a * a, a * a, 1.0001 * a * a, 4 * a * a, 1e6):
            if s > a * a:
                r = np.sqrt(s); rho0 = 2 * a * r - a * a; rho1 = a / r; rho2 = -rho1 / (2 * s)
            else:
                rho0, rho1, rho2 = s, 1.0, 0.0
            cases.append(dict(a=a, s=s, rho=[rho0, rho1, rho2]))
    return cases


def gen_mini_ba():
    import helpers
    import obvi_ba
    prob = synth.make_problem(P=6, L=40, O=2, seed=42, min_obj_obs=3, object_classes=("bench",), bbox_noise=5.0)
    o = helpers.oracle_ba()
    synth.upload(o, prob)
    c_rob, _, _ = o.evaluate(True)
    c_raw, _, _ = o.evaluate(False)
    prm = helpers.ba_params(max_it=15)
    s = o.solve(prm)
    trace = [dict(iteration=it.iteration, cost=it.cost, step_norm=it.step_norm, successful=it.step_is_successful) for it in o.iterations()]
    keep = ["K", "ext", "poses", "pose_const", "points", "point_const", "objects", "object_const", "rp_pose", "rp_point", "rp_cam", "rp_pixel",
            "bb_obj", "bb_pose", "bb_cam", "bb_corners", "bb_cov", "sp_obj", "sp_mean", "sp_cov", "rl_a", "rl_b", "rl_t", "rl_aa", "rl_cov"]
    scal = ["rp_sigma", "rp_huber", "bb_huber", "bb_invalid", "sp_huber", "rl_huber"]
    return dict(problem={k: np.asarray(prob[k]).tolist() for k in keep} | {k: float(prob[k]) for k in scal},
                cost_robust=c_rob, cost_raw=c_raw, solver=dict(max_it=15, nonmono=True, ftol=1e-6, radius=100.0, max_radius=1e4),
                termination=s.message.decode(), num_iterations=s.num_iterations, final_cost=s.final_cost, trace=trace,
                final_poses=o.get_poses().tolist(), final_objects=o.get_objects().tolist())


def main():
    rng = np.random.Generator(np.random.MT19937(20241008))
    json.dump(gen_reproj(rng), open(os.path.join(OUT, "reproj_numpy.json"), "w"))
    json.dump(gen_bbox(rng), open(os.path.join(OUT, "bbox_numpy.json"), "w"))
    json.dump(gen_huber(), open(os.path.join(OUT, "huber.json"), "w"))
    json.dump(gen_mini_ba(), open(os.path.join(OUT, "mini_ba.json"), "w"))
    for f in ("reproj_numpy.json", "bbox_numpy.json", "huber.json", "mini_ba.json"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
