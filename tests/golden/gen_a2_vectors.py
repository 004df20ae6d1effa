#!/usr/bin/env python
"""Golden vectors for SURVEY row a2: ReprojectionCostFunctorAnalyticJacobian::Evaluate.

Runs ONLY in the build container (it needs /root/reference); the GPU box and the tests use the JSON it
writes.  The body of `Evaluate` (include/refactoring/factors/reprojection_cost_functor_analytic_jacobian.h,
the statements between the commented-out timer block and the closing `return true;`) is dependency-free
scalar C++ (std::pow / sqrt / sin / cos / max on double arrays) although the header around it needs
Ceres and Eigen.  This script cuts that statement range out of the reference file AT RUN TIME into a
scratch directory under /tmp, puts it behind a free-function signature whose arguments carry the five
members the constructor computes (reprojection_cost_functor_analytic_jacobian.cpp:9-18), compiles it
with g++ and evaluates it on seeded inputs.  No reference text is stored in this repository: the
committed artefacts are this recipe and the vectors (inputs and the reference's outputs), which are data.

The cases: generic poses / points, small rotation vectors (|aa| from 1e-3 down to exactly 0, where the
production functor a3 switches to its constant branch and a2 does not), and points whose camera depth
straddles the clamp `max(z, 1e-15)` (behind the camera, exactly on the image plane, barely in front).

    python tests/golden/gen_a2_vectors.py        # rewrites tests/golden/a2_analytic_jacobian.json
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference/include/refactoring/factors/reprojection_cost_functor_analytic_jacobian.h"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "a2_analytic_jacobian.json")

WRAP_HEAD = r"""
#include <algorithm>
#include <cmath>
extern "C" int a2_eval(const double* pose6, const double* point3, const double* cam_rel_bl_vec_,
                       double rect_feature_x_, double rect_feature_y_, double rectified_error_multiplier_x_,
                       double rectified_error_multiplier_y_, double* residuals, double* Jpose, double* Jpoint) {
  const double kEpsilon = 1e-15;
  const double* parameters[2] = {pose6, point3};
  double* jacobians[2] = {Jpose, Jpoint};
"""
WRAP_TAIL = "\n  return 1;\n}\n"


def build():
    lines = open(REF).read().split("\n")
    begin = next(i for i, s in enumerate(lines) if "const double *robot_pose_block = parameters[0];" in s)
    end = next(i for i in range(begin, len(lines)) if lines[i] == "    return true;")   # the function's own return (the early-out for a null `jacobians` is indented deeper)
    body = "\n".join(lines[begin:end])
    assert re.search(r"std::max<double>\(_tmp\d+, kEpsilon\)", body), "the depth clamp is expected inside the range"
    tmp = tempfile.mkdtemp(prefix="a2_ref_", dir="/tmp")
    src, so = os.path.join(tmp, "a2.cpp"), os.path.join(tmp, "a2.so")
    with open(src, "w") as f:
        f.write(WRAP_HEAD + body + WRAP_TAIL)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.a2_eval.restype = C.c_int
    lib.a2_eval.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 4 + [C.POINTER(C.c_double)] * 3
    return lib


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def evaluate(lib, pose, point, K, ext, pixel, sigma):
    pose, point, ext = (np.ascontiguousarray(v, dtype=np.float64) for v in (pose, point, ext))
    r, Jp, Jl = np.zeros(2), np.zeros(12), np.zeros(6)
    lib.a2_eval(dp(pose), dp(point), dp(ext), (pixel[0] - K[2]) / K[0], (pixel[1] - K[3]) / K[1], K[0] / sigma, K[1] / sigma, dp(r), dp(Jp), dp(Jl))
    return r, Jp, Jl


def rot(aa):
    th = np.linalg.norm(aa)
    if th == 0.0:
        return np.eye(3)
    k = aa / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def quat_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    lib = build()
    rng = np.random.default_rng(20240601)
    K = [525.0, 525.0, 319.5, 239.5]
    cases = []

    def add(kind, pose, point, ext, pixel, sigma):
        r, Jp, Jl = evaluate(lib, pose, point, K, ext, pixel, sigma)
        cases.append({"kind": kind, "K": K, "ext_qxyzw_t": [float(v) for v in ext], "pose_t_aa": [float(v) for v in pose], "point": [float(v) for v in point],
                      "pixel": [float(v) for v in pixel], "sigma": float(sigma), "residual": r.tolist(), "J_pose_2x6": Jp.tolist(), "J_point_2x3": Jl.tolist()})

    # the survey's tuple, now with its Jacobians
    add("survey_tuple", [1.0, 2.0, 0.5, 0.1, -0.2, 0.3], [6.0, 2.5, 0.7], [-0.5, 0.5, -0.5, 0.5, 0.1, 0.05, 0.3], [300.0, 200.0], 1.5)

    def random_ext():
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        return np.concatenate([q, rng.normal(scale=0.2, size=3)])

    def point_at_depth(pose, ext, z, xy_scale=0.3):
        """world point whose camera coordinates are (x, y, z)"""
        pc = np.array([rng.normal(scale=xy_scale), rng.normal(scale=xy_scale), z])
        Rr, Re = rot(np.asarray(pose[3:])), quat_rot(ext[:4])
        pr = Re @ pc + ext[4:]
        return Rr @ pr + np.asarray(pose[:3])

    for i in range(24):   # generic
        ext = random_ext() if i % 3 else np.array([-0.5, 0.5, -0.5, 0.5, 0.1, 0.05, 0.3])
        pose = np.concatenate([rng.normal(scale=2.0, size=3), rng.normal(scale=0.8, size=3)])
        add("generic", pose, point_at_depth(pose, ext, rng.uniform(1.0, 15.0), 1.5), ext, rng.uniform([0, 0], [640, 480]), rng.uniform(0.5, 3.0))
    for mag in (1e-3, 1e-5, 1e-7, 3e-8, 1e-8, 5e-9, 1e-10, 0.0):   # small rotation vectors: no constant branch in a2
        for _ in range(2):
            ext = np.array([-0.5, 0.5, -0.5, 0.5, 0.1, 0.05, 0.3])
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            pose = np.concatenate([rng.normal(scale=2.0, size=3), mag * d])
            add("small_angle_%g" % mag, pose, point_at_depth(pose, ext, rng.uniform(2.0, 10.0), 1.0), ext, rng.uniform([0, 0], [640, 480]), 1.0)
    ext = np.array([-0.5, 0.5, -0.5, 0.5, 0.0, 0.0, 0.0])
    for z in (-5.0, -1e-3, -1e-12, 0.0, 5e-16, 2e-15, 1e-12, 1e-6, 1e-3):   # the clamp max(z, 1e-15) and its gated derivative
        pose = np.concatenate([rng.normal(scale=1.0, size=3), rng.normal(scale=0.5, size=3)])
        add("depth_%g" % z, pose, point_at_depth(pose, ext, z, 1e-3 if abs(z) < 1e-9 else 0.3), ext, [320.0, 240.0], 1.0)
    # identity pose, camera frame == robot frame up to the optical rotation, point exactly on the image plane / behind it: exact clamp cases
    for z in (0.0, -2.0):
        pose = np.zeros(6)
        ext = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
        add("exact_depth_%g" % z, pose, [0.25, -0.125, z], ext, [320.0, 240.0], 2.0)

    doc = {"_provenance": "Outputs of the reference's own ReprojectionCostFunctorAnalyticJacobian::Evaluate body (SURVEY row a2; "
                          "include/refactoring/factors/reprojection_cost_functor_analytic_jacobian.h of /root/reference), compiled stand-alone in /tmp by "
                          "tests/golden/gen_a2_vectors.py (g++ -O1 -ffp-contract=off) in the build container.  Jacobians are the functor's layout: pose 2x6 row-major "
                          "[d/dt, d/daa], point 2x3 row-major.  Data only (inputs and the reference's outputs).",
           "kEpsilon": 1e-15, "cases": cases}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    sys.exit(main())
