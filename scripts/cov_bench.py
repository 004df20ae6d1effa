#!/usr/bin/env python3
"""Times obvi_ba_object_covariances on BASELINE config #3 (200 objects, own blocks) and, with --oracle, the CPU restatement
on a smaller problem of the same shape.  usage: python scripts/cov_bench.py [--oracle]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python"), os.path.join(ROOT, "tests")]
import numpy as np
import obvi_ba, synth

prob = synth.make_problem(P=2000, L=300000, O=200, seed=20241008, const_poses=1, min_obj_obs=10)
g = obvi_ba.BundleAdjuster(device_id=0)
synth.upload(g, prob)
g.solve(obvi_ba.SolverParams(max_num_iterations=10, allow_non_monotonic_steps=True, function_tolerance=1e-6, gradient_tolerance=1e-10,
                             parameter_tolerance=1e-8, initial_trust_region_radius=1e4, max_trust_region_radius=1e16))
ids = np.arange(len(prob["objects"]))
g.object_covariances(ids)
t = time.time(); n = 5
for _ in range(n):
    c = g.object_covariances(ids)
dt = (time.time() - t) / n
sd = np.sqrt(np.einsum("oii->oi", c))
print("covariances of %d objects: %.2f ms per call; median sigma xyz %.3g m, yaw %.3g rad, dims %.3g m" %
      (len(ids), dt * 1e3, np.median(sd[:, :3]), np.median(sd[:, 3]), np.median(sd[:, 4:])))
pairs = np.array([(a, b) for a in range(20) for b in range(a + 1, 20)])
t = time.time(); g.object_covariances(pairs[:, 0], pairs[:, 1]); print("190 cross blocks: %.2f ms" % ((time.time() - t) * 1e3))
if "--oracle" in sys.argv:
    import helpers
    small = synth.make_problem(P=500, L=50000, O=50, seed=3, const_poses=1, min_obj_obs=10)
    o = helpers.oracle_ba(); synth.upload(o, small)
    g2 = obvi_ba.BundleAdjuster(device_id=0); synth.upload(g2, small)
    ids2 = np.arange(len(small["objects"]))
    t = time.time(); co = o.object_covariances(ids2); to = time.time() - t
    g2.object_covariances(ids2); t = time.time(); cg = g2.object_covariances(ids2); tg = time.time() - t
    print("P=500 / 50 objects: oracle %.2f s, device %.2f ms, max relative difference %.2e" % (to, tg * 1e3, (np.abs(cg - co).max(axis=(1, 2)) / np.abs(co).max(axis=(1, 2))).max()))
