#!/usr/bin/env python3
"""BASELINE config #5 on one GPU: sessions over the same place chained through the long-term map
(ltm_trajectory_sequence_executor.py:45-92 runs the reference's sessions one after the other in the same way): session s
= local-BA-sized problem (500 keyframes / 50 000 features) over one shared object set; it starts from the map of session
s-1 (ellipsoid estimates + marginal covariances as IndependentObjectMapFactor priors), runs the two-phase BA and ends by
extracting the new map on the device.  Prints per-session times and the map's error against the synthetic truth.
usage: python scripts/multi_session.py [sessions=16] [objects=200]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import obvi_ba, synth

n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_objects = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prm = obvi_ba.SolverParams(max_num_iterations=50, allow_non_monotonic_steps=True, function_tolerance=1e-4, gradient_tolerance=1e-10,
                           parameter_tolerance=1e-8, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
ltm = None   # (object ids, means, covariances)
g = obvi_ba.BundleAdjuster(device_id=0)
for s in range(n_sessions):
    t0 = time.time()
    prob = synth.make_problem(P=500, L=50000, O=n_objects, seed=1000 + s, object_seed=77, const_poses=1, min_obj_obs=10, object_classes=("bench",))
    if ltm is not None:
        prob["objects"][ltm[0]] = ltm[1]                          # a mapped object starts from the map
        prob.update(lt_obj=ltm[0].astype(np.uint32), lt_mean=ltm[1], lt_cov=ltm[2].reshape(-1, 49), lt_huber=1.0)
    t1 = time.time()
    synth.upload(g, prob)
    s1 = g.solve(prm)                                             # phase I
    mask, nex = g.select_outliers(0, 0.1)                         # phase II without the worst 10 % of the visual factors
    g.set_active_mask(0, mask)
    s2 = g.solve(prm)
    ids = np.arange(len(prob["objects"]), dtype=np.uint32)
    t2 = time.time()
    g.set_active_mask(0, np.ones_like(mask))                      # the extraction problem holds every factor again (a feature left with one sighting is rank deficient)
    cov = g.object_covariances(ids)
    t3 = time.time()
    est = g.get_objects()
    seen = np.abs(cov).max(axis=(1, 2)) > 0
    err = np.linalg.norm(est[seen, :3] - prob["gt_objects"][seen, :3], axis=1)
    sd = np.sqrt(np.einsum("oii->oi", cov[seen])[:, :3]).mean(axis=1)
    print("session %2d: %3d objects mapped | BA %5.1f ms (%2d + %2d iterations) | map extraction %5.2f ms | problem generation %4.0f ms | centre error median %.3f m, sigma median %.3f m"
          % (s, int(seen.sum()), (t2 - t1) * 1e3, s1.num_iterations, s2.num_iterations, (t3 - t2) * 1e3, (t1 - t0) * 1e3, np.median(err), np.median(sd)), flush=True)
    ltm = (ids[seen], est[seen], cov[seen])
