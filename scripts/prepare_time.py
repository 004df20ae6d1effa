#!/usr/bin/env python3
"""Host-side cost of a new problem: upload (set_*) and the symbolic phase (prepare(): ordering, visit lists, tile plan) against the
device time of an LM iteration.  usage: python scripts/prepare_time.py [P L O]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import obvi_ba, synth
P, L, O = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (500, 50000, 50)
prob = synth.make_problem(P=P, L=L, O=O, seed=5, const_poses=1, min_obj_obs=10)
g = obvi_ba.BundleAdjuster(device_id=0)
one = obvi_ba.SolverParams(max_num_iterations=1, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0,
                           parameter_tolerance=0.0, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
synth.upload(g, prob); g.solve(one)          # warm the device and the allocations
for rep in range(3):
    t0 = time.time(); synth.upload(g, prob); t1 = time.time()
    g.solve(one); t2 = time.time()
    g.solve(one); t3 = time.time()
    print("P=%d L=%d O=%d obs=%d: upload %.1f ms, first 1-iteration solve %.1f ms, second %.1f ms -> symbolic phase about %.1f ms"
          % (P, L, O, len(prob["rp_pose"]), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t2 - t1 - (t3 - t2)) * 1e3), flush=True)
