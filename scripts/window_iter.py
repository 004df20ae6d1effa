#!/usr/bin/env python3
"""LM iteration of a local-BA window (50 keyframes / 5 000 features / 5 objects, ~49 k sightings): wall time per iteration and the
phase split (profiling level 1).  usage: python scripts/window_iter.py [P L O]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import obvi_ba, synth
P, L, O = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50, 5000, 5)
prob = synth.make_problem(P=P, L=L, O=O, seed=5, const_poses=5, min_obj_obs=10)
g = obvi_ba.BundleAdjuster(device_id=0)
synth.upload(g, prob)
def prm(n): return obvi_ba.SolverParams(max_num_iterations=n, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0,
                                        parameter_tolerance=0.0, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
g.solve(prm(5))
for lvl in (0, 1):
    synth.upload(g, prob); g.solve(prm(1)); g.set_profiling(lvl)
    synth.upload(g, prob); g.solve(prm(1))
    t = time.time(); s = g.solve(prm(40)); dt = time.time() - t
    print("profiling %d: %d iterations, %.3f ms per iteration (wall)" % (lvl, s.num_iterations, 1e3 * dt / max(1, s.num_iterations)))
kt = g.kernel_times()
tot = sum(v[0] / max(1, v[1]) for v in kt.values())
print("  ".join("%s %.0f us" % (k, 1e3 * v[0] / max(1, v[1])) for k, v in kt.items()), " | sum %.0f us" % (1e3 * tot))
print({k: int(v) for k, v in g.problem_stats().items()})
