"""LM iteration of a pose-graph + objects stage (relative-pose, bounding-box and prior factors only; no reprojection factors): wall time per
iteration and the level plan (OBVI_DEBUG_PLAN=1).  usage: python scripts/pgo_iter.py [P O]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, "obvi-slam_amd", "python")]
import numpy as np, obvi_ba, synth
P, O = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (300, 20)
prob = synth.make_problem(P=P, L=max(1000, 10 * P), O=O, seed=7, const_poses=1, min_obj_obs=10)
g = obvi_ba.BundleAdjuster(device_id=0)
synth.upload(g, prob, reproj=False)
def prm(n): return obvi_ba.SolverParams(max_num_iterations=n, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0,
                                        parameter_tolerance=0.0, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
g.solve(prm(5))
synth.upload(g, prob, reproj=False)
t = time.time(); s = g.solve(prm(100)); dt = time.time() - t
print("P=%d O=%d: %d iterations, %.3f ms per iteration (wall)" % (P, O, s.num_iterations, 1e3 * dt / max(1, s.num_iterations)))
print({k: int(v) for k, v in g.problem_stats().items()})
