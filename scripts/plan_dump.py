#!/usr/bin/env python3
"""Prints the level plan of the tile Cholesky for a bench configuration (OBVI_DEBUG_PLAN=1 makes prepare() list nodes and levels on stderr).
usage: OBVI_DEBUG_PLAN=1 python scripts/plan_dump.py [P L O]      |  ... plan_dump.py sessions S   (S config-5 sessions fused into one problem)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import obvi_ba, synth
if len(sys.argv) > 2 and sys.argv[1] == "sessions":
    prob = synth.join_problems(synth.make_sessions(int(sys.argv[2]), 500, 50000, 200, 20241008 + 5, 20241008 + 5, const_poses=1, min_obj_obs=10))
else:
    P, L, O = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2000, 300000, 200)
    prob = synth.make_problem(P=P, L=L, O=O, seed=20241008 + 3, const_poses=1, min_obj_obs=10)
g = obvi_ba.BundleAdjuster(device_id=0)
synth.upload(g, prob)
g.evaluate(True, False)
print({k: int(v) for k, v in g.problem_stats().items()})
