#!/usr/bin/env python3
"""ms per LM step of BASELINE config #3 with the reference's 7-parameter ellipsoid block and with the 9-parameter one (obvi_ba_options.object_block_size = 9;
the objects tilted by up to 0.2 rad): the same 30 steps at zero tolerances as bench.py times.  usage: python scripts/nine_dof_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import obvi_ba, synth

prob7 = synth.make_problem(P=2000, L=300000, O=200, seed=20241008 + 3, const_poses=1, min_obj_obs=10)
prob9 = synth.nine_dof(prob7, tilt=0.2, seed=1)
prm = lambda n: obvi_ba.SolverParams(max_num_iterations=n, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0,
                                     initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
for rep in range(2):
    for od, prob in ((7, prob7), (9, prob9)):
        ba = obvi_ba.BundleAdjuster(device_id=0, object_block_size=od)
        synth.upload(ba, prob)
        ba.solve(prm(3))
        synth.upload(ba, prob)
        ba.evaluate(True, False)          # (the symbolic phase of the re-uploaded problem, outside the timed solve)
        t0 = time.perf_counter(); s = ba.solve(prm(30)); dt = time.perf_counter() - t0
        st = ba.problem_stats()
        print("object block %d: %.4f ms per LM step (%d steps), reduced rows %d, final cost %.6e" % (od, 1e3 * dt / (s.num_iterations - 1), s.num_iterations - 1, st["reduced_rows"], s.final_cost), flush=True)
        ba.close()
