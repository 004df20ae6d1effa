"""Per-iteration wall times of a config-#3 solve behind W warm-up iterations (dev check): python scripts/iter_times.py [W]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, "obvi-slam_amd", "python"), R]
import numpy as np, obvi_ba, synth, bench
cfg = bench.CONFIGS[3]
prob = synth.make_problem(P=cfg["P"], L=cfg["L"], O=cfg["O"], seed=20241008 + 3, const_poses=cfg["const_poses"], min_obj_obs=10)
ba = obvi_ba.BundleAdjuster(device_id=0)
synth.upload(ba, prob)
ba.evaluate(True, False); ba.evaluate(True, False)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if W > 0: ba.solve(bench.solver_params(obvi_ba, W))
t0 = time.perf_counter()
s = ba.solve(bench.solver_params(obvi_ba, 16))
dt = time.perf_counter() - t0
its = ba.iterations()
print("W=%d: %.3f ms/step; per iteration ms:" % (W, 1e3 * dt / 16), " ".join("%.2f" % (1e3 * it.iteration_time_in_seconds) for it in its))
print("ok flags:", "".join(str(int(it.step_is_successful)) for it in its))
