"""How long do obvi_ba_create / destroy take, and the first solve on a fresh handle against a reused one?  (GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import obvi_ba, synth
prob = synth.make_problem(P=100, L=2000, O=6, seed=5, const_poses=1, min_obj_obs=6, object_classes=("bench",))
prm = obvi_ba.SolverParams(max_num_iterations=5, allow_non_monotonic_steps=True, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
g0 = obvi_ba.BundleAdjuster(device_id=0); synth.upload(g0, prob, reproj=False); g0.solve(prm)
for rep in range(4):
    t0 = time.perf_counter(); g = obvi_ba.BundleAdjuster(device_id=0); t1 = time.perf_counter()
    synth.upload(g, prob, reproj=False); t2 = time.perf_counter(); s = g.solve(prm); t3 = time.perf_counter(); g.close(); t4 = time.perf_counter()
    synth.upload(g0, prob, reproj=False); t5 = time.perf_counter(); g0.solve(prm); t6 = time.perf_counter()
    print("create %.2f ms, upload %.2f, solve(%d its) %.2f, destroy %.2f | reused handle: upload %.2f solve %.2f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, s.num_iterations, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3))
