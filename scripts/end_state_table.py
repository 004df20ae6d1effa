#!/usr/bin/env python3
"""End state of the reference's two-phase optimisation on the HIP path (default and deterministic handle) against the CPU oracle, at
BASELINE sizes.  obvi-slam_amd/python/end_state.py has the procedure and the reasoning.
usage: python scripts/end_state_table.py [config=2|2o|3] [oracle_threads=20] [polish_iterations=200] [arbiter=0|1] [second_oracle=0|1] > profiles/r05_end_state_<config>.txt
  arbiter=1 adds the extended-precision build of the oracle (oracle/libobvi_oracle_ld.so) as a fourth run: its distance to each fp64 run says how much of
  their difference is the problem's conditioning
  2 : BASELINE config #2 (500 keyframes / 50 000 features, reprojection only, first 5 poses constant), local_ba block
  2o: BASELINE config #2 + objects (500 keyframes / 50 000 features / 50 objects, first 5 poses constant), local_ba block (50 it / 1e-3, 100 it / 1e-4)
  3 : BASELINE config #3 (2 000 / 300 000 / 200 objects), global_ba block (250 it / 1e-6 twice); the oracle needs ~2 s per LM step: minutes"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import end_state, obvi_ba, synth

which = sys.argv[1] if len(sys.argv) > 1 else "2o"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 20
polish = int(sys.argv[3]) if len(sys.argv) > 3 else 200
with_arbiter = len(sys.argv) > 4 and int(sys.argv[4]) != 0
if which == "2":
    prob = synth.make_problem(P=500, L=50000, O=0, seed=20241008 + 2, const_poses=5)
    block, name = end_state.LOCAL_BA, "config #2: 500 KF / 50k features, reprojection only, first 5 poses constant, local_ba_iteration_params"
elif which == "2o":
    prob = synth.make_problem(P=500, L=50000, O=50, seed=3, const_poses=5, min_obj_obs=10)
    block, name = end_state.LOCAL_BA, "config #2 + objects: 500 KF / 50k features / 50 objects, local_ba_iteration_params"
elif which == "3":
    prob = synth.make_problem(P=2000, L=300000, O=200, seed=20241008 + 3, const_poses=1, min_obj_obs=10)
    block, name = end_state.GLOBAL_BA, "config #3: 2000 KF / 300k features / 200 objects, global_ba_iteration_params"
else:
    P, L, O = (int(x) for x in which.split(","))
    prob = synth.make_problem(P=P, L=L, O=O, seed=3, const_poses=5, min_obj_obs=6)
    block, name = end_state.LOCAL_BA, "%d KF / %d features / %d objects, local_ba_iteration_params" % (P, L, O)
lib = os.path.join(ROOT, "oracle", "libobvi_oracle.so")
ctypes.CDLL(lib).oracle_set_threads(ctypes.c_int32(max(1, min(threads, os.cpu_count() or 1))))
legs = {}
runs = [("hip_default", lambda: obvi_ba.BundleAdjuster(device_id=0)), ("hip_deterministic", lambda: obvi_ba.BundleAdjuster(device_id=0, deterministic=True)),
        ("oracle", lambda: obvi_ba.BundleAdjuster(library=lib, prefix="oracle_"))]
if len(sys.argv) > 5 and int(sys.argv[5]) != 0:
    # the fp64 oracle once more on ONE THREAD FEWER: the same arithmetic in another summation order -- how far apart two runs of the CHECKER end
    import shutil, tempfile
    lib_b = os.path.join(tempfile.mkdtemp(), "libobvi_oracle_b.so")      # a second copy of the library: its thread count is a global of the library
    shutil.copy(lib, lib_b)
    ctypes.CDLL(lib_b).oracle_set_threads(ctypes.c_int32(max(1, min(threads, os.cpu_count() or 1) - 1)))
    runs.append(("oracle_one_thread_fewer", lambda: obvi_ba.BundleAdjuster(library=lib_b, prefix="oracle_")))
if with_arbiter:
    lib_ld = os.path.join(ROOT, "oracle", "libobvi_oracle_ld.so")
    ctypes.CDLL(lib_ld).oracle_set_threads(ctypes.c_int32(max(1, min(threads, os.cpu_count() or 1))))
    runs.append(("arbiter", lambda: obvi_ba.BundleAdjuster(library=lib_ld, prefix="oracle_")))
for leg, make in runs:
    ba = make()
    t0 = time.time()
    legs[leg] = end_state.run_two_phase(ba, prob, obvi_ba, synth, block=block, polish_iterations=polish)
    legs[leg]["seconds"] = time.time() - t0
    ba.close()
    r = legs[leg]
    print("# %-18s %.1f s | phase I %d it -> %.9g | phase II %d it -> %.12g (%s) | polish %d it -> %.12g (%s)" % (
        leg, r["seconds"], r["phase_1"]["iterations"], r["phase_1"]["final_cost"], r["phase_2"]["iterations"], r["phase_2"]["final_cost"], r["phase_2"]["message"][:40],
        r["polish"]["iterations"], r["polish"]["final_cost"], r["polish"]["message"][:40]), flush=True)
print("# " + name)
pairs = [("hip_default", "oracle"), ("hip_deterministic", "oracle"), ("hip_default", "hip_deterministic")]
if "oracle_one_thread_fewer" in legs:
    pairs += [("oracle_one_thread_fewer", "oracle"), ("hip_default", "oracle_one_thread_fewer")]
if with_arbiter:
    pairs += [("hip_default", "arbiter"), ("hip_deterministic", "arbiter"), ("oracle", "arbiter")]
for a, b in pairs:
    print(json.dumps({"pair": [a, b], **end_state.compare(legs[a], legs[b])}))
