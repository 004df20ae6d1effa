#!/usr/bin/env python3
"""Generates tests/golden/config3w_end_state.npz: the CPU ORACLE's end state on "config 3w" -- BASELINE config #3's sizes (2 000 keyframes / 300 000 features /
200 objects) as a WELL-POSED problem -- through the reference's two-phase global-BA block (offline_problem_runner.h:541-894; values of
config/base7a_2_fallback.json), then `POLISH` more iterations with zero function tolerance.  tests/test_gpu_end_state.py holds both HIP modes to it at
BASELINE.md 2.4 (iii)'s bar (final cost 1e-6 relative, poses 1e-6 m / rad).

Config 3w = synth.config3w(): config #3's generator with (a) five constant poses + the odometry factors of all consecutive frames (scale gauge fixed everywhere),
(b) only features whose first and last ray meet at >= 3 degrees (the camera looks along the direction of travel: a feature straight ahead has no observable depth),
(c) every feature / object initialised relative to the ESTIMATED pose of its anchor frame, >= 0.5 m in front of every observing camera, (d) one shape class with
distinct horizontal axes (the yaw of an ellipsoid with dx = dy is unobservable), (e) a stereo rig (second camera 0.12 m to the right: metric scale at every frame).
Which of these it takes was measured, not assumed (scripts/r06_wellposed_variants.py, profiles/r06_end_state_config3w.txt: the distance between the library's default
and deterministic handles, two round-off realisations of one algorithm, after the block): without (e) the 2 000-keyframe chain still ends 1e-5 apart in cost.  The oracle's result does not depend on its thread count (round 6:
tests/test_oracle_solver.py::test_the_oracle_is_bit_identical_for_any_number_of_host_threads), so any box generates the same file up to libm's last bits.

usage: python tests/golden/gen_config3w_end_state.py [threads] ; about 10 minutes on 8 threads.  Stores: poses, objects, every 100th feature, costs, LM
sequences, sizes and hashes of the excluded sets -- data only."""
import ctypes, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import end_state, obvi_ba, synth

POLISH = 20
threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
small = os.environ.get("OBVI_CONFIG3W_SMALL")      # "P,L,O": a small instance with the same recipe (tests of the recipe itself)
lib = os.path.join(ROOT, "oracle", "libobvi_oracle.so")
ctypes.CDLL(lib).oracle_set_threads(ctypes.c_int32(threads))
prob = synth.config3w(*(int(x) for x in small.split(","))) if small else synth.config3w()
print(synth.problem_stats(prob), flush=True)
ba = obvi_ba.BundleAdjuster(library=lib, prefix="oracle_")
t0 = time.time()
r = end_state.run_two_phase(ba, prob, obvi_ba, synth, block=end_state.GLOBAL_BA, polish_iterations=POLISH)
print("%.0f s | phase I %d it -> %.12g | phase II %d it -> %.12g | polish %d it -> %.12g" % (time.time() - t0, r["phase_1"]["iterations"], r["phase_1"]["final_cost"],
      r["phase_2"]["iterations"], r["phase_2"]["final_cost"], r["polish"]["iterations"], r["polish"]["final_cost"]), flush=True)
out = dict(polish_iterations=POLISH, stats=np.array([len(prob["poses"]), len(prob["points"]), len(prob["objects"]), len(prob["rp_pose"]), len(prob["bb_obj"])]))
for ph in ("phase_1", "phase_2", "polish"):
    out[ph + "_iterations"] = r[ph]["iterations"]; out[ph + "_initial_cost"] = r[ph]["initial_cost"]; out[ph + "_final_cost"] = r[ph]["final_cost"]
    out[ph + "_accepted"] = np.array(r[ph]["accepted"], np.uint8)
for t, m in r["excluded"].items():
    out["excluded_%d_count" % t] = int((np.asarray(m) == 0).sum())
    out["excluded_%d_sha256" % t] = hashlib.sha256(np.asarray(m, np.uint8).tobytes()).hexdigest()
    out["excluded_%d_bits" % t] = np.packbits(np.asarray(m, np.uint8))          # the mask itself (1 = kept), so that a differing set can be counted
out["phase_1_state_poses"] = r["state_1"]["poses"]                            # where phase I stopped (the state the cut is taken at)
for st in ("state_2", "state_polished"):
    out[st + "_poses"] = r[st]["poses"]; out[st + "_objects"] = r[st]["objects"]; out[st + "_points_every_100th"] = r[st]["points"][::100]
name = "config3w_end_state.npz" if not small else "config3w_small_end_state.npz"
np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), **out)
print("wrote", name)
