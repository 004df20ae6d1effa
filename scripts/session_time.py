#!/usr/bin/env python3
"""Wall time of a sliding-window session through the C++ host mirror (run_offline_ba): per-frame two-phase local BA over a
window of 50 frames, global BA every 100 frames (config/base7a_2_fallback.json shapes), against the device time the solves
report.  usage: python scripts/session_time.py [P L O]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import synth, scene_io
P, L, O = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (300, 30000, 20)
prob = synth.make_problem(P=P, L=L, O=O, seed=4, min_obj_obs=10, bbox_noise=5.0, object_classes=("bench",))
d = tempfile.mkdtemp()
scene, out, csv = os.path.join(d, "scene.bin"), os.path.join(d, "out.json"), os.path.join(d, "opt.csv")
(scene_io.write_scene if os.environ.get("OBVI_SCENE_TEXT") else scene_io.write_scene_binary)(prob, scene)   # the binary form loads in ~5 ms, the text form in ~95
t = time.time()
subprocess.check_call([os.path.join(ROOT, "obvi-slam_amd", "host", "run_offline_ba"), scene, out, "--window", "50", "--gba-frequency", "100", "--csv", csv, "--merge-distance", "-1"])
wall = time.time() - t
res = json.load(open(out))
rows = [ln.split(",") for ln in open(csv).read().strip().split("\n")[1:]]
solver = sum(float(r[8]) for r in rows); its = sum(int(r[12]) for r in rows)
print("P=%d L=%d O=%d: %d optimisations, %d LM iterations; wall %.2f s, solver time (sum of total_time) %.2f s -> %.1f ms per optimisation, %.2f ms outside the solver per optimisation"
      % (P, L, O, len(rows), its, wall, solver, 1e3 * wall / len(rows), 1e3 * (wall - solver) / len(rows)))
