#!/usr/bin/env python
"""BASELINE config 3 end to end through the C++ host mirror (run_offline_ba --global-ba): writes the binary scene, runs the driver with
OBVI_HOST_TIMING=1 OBVI_API_TIMING=1, prints the driver's JSON line and both timing reports.  usage: scripts/e2e_cpp.py [P L O] [runs]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python"))
import scene_io, synth
P, L, O = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2000, 300000, 200)
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 2
prob = synth.make_problem(P=P, L=L, O=O, seed=20241008 + 3, const_poses=1, min_obj_obs=10)
with tempfile.TemporaryDirectory() as td:
    scene, out = os.path.join(td, "scene.bin"), os.path.join(td, "out.json")
    scene_io.write_scene_binary(prob, scene)
    for r in range(runs):
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "obvi-slam_amd", "host", "run_offline_ba"), scene, out, "--global-ba", "--merge-distance", "-1"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, OBVI_HOST_TIMING=os.environ.get("OBVI_HOST_TIMING", "1"), OBVI_API_TIMING="1"))
        wall = time.perf_counter() - t0
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        rep = json.loads(line[-1]) if line else {}
        print("run %d: process %.0f ms | scene load %.0f, pose graph %.0f, runFullOptimization %.0f ms | %s" % (
            r, 1e3 * wall, rep.get("scene_load_ms", -1), rep.get("pose_graph_ms", -1), rep.get("run_full_optimization_ms", -1),
            [(x["kind"], x["iterations"]) for x in rep.get("records", [])]))
    print(p.stderr[-6000:])
