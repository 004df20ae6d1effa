#!/usr/bin/env python
"""The lock-step session of tests/test_lockstep_session.py with a THIRD backend: the extended-precision arbiter
(oracle/libobvi_oracle_ld.so, loaded by tests/lockstep_shim.cpp when OBVI_LOCKSTEP_ARBITER is set).  Every optimisation of the 80-frame
two-phase session starts from the same values on HIP (deterministic handle), oracle and arbiter; where HIP and the oracle do NOT run the
same LM sequence (about 9 of 165 solves), the table says how long each follows the arbiter and where each ends relative to it.
Writes gpurun_out/arbiter_session.json; prints the table (committed as profiles/r04_arbiter_session.txt)."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python"))
import numpy as np  # noqa: E402
import scene_io  # noqa: E402
import synth  # noqa: E402


def main():
    driver = os.path.join(ROOT, "tests", "run_offline_ba_lockstep")
    arb = os.path.join(ROOT, "oracle", "libobvi_oracle_ld.so")
    for f in (driver, arb):
        if not os.path.exists(f):
            raise SystemExit("%s missing: python -c 'import __graft_entry__ as g; g.build()'" % f)
    prob = synth.make_problem(P=80, L=1500, O=4, seed=21, min_obj_obs=12, bbox_noise=5.0, object_classes=("bench", "trashcan"), stereo=True)   # the scene of tests/test_host_mirror.py
    modes = sys.argv[1:] or ["--deterministic"]
    report = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "scene.txt")
        scene_io.write_scene(prob, path)
        for mode in modes:
            out, log = os.path.join(td, "out.json"), os.path.join(td, "lock.jsonl")
            cmd = [driver, path, out, "--window", "20", "--gba-frequency", "25", "--ltm"] + ([mode] if mode.startswith("--") else [])
            subprocess.check_call(cmd, timeout=3000, env=dict(os.environ, OBVI_LOCKSTEP_LOG=log, OBVI_LOCKSTEP_ARBITER=arb))
            recs = [json.loads(ln) for ln in open(log)]
            pairs, last = [], None
            for r in recs:
                if r["call"] == "solve":
                    last = r
                elif r["call"] == "arbiter" and last is not None:
                    pairs.append((last, r)); last = None
            busy = [(s, a) for s, a in pairs if s["initial_cost"] > 1e-3]
            same = lambda s: s["iterations_hip"] == s["iterations_oracle"] and s["same_accept_sequence"] == 1 and s["termination_hip"] == s["termination_oracle"]   # noqa: E731
            apart = [(s, a) for s, a in busy if not same(s)]
            together = [(s, a) for s, a in busy if same(s)]
            print("mode %s: %d solves with work; HIP and oracle run the same LM sequence in %d, part ways in %d" % (mode, len(busy), len(together), len(apart)))
            print("  where they run together: |final cost - arbiter| / cost  median HIP %.1e oracle %.1e, worst HIP %.1e oracle %.1e; poses median HIP %.1e oracle %.1e" % (
                np.median([a["hip_final_cost_rel"] for _, a in together]), np.median([a["oracle_final_cost_rel"] for _, a in together]),
                max(a["hip_final_cost_rel"] for _, a in together), max(a["oracle_final_cost_rel"] for _, a in together),
                np.median([a["hip_pose_diff"] for _, a in together]), np.median([a["oracle_pose_diff"] for _, a in together])))
            print("  where they part ways:  iterations HIP / oracle / arbiter | follows the arbiter for (HIP / oracle) iterations | final cost vs arbiter (HIP / oracle) | poses vs arbiter (HIP / oracle)")
            hip_closer = 0
            for s, a in apart:
                print("    %4d / %4d / %4d   |  %4d / %4d  |  %.1e / %.1e  |  %.1e / %.1e" % (a["iterations_hip"], a["iterations_oracle"], a["iterations_arbiter"], a["hip_follows"], a["oracle_follows"],
                                                                                           a["hip_final_cost_rel"], a["oracle_final_cost_rel"], a["hip_pose_diff"], a["oracle_pose_diff"]))
                hip_closer += a["hip_final_cost_rel"] <= a["oracle_final_cost_rel"]
            print("  HIP ends closer to the arbiter than the oracle does in %d of these %d; follows it longer in %d, shorter in %d" % (
                hip_closer, len(apart), sum(a["hip_follows"] > a["oracle_follows"] for _, a in apart), sum(a["hip_follows"] < a["oracle_follows"] for _, a in apart)))
            report[mode] = {"busy": len(busy), "together": len(together), "apart": [dict(a) for _, a in apart],
                            "together_median": {"hip_final_cost_rel": float(np.median([a["hip_final_cost_rel"] for _, a in together])), "oracle_final_cost_rel": float(np.median([a["oracle_final_cost_rel"] for _, a in together]))}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "arbiter_session.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
