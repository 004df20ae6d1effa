#!/usr/bin/env python3
"""Which variant of a config-3-size problem do two fp64 realisations of the same algorithm (default handle: atomics; deterministic handle: fixed order) end at the same
point on?  The two differ by round-off only, so their distance after the reference's two-phase global-BA block is the decidability of the problem (no oracle needed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "obvi-slam_amd", "python")]
import numpy as np
import end_state, obvi_ba, synth

def variant(name):
    kw = dict(P=2000, L=300000, O=200, seed=20241008 + 3, const_poses=5, min_obj_obs=10, object_classes=("bench",), min_parallax_deg=3.0)
    if name == "3w": return synth.make_well_posed(synth.make_problem(**kw))
    if name == "stereo": return synth.make_well_posed(synth.make_problem(stereo=True, **kw))
    if name == "anchors100":
        q = synth.make_well_posed(synth.make_problem(**kw)); q["pose_const"] = q["pose_const"].copy(); q["pose_const"][::100] = 1
        q["poses"] = q["poses"].copy(); q["poses"][::100] = q["gt_poses"][::100]; return q
    if name == "no_outliers": return synth.make_well_posed(synth.make_problem(outlier_frac=0.0, **kw))
    if name == "P500":
        kw.update(P=500, L=75000, O=50); return synth.make_well_posed(synth.make_problem(**kw))
    if name == "P1000":
        kw.update(P=1000, L=150000, O=100); return synth.make_well_posed(synth.make_problem(**kw))
    raise SystemExit(name)

for name in sys.argv[1:]:
    prob = variant(name)
    legs = {}
    for leg, det in (("default", False), ("deterministic", True)):
        ba = obvi_ba.BundleAdjuster(device_id=0, deterministic=det)
        legs[leg] = end_state.run_two_phase(ba, prob, obvi_ba, synth, block=end_state.GLOBAL_BA, polish_iterations=20)
        ba.close()
    c = end_state.compare(legs["default"], legs["deterministic"])
    print("%-12s" % name, "N_r %d" % len(prob["rp_pose"]), "| I", c["phase_1"]["iterations"], "%.1e" % c["phase_1"]["final_cost_rel"], "| excl diff", c["excluded_differ_in"],
          "| II", c["phase_2"]["iterations"], "%.1e" % c["phase_2"]["final_cost_rel"], "| polish", c["polish"]["iterations"], "%.1e" % c["polish"]["final_cost_rel"],
          "| poses after II %.1e m %.1e rad, polished %.1e m" % (c["state_after_phase_2"]["pose_translation_max_m"], c["state_after_phase_2"]["pose_rotation_max_rad"], c["state_polished"]["pose_translation_max_m"]), flush=True)
