#!/usr/bin/env python3
"""Build container only: the VALUES of two of the reference's parameter files (config/base7a_2_fallback.json: schema 12, the values SURVEY 5.6 quotes;
config/update_revision_base.json: schema 14, the one version the reference's reader accepts) for the entries that reach the optimisation path, re-emitted as
tests/golden/config_*.json in the same nesting.  Data only: numbers, flags, class names.  usage: python tests/golden/gen_config_fixtures.py"""
import json, os
REF = "/root/reference/config"
HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = {
    "config_schema_version": None, "config_version_id": None,
    "visual_feature_params": ["reprojection_error_std_dev"],
    "local_ba_iteration_params": None, "global_ba_iteration_params": None, "final_ba_iteration_params": None,
    "pgo_solver_params": None, "ltm_tunable_params": None, "shape_dimension_priors": None,
    "bounding_box_front_end_params": ["post_session_object_merge_params"],
    "sliding_window_params": None, "optimization_factors_enabled_params": None, "object_visual_pose_graph_residual_params": None, "limit_traj_eval_params": None,
}
for name in ("base7a_2_fallback", "update_revision_base"):
    src = json.load(open(os.path.join(REF, name + ".json")))["config"]
    out = {}
    for key, sub in KEEP.items():
        out[key] = src[key] if sub is None else {k: src[key][k] for k in sub}
    with open(os.path.join(HERE, "config_%s.json" % name), "w") as f:
        json.dump({"config": out}, f, indent=1)
    print(name, "schema", out["config_schema_version"], len(json.dumps(out)), "bytes")
