"""Pose-graph checkpoints: the reference stores a pose-graph state with cv::FileStorage
(object_and_reprojection_feature_pose_graph_file_storage_io.h) and replays it through run_opt_from_pg_state.  CPU: the reader /
writer of obvi-slam_amd/host/obvi_checkpoint_io.h on the state of the reference's own round-trip test, laid out the way OpenCV
writes JSON (tests/golden/gen_checkpoint_fixture.py).  GPU: a session's final checkpoint replayed the run_opt_from_pg_state way."""
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
import scene_io
import synth
from test_host_mirror import driver  # noqa: F401  (fixture)

GOLDEN = os.path.join(helpers.ROOT, "tests", "golden")


def _tolerant_json(path):
    """our writer prints plain JSON numbers; keep the door open for OpenCV's "4." spelling"""
    import re
    txt = open(path).read()
    txt = re.sub(r"(?<![\w.])(-?\d+)\.(?=[\s,\]}])", r"\1.0", txt)
    return json.loads(txt)


def _kv(seq):
    return {e["k"]: e["v"] for e in seq}


def _mat(m, rows, cols):
    assert (m["Rows"], m["Cols"]) == (rows, cols) and len(m["Data"]) == rows * cols
    return np.array(m["Data"], dtype=float)


def _pairs(seq):
    return sorted((int(e["f"]), int(e["s"])) for e in seq)


def test_reads_the_state_of_the_references_round_trip_test(driver, tmp_path):
    src = os.path.join(GOLDEN, "pose_graph_state_reference_roundtrip.json")
    exp = json.load(open(os.path.join(GOLDEN, "pose_graph_state_reference_roundtrip.expected.json")))
    out = str(tmp_path / "state.json")
    subprocess.check_call([driver, "--checkpoint-roundtrip", src, out])
    pg = _tolerant_json(out)["pose_graph"]
    low = pg["reprojection_low_level_feature_pose_graph_state"]["low_level_pg_state"]
    rep = pg["reprojection_low_level_feature_pose_graph_state"]
    obj = pg["obj_only_pose_graph_state_"]
    # ids are decimal strings, FactorType an int
    assert low["min_frame_id"] == "0" and low["max_frame_id"] == "500" and low["max_feature_factor_id"] == "9825256" and low["max_pose_factor_id"] == "135"
    assert low["visual_factor_type"] == 0
    for cam, (t, angle, axis) in exp["extrinsics"].items():
        e = _kv(low["camera_extrinsics_by_camera"])[cam]
        assert np.allclose(_mat(e["transl"], 3, 1), t, rtol=0, atol=0)
        # Eigen keeps (angle, axis) as given; the host types keep the rotation vector angle * axis: the product must survive
        assert np.allclose(e["rot"]["angle"] * _mat(e["rot"]["axis"], 3, 1), angle * np.array(axis), rtol=1e-14)
    for cam, k in exp["intrinsics"].items():   # host intrinsics are (fx, fy, cx, cy): entries (0,0), (1,1), (0,2), (1,2) of the matrix
        m = _mat(_kv(low["camera_intrinsics_by_camera"])[cam], 3, 3)
        assert (m[0], m[4], m[2], m[5]) == (k[0], k[4], k[2], k[5])
    for f, p in exp["robot_poses"].items():
        assert np.array_equal(_mat(_kv(low["robot_poses"])[f], 6, 1), p)
    for name in ("pose_factors_by_frame", "visual_factors_by_feature"):
        got = _kv(low[name])
        assert set(got) == set(exp[name])
        for k, v in exp[name].items():
            assert _pairs(got[k]) == sorted(map(tuple, v))
    got = _kv(low["visual_feature_factors_by_frame"])
    for k, v in exp["visual_feature_factors_by_frame"].items():      # a vector: order kept
        assert [(e["i"], int(e["v"]["f"]), int(e["v"]["s"])) for e in got[k]] == [(i, a, b) for i, (a, b) in enumerate(v)]
    for fid, f in exp["pose_factors"].items():
        g = _kv(low["pose_factors"])[fid]
        assert (g["frame_id_1"], g["frame_id_2"]) == (str(f["f1"]), str(f["f2"]))
        assert np.array_equal(_mat(g["pose_deviation_cov"], 6, 6), f["cov"]) and np.array_equal(_mat(g["measured_pose_deviation"]["transl"], 3, 1), f["t"])
        assert np.allclose(g["measured_pose_deviation"]["rot"]["angle"] * _mat(g["measured_pose_deviation"]["rot"]["axis"], 3, 1), f["angle"] * np.array(f["axis"]), rtol=1e-14)
    for fid, f in exp["factors"].items():
        g = _kv(low["factors"])[fid]
        assert (g["frame_id"], g["feature_id"], g["camera_id"]) == (str(f["frame"]), str(f["feat"]), str(f["cam"]))
        assert np.array_equal(_mat(g["feature_pos"], 2, 1), f["px"]) and g["reprojection_error_std_dev"] == f["sd"]
    assert _kv(low["last_observed_frame_by_feature"]) == {k: str(v) for k, v in exp["last_observed_frame_by_feature"].items()}
    assert _kv(low["first_observed_frame_by_feature"]) == {k: str(v) for k, v in exp["first_observed_frame_by_feature"].items()}
    assert (rep["min_feature_id"], rep["max_feature_id"]) == ("10", "50")
    for k, p in exp["feature_positions"].items():
        assert np.array_equal(_mat(_kv(rep["feature_positions"])[k], 3, 1), p)
    for name, (mean, cov) in exp["classes"].items():
        g = _kv(obj["mean_and_cov_by_semantic_class"])[name]
        assert np.array_equal(_mat(g["f"], 3, 1), mean) and np.array_equal(_mat(g["s"], 3, 3), cov)
    assert (obj["min_object_id"], obj["max_object_id"]) == ("93", "19038")
    for k, e in exp["ellipsoids"].items():
        assert np.array_equal(_mat(_kv(obj["ellipsoid_estimates"])[k], 7, 1), e)
    assert _kv(obj["semantic_class_for_object"]) == exp["semantic_class_for_object"]
    assert sorted(obj["long_term_map_object_ids"], key=int) == sorted(map(str, exp["long_term_map_object_ids"]), key=int)
    assert [obj[k] for k in ("min_object_observation_factor", "max_object_observation_factor", "min_obj_specific_factor", "max_obj_specific_factor")] == ["13", "93", "31", "193"]
    for fid, f in exp["object_observation_factors"].items():
        g = _kv(obj["object_observation_factors"])[fid]
        assert (g["frame_id"], g["camera_id"], g["object_id"]) == (str(f["frame"]), str(f["cam"]), str(f["obj"]))
        assert np.array_equal(_mat(g["bounding_box_corners"], 4, 1), f["corners"]) and np.array_equal(_mat(g["bounding_box_corners_covariance"], 4, 4), f["cov"])
        assert g["detection_confidence"] == f["conf"]
    for fid, f in exp["shape_dim_prior_factors"].items():
        g = _kv(obj["shape_dim_prior_factors"])[fid]
        assert g["object_id"] == str(f["obj"]) and np.array_equal(_mat(g["mean_shape_dim"], 3, 1), f["mean"]) and np.array_equal(_mat(g["shape_dim_cov"], 3, 3), f["cov"])
    for name in ("observation_factors_by_frame", "observation_factors_by_object", "object_only_factors_by_object"):
        got = _kv(obj[name])
        assert set(got) == set(exp[name])
        for k, v in exp[name].items():
            assert _pairs(got[k]) == sorted(map(tuple, v))
    # and the file our writer produced reads back to the same file (fixed point)
    out2 = str(tmp_path / "state2.json")
    subprocess.check_call([driver, "--checkpoint-roundtrip", out, out2])
    assert open(out).read() == open(out2).read()


def test_malformed_checkpoints_are_refused(driver, tmp_path):
    bad = tmp_path / "bad.json"
    for text in ("", "{", '{"pose_graph": 3}', '{"pose_graph": {"reprojection_low_level_feature_pose_graph_state": {}}}'):
        bad.write_text(text)
        p = subprocess.run([driver, "--checkpoint-roundtrip", str(bad), str(tmp_path / "o.json")], capture_output=True, text=True)
        assert p.returncode == 1 and "Could not read pose graph state" in p.stderr
    p = subprocess.run([driver, "--checkpoint-roundtrip", str(tmp_path / "missing.json"), str(tmp_path / "o.json")], capture_output=True, text=True)
    assert p.returncode == 1 and "does not exist" in p.stderr


def test_reader_is_strict_about_ids_depth_numbers_and_escapes(driver, tmp_path):
    """Ids must be whole decimal strings that fit 64 bits (not "12abc", "", "-1": strtoull would have read 12, 0, 2^64-1), a bare number only
    while a double holds it exactly; nesting is capped (a hostile file must not exhaust the stack); numbers parse the same in a comma-decimal
    locale; \\u escapes become UTF-8 (semantic class names)."""
    src = os.path.join(GOLDEN, "pose_graph_state_reference_roundtrip.json")
    text = open(src).read()
    bad, out = tmp_path / "bad.json", str(tmp_path / "o.json")
    key = '"max_frame_id":"500"'
    assert key in text
    for repl in ('"500abc"', '""', '"-1"', '"18446744073709551616"', '1e300', '12.5', '-3'):
        bad.write_text(text.replace(key, '"max_frame_id":' + repl, 1))
        p = subprocess.run([driver, "--checkpoint-roundtrip", str(bad), out], capture_output=True, text=True)
        assert p.returncode == 1 and "Could not read pose graph state" in p.stderr, repl
    for repl in ('"18446744073709551615"', '4096'):                       # the largest id; a small bare number
        bad.write_text(text.replace(key, '"max_frame_id":' + repl, 1))
        assert subprocess.run([driver, "--checkpoint-roundtrip", str(bad), out]).returncode == 0, repl
        assert ('"max_frame_id": "%s"' % repl.strip('"')) in open(out).read()
    bad.write_text("[" * 100000 + "]" * 100000)                           # 100 000 levels deep: an error, not a crash
    p = subprocess.run([driver, "--checkpoint-roundtrip", str(bad), out], capture_output=True, text=True)
    assert p.returncode == 1 and "nesting" in p.stderr
    # the same file under a comma-decimal locale (if the box has one) reads and writes the same numbers
    env = dict(os.environ, LC_ALL="de_DE.UTF-8", LANG="de_DE.UTF-8")
    a, b = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    subprocess.check_call([driver, "--checkpoint-roundtrip", src, a])
    subprocess.check_call([driver, "--checkpoint-roundtrip", src, b], env=env)
    assert open(a).read() == open(b).read()
    # a class name with non-ASCII characters written as \\u escapes comes back as UTF-8
    assert '"chair"' in text
    bad.write_text(text.replace('"chair"', '"caf\\u00e9 \\ud83d\\ude00"'))
    assert subprocess.run([driver, "--checkpoint-roundtrip", str(bad), out]).returncode == 0
    assert "caf\u00e9 \U0001F600" in open(out, encoding="utf-8").read()


@pytest.mark.gpu
def test_checkpoint_replay_the_run_opt_from_pg_state_way(driver, tmp_path):
    """A session writes its final state as long_term_map_checkpoint.json (optimization_runner.h:499-507); the checkpoint is then
    (a) flattened into the final global problem and solved on the HIP path and on the oracle from the same arrays -- same LM
    trajectory -- and (b) replayed end to end: run_opt_from_pg_state.cpp:160-312 = pose graph from the state, no frame data added,
    final global BA on the device, long-term map."""
    import obvi_ba
    prob = synth.make_problem(P=60, L=1200, O=3, seed=5, min_obj_obs=12, bbox_noise=5.0, object_classes=("bench",), stereo=True)
    scene = str(tmp_path / "scene.txt")
    scene_io.write_scene(prob, scene)
    ck = tmp_path / "ck"; ck.mkdir()
    out1 = str(tmp_path / "session.json")
    subprocess.check_call([driver, scene, out1, "--window", "20", "--gba-frequency", "25", "--save-checkpoint", str(ck)], timeout=600)
    ckpt = str(ck / "long_term_map_checkpoint.json")
    state = json.load(open(ckpt))["pose_graph"]
    low = state["reprojection_low_level_feature_pose_graph_state"]["low_level_pg_state"]
    assert low["max_frame_id"] == "59" and len(low["robot_poses"]) == 60 and len(low["factors"]) == len(prob["rp_pose"])
    session = json.load(open(out1))
    poses_ck = np.array([_mat(_kv(low["robot_poses"])[str(f)], 6, 1) for f in range(60)])
    assert np.array_equal(poses_ck, np.array(session["poses"]))                   # the checkpoint holds the session's final estimate, digit for digit
    # (a) the checkpoint's global problem, flattened by the host mirror, on both back ends
    flat = str(tmp_path / "flat.json")
    subprocess.check_call([driver, "--from-checkpoint", ckpt, flat, "--dump-build", "0", "59"])
    fp = json.load(open(flat))
    n_rp, n_bb = len(fp["rp_pose"]), len(fp["bb_obj"])
    assert n_rp > 5000 and n_bb > 20 and len(fp["frames"]) == 60
    prm = helpers.ba_params(max_it=12, ftol=1e-9)
    res = []
    for ba in (helpers.oracle_ba(), helpers.product_ba()):
        ba.set_cameras(np.array(fp["cam_K"]).reshape(-1, 4), np.array(fp["cam_ext"]).reshape(-1, 7))
        # start away from the session's optimum so that there is a trajectory to compare
        rng = np.random.default_rng(2)
        ba.set_poses(np.array(fp["poses"]).reshape(-1, 6) + 2e-3 * rng.normal(size=(60, 6)), np.array(fp["pose_const"], dtype=np.uint8))
        ba.set_points(np.array(fp["points"]).reshape(-1, 3), np.array(fp["point_const"], dtype=np.uint8))
        ba.set_objects(np.array(fp["object_values"]).reshape(-1, 7), np.array(fp["object_const"], dtype=np.uint8))
        ba.set_reproj(np.array(fp["rp_pose"], dtype=np.uint32), np.array(fp["rp_point"], dtype=np.uint32), np.array(fp["rp_cam"], dtype=np.uint16), np.array(fp["rp_pixel"]).reshape(-1, 2),
                      np.array(fp["rp_sigma"]), 1.0)
        ba.set_bbox(np.array(fp["bb_obj"], dtype=np.uint32), np.array(fp["bb_pose"], dtype=np.uint32), np.array(fp["bb_cam"], dtype=np.uint16), np.array(fp["bb_corners"]).reshape(-1, 4),
                    np.array(fp["bb_cov"]).reshape(-1, 16), 0.5, 1000.0)
        ba.set_shape_priors(np.array(fp["sp_obj"], dtype=np.uint32), np.array(fp["sp_mean"]).reshape(-1, 3), np.array(fp["sp_cov"]).reshape(-1, 9), 10.0)
        s = ba.solve(prm)
        res.append((s, [i.step_is_successful for i in ba.iterations()], [i.cost for i in ba.iterations()], ba.get_poses(), ba.get_objects()))
    (so, ao, co, po, oo), (sg, ag, cg, pg_, og) = res
    assert sg.num_iterations == so.num_iterations and ag == ao and sg.termination_type == so.termination_type
    assert max(abs(a - b) / b for a, b in zip(cg, co)) < 1e-8 and np.abs(pg_ - po).max() < 1e-7 and np.abs(og - oo).max() < 1e-6
    # (b) end-to-end replay
    out2 = str(tmp_path / "replay.json")
    ltm_file = str(tmp_path / "long_term_map.json")
    subprocess.check_call([driver, "--from-checkpoint", ckpt, out2, "--ltm", "--long-term-map-output", ltm_file], timeout=600)
    rep = json.load(open(out2))
    written = json.load(open(ltm_file))["long_term_map"]                                   # the reference's map file from a replayed checkpoint (ltm_extraction_only's job)
    assert {e["object_id"] for e in written["ellipsoid_results"]["ellipsoid_results_map"]} == set(rep["long_term_map"]) and all(e["class"] for e in written["ellipsoid_results"]["ellipsoid_results_map"])
    kinds = [r["kind"] for r in rep["records"]]
    assert rep["ok"] and "gba_phase_1" in kinds and "gba_phase_2" in kinds and "pgo" in kinds and not any(k.startswith("lba") for k in kinds)
    gba1 = [r for r in rep["records"] if r["kind"] == "gba_phase_1"][0]
    assert (gba1["min_frame"], gba1["max_frame"], gba1["n_poses"]) == (0, 59, 60) and gba1["final_cost"] <= gba1["initial_cost"] * (1 + 1e-9)
    assert set(rep["long_term_map"]) == set(session["objects"])
    for e in rep["long_term_map"].values():
        cov = np.array(e["covariance"]).reshape(7, 7)
        assert np.all(np.linalg.eigvalsh(0.5 * (cov + cov.T)) > 0)
    # replaying a converged session moves the trajectory only within the final BA's tolerance
    assert np.abs(np.array(rep["poses"]) - np.array(session["poses"])).max() < 5e-2
