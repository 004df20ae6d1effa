"""The C++ host-side mirror of the reference's optimiser interface (obvi-slam_amd/host/).
CPU: the flattening done by buildPoseGraphOptimization against an independent numpy statement of the selection
rules of object_pose_graph_optimizer.h:126-632.  GPU: a whole sliding-window session through OfflineProblemRunner."""
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
import scene_io
import synth

HOST = os.path.join(helpers.ROOT, "obvi-slam_amd", "host")
DRIVER = os.path.join(HOST, "run_offline_ba")


@pytest.fixture(scope="module")
def driver():
    if not os.path.exists(DRIVER):
        if not os.path.exists(helpers.PRODUCT_LIB):
            import __graft_entry__ as ge
            ge.build()
        subprocess.check_call(["make", "-C", HOST])
    return DRIVER


ORACLE_DRIVER = os.path.join(helpers.ROOT, "tests", "run_offline_ba_oracle")


@pytest.fixture(scope="module")
def oracle_driver():
    """The same driver compiled against the CPU oracle (tests/oracle_abi_shim.h renames the C ABI): the host mirror's whole session
    logic runs without a GPU, and gives the HIP session something independent to be compared with."""
    srcs = [os.path.join(HOST, "run_offline_ba.cpp"), os.path.join(helpers.ROOT, "tests", "oracle_abi_shim.cpp")]
    if not os.path.exists(helpers.ORACLE_LIB):
        subprocess.check_call(["make", "-C", os.path.join(helpers.ROOT, "oracle")])
    deps = srcs + [os.path.join(helpers.ROOT, "tests", "oracle_abi_shim.h"), helpers.ORACLE_LIB] + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if not os.path.exists(ORACLE_DRIVER) or os.path.getmtime(ORACLE_DRIVER) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(helpers.ROOT, "include"), "-I" + HOST, "-include", os.path.join(helpers.ROOT, "tests", "oracle_abi_shim.h"),
                               "-o", ORACLE_DRIVER] + srcs + ["-L" + os.path.join(helpers.ROOT, "oracle"), "-lobvi_oracle", "-Wl,-rpath,$ORIGIN/../oracle"])
    return ORACLE_DRIVER


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    prob = synth.make_problem(P=80, L=1500, O=4, seed=21, min_obj_obs=12, bbox_noise=5.0, object_classes=("bench", "trashcan"), stereo=True)
    path = str(tmp_path_factory.mktemp("scene") / "scene.txt")
    new_id = scene_io.write_scene(prob, path)
    return prob, path, new_id


def expected_build(prob, new_id, fmin, fmax, excluded_every=0, min_feat_obs=5, min_obj_obs=10, n_const=5):
    """object_pose_graph_optimizer.h:196-238 (features), :308-340 (objects), :424-472 (constant poses), :240-299 (odometry)."""
    frames = np.arange(fmin, fmax + 1)
    # visual factor ids are assigned frame by frame in file order (the data adder adds frame f's observations in order)
    order = np.lexsort((np.arange(len(prob["rp_pose"])), prob["rp_pose"]))
    fid = np.empty(len(order), dtype=np.int64); fid[order] = np.arange(len(order))
    inwin = (prob["rp_pose"] >= fmin) & (prob["rp_pose"] <= fmax)
    if excluded_every:
        inwin &= (fid % excluded_every != 0)
    cnt = np.bincount(prob["rp_point"][inwin], minlength=len(prob["points"]))
    feats = np.nonzero(cnt >= min_feat_obs)[0]
    rp_keep = inwin & np.isin(prob["rp_point"], feats)
    obs_per_frame = np.bincount(prob["rp_pose"][rp_keep], minlength=len(prob["poses"]))
    b_in = (prob["bb_pose"] >= fmin) & (prob["bb_pose"] <= fmax) & np.array([int(o) in new_id for o in prob["bb_obj"]])
    ocnt = np.bincount(prob["bb_obj"][b_in], minlength=len(prob["objects"]))
    objs_old = np.nonzero(ocnt >= min_obj_obs)[0]
    objs = sorted(new_id[int(o)] for o in objs_old)
    bb_keep = b_in & np.isin(prob["bb_obj"], objs_old)
    const = np.zeros(len(frames), np.uint8)
    if fmin == 0:
        const[0] = 1
    else:
        const[:min(n_const, len(frames))] = 1
    low = [f for f in frames if obs_per_frame[f] < 50]
    rel = set()
    for f in low:
        for a, b in ((f - 1, f), (f, f + 1)):
            if a >= fmin and b <= fmax:
                rel.add((a, b))
    return dict(frames=frames, features=feats, objects=objs, pose_const=const, n_rp=int(rp_keep.sum()), n_bb=int(bb_keep.sum()), n_sp=len(objs), rel=rel, fid=fid, rp_keep=rp_keep)


@pytest.mark.parametrize("window", [(0, 79), (20, 70), (30, 40), (0, 6)])
def test_build_flattening_matches_selection_rules(driver, scene, tmp_path, window):
    prob, path, new_id = scene
    out = str(tmp_path / "build.json")
    subprocess.check_call([driver, path, out, "--dump-build", str(window[0]), str(window[1])])
    got = json.load(open(out))
    exp = expected_build(prob, new_id, *window)
    assert np.array_equal(np.array(got["frames"], dtype=np.int64), exp["frames"])
    assert np.array_equal(np.array(got["features"], dtype=np.int64), exp["features"])
    assert [int(o) for o in got["objects"]] == exp["objects"]
    assert np.array_equal(np.array(got["pose_const"], dtype=np.uint8), exp["pose_const"])
    assert not any(got["point_const"]) and not any(got["object_const"])
    assert len(got["rp_pose"]) == exp["n_rp"] and len(got["bb_obj"]) == exp["n_bb"] and len(got["sp_obj"]) == exp["n_sp"]
    frames = np.array(got["frames"], dtype=np.int64)
    assert {(int(frames[int(a)]), int(frames[int(b)])) for a, b in zip(got["rl_a"], got["rl_b"])} == exp["rel"]
    assert got["num_blocks"] == exp["n_rp"] + exp["n_bb"] + exp["n_sp"] + len(got["rl_a"])
    # every flattened observation refers to an included feature and an in-window frame
    feats = np.array(got["features"], dtype=np.int64)
    pix = np.array(got["rp_pixel"]).reshape(-1, 2)
    key_got = sorted(zip(frames[np.array(got["rp_pose"], dtype=np.int64)].tolist(), feats[np.array(got["rp_point"], dtype=np.int64)].tolist(), pix[:, 0].tolist()))
    keep = exp["rp_keep"]
    key_exp = sorted(zip(prob["rp_pose"][keep].tolist(), prob["rp_point"][keep].tolist(), prob["rp_pixel"][keep, 0].tolist()))
    assert key_got == key_exp


def test_build_orders_residual_blocks_by_factor_id_whatever_the_insertion_order(driver, scene, tmp_path):
    """Residual blocks go by factor id (the reference walks an ordered set of ids).  The build flattens the pose graph's per-frame records
    as they come and only sorts when the ids do not ascend: with the frames entered last frame first they descend from frame to frame,
    and the result must be the same selection in ascending id order."""
    prob, path, new_id = scene
    outs = []
    for extra in ([], ["--frames-reversed"]):
        out = str(tmp_path / ("build%d.json" % len(outs)))
        subprocess.check_call([driver, path, out, "--dump-build", "20", "70"] + extra)
        outs.append(json.load(open(out)))
    exp = expected_build(prob, new_id, 20, 70)
    for got in outs:
        ids = np.array(got["rp_factor_ids"], dtype=np.int64)
        assert len(ids) == exp["n_rp"] and np.all(np.diff(ids) > 0)
        assert np.array_equal(np.array(got["features"], dtype=np.int64), exp["features"])
    fwd, rev = outs
    frames_f = np.array(fwd["frames"], dtype=np.int64)[np.array(fwd["rp_pose"], dtype=np.int64)]
    frames_r = np.array(rev["frames"], dtype=np.int64)[np.array(rev["rp_pose"], dtype=np.int64)]
    assert np.all(np.diff(frames_f) >= 0) and np.all(np.diff(frames_r) <= 0) and frames_r[0] > frames_r[-1]
    key = lambda g, fr: sorted(zip(fr.tolist(), np.array(g["features"], dtype=np.int64)[np.array(g["rp_point"], dtype=np.int64)].tolist(), np.array(g["rp_pixel"]).reshape(-1, 2)[:, 0].tolist()))
    assert key(fwd, frames_f) == key(rev, frames_r)


def test_build_honours_excluded_factors(driver, scene, tmp_path):
    """Phase II: excluded_feature_factor_types_and_ids drop factors *before* the min-observation filter (:886-905, :826-861)."""
    prob, path, new_id = scene
    out = str(tmp_path / "build.json")
    subprocess.check_call([driver, path, out, "--dump-build", "10", "60", "--excluded-every", "7"])
    got = json.load(open(out))
    exp = expected_build(prob, new_id, 10, 60, excluded_every=7)
    assert got["num_excluded"] > 0
    assert np.array_equal(np.array(got["features"], dtype=np.int64), exp["features"]) and len(got["rp_pose"]) == exp["n_rp"]


@pytest.mark.parametrize("window,every", [((0, 79), 7), ((20, 70), 3), ((30, 40), 2), ((10, 60), 5)])
def test_phase_two_masks_select_what_the_rebuild_selects(driver, scene, tmp_path, window, every):
    """excludeFromBuiltProblem (masks on the phase-I problem) against buildPoseGraphOptimization with the excluded set (what the
    reference does for phase II, offline_problem_runner.h:803-892): the surviving factors must be the same, id by id, and so must
    the blocks they leave in the problem."""
    prob, path, new_id = scene
    a, b = str(tmp_path / "rebuilt.json"), str(tmp_path / "masked.json")
    subprocess.check_call([driver, path, a, "--dump-build", str(window[0]), str(window[1]), "--excluded-every", str(every)])
    subprocess.check_call([driver, path, b, "--dump-build", str(window[0]), str(window[1]), "--excluded-every", str(every), "--phase-two-masks"])
    reb, msk = json.load(open(a)), json.load(open(b))
    ids = lambda d, k: np.array(d[k], dtype=np.int64)   # noqa: E731
    if not msk["masks_ok"]:
        # the rebuild would add odometry factors for a frame that fell below the per-frame minimum: the runner rebuilds then
        assert len(reb["rl_a"]) > len(msk["rl_a"])
        return
    for fam in ("rp", "bb", "sp"):
        keep = ids(msk, fam + "_id")[np.array(msk["mask_" + fam], dtype=bool)]
        assert np.array_equal(np.sort(keep), np.sort(ids(reb, fam + "_id"))), fam
    assert len(reb["rl_a"]) == len(msk["rl_a"])
    assert msk["mask_n_features"] == len(reb["features"]) and msk["mask_n_objects"] == len(reb["objects"])
    # blocks: features / objects with a surviving factor == the rebuilt problem's lists
    feats = np.unique(ids(msk, "features")[ids(msk, "rp_point")[np.array(msk["mask_rp"], dtype=bool)]])
    assert np.array_equal(feats, np.sort(ids(reb, "features")))
    objs = np.unique(ids(msk, "objects")[ids(msk, "bb_obj")[np.array(msk["mask_bb"], dtype=bool)]])
    assert np.array_equal(objs, np.sort(ids(reb, "objects")))
    assert np.array_equal(ids(msk, "frames"), ids(reb, "frames")) and np.array_equal(ids(msk, "pose_const"), ids(reb, "pose_const"))


def test_window_provider_and_gba_rule():
    """run_opt_utils.h:101-116 and optimization_runner.h:195-203 restated in numpy vs a brute-force table from the C++ rule."""
    def window(f, mx, freq=30, w=50):
        if f == mx or f % freq == 0 or f < w:
            return 0
        return f - w
    mx = 200
    gba = [f for f in range(1, mx + 1) if f - window(f, mx) > 50]
    assert gba == [60, 90, 120, 150, 180, 200]
    assert window(49, mx) == 0 and window(51, mx) == 1 and window(199, mx) == 149


def check_session(prob, out, csv):
    """offline_problem_runner.h:100-274: per-frame sliding-window two-phase BA, PGO + object optimisation at global-BA
    frames, final global BA; the CSV has the reference's columns."""
    res = json.load(open(out))
    assert res["ok"]
    # long-term map (the output extractor's covariance step, long_term_object_map_extraction.h:381-527): every object of the
    # final problem with its estimate and a symmetric positive-definite 7x7 marginal covariance
    ltm = res["long_term_map"]
    assert ltm and set(ltm) <= set(res["objects"])
    for oid, e in ltm.items():
        cov = np.array(e["covariance"]).reshape(7, 7)
        assert np.allclose(e["mean"], res["objects"][oid], rtol=0, atol=1e-12)
        assert np.abs(cov - cov.T).max() <= 1e-9 * np.abs(cov).max() and np.all(np.linalg.eigvalsh(0.5 * (cov + cov.T)) > 0)
    recs = res["records"]
    kinds = [r["kind"] for r in recs]
    P = len(prob["poses"])
    # one two-phase local BA per frame that is not a global-BA frame
    def window(f):
        return 0 if (f == P - 1 or f % 25 == 0 or f < 20) else f - 20
    gba_frames = [f for f in range(1, P) if f - window(f) > 20]
    assert gba_frames == [25, 50, 75, 79]
    lba1 = [r for r in recs if r["kind"] == "lba_phase_1"]
    assert [r["max_frame"] for r in lba1] == [f for f in range(1, P) if f not in gba_frames]
    assert all(r["min_frame"] == window(r["max_frame"]) for r in lba1)
    assert kinds.count("pgo") == len(gba_frames) + 1 and kinds.count("pre_pgo_track") == len(gba_frames) + 1
    # use_visual_features_on_global_ba = 0: no visual BA at the intermediate global-BA frames, but the final one runs it
    assert kinds.count("gba_phase_1") == 1 and kinds.count("gba_phase_2") == 1 and recs[-1]["kind"] in ("gba_phase_2", "reverted")
    for r in recs:
        if r["kind"].endswith("phase_1") or r["kind"].endswith("phase_2"):
            assert r["final_cost"] <= r["initial_cost"] * (1 + 1e-9) and r["iterations"] >= 1
    ph2 = [r for r in recs if r["kind"] == "lba_phase_2" and r["max_frame"] > 30]
    assert ph2 and all(r["n_excluded"] > 0 for r in ph2)
    # the optimised trajectory is closer to the truth than the odometry it started from
    poses = np.array(res["poses"])
    err0 = np.linalg.norm(prob["poses"][:, :3] - prob["gt_poses"][:, :3], axis=1).mean()
    err1 = np.linalg.norm(poses[:, :3] - prob["gt_poses"][:, :3], axis=1).mean()
    assert err1 < 0.7 * err0          # stereo rig: scale is observable, BA must beat the odometry prior
    header = open(csv).readline().strip()
    assert header == ("max_frame_id,outliers_excluded?,local_ba?,global_ba?,global_pgo?,num_poses,num_objects,num_visual_features,"
                      "total_ceres_time,linear_solver_time,jacobian_time,residual_time,num_ceres_iterations")
    rows = open(csv).read().strip().split("\n")[1:]
    assert len(rows) >= 2 * len(lba1)
    return res


@pytest.fixture(scope="module")
def oracle_session(oracle_driver, scene, tmp_path_factory):
    prob, path, _ = scene
    d = tmp_path_factory.mktemp("oracle_session")
    out, csv = str(d / "out.json"), str(d / "ceres_opt_summary.csv")
    subprocess.check_call([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25", "--csv", csv, "--ltm"], timeout=1200)
    return out, csv


def test_offline_runner_session_through_the_oracle(oracle_session, scene):
    """The host mirror's session logic end to end on the CPU: the driver bound to the oracle instead of libobvi_ba.so."""
    check_session(scene[0], *oracle_session)


TIMING_COLUMNS = (8, 9, 10, 11)   # total_ceres_time, linear_solver_time, jacobian_time, residual_time of ceres_opt_summary.csv


def _runner_hooks(stderr):
    return json.loads([ln for ln in stderr.splitlines() if ln.startswith("runner_hooks ")][-1][len("runner_hooks "):])


def test_reference_shaped_runner_is_the_same_session(oracle_driver, oracle_session, scene, tmp_path):
    """OfflineProblemRunner<InputProblemData, VisualFeatureFactorType, OutputProblemData, CachedFactorInfo, PoseGraphType> with the reference's fifteen
    constructor arguments (offline_problem_runner.h:27-98; a construction site as optimization_runner.h:509-543 writes it): the same session, digit for
    digit, as the short constructor + setters (the per-factor hooks run: a creator that makes every residual changes nothing); the visualization callback is called where
    the reference calls it (:164, :392, :512, :908, :245, :263)."""
    prob, path, _ = scene
    out = str(tmp_path / "out.json")
    r = subprocess.run([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25", "--ltm", "--reference-shaped-runner", "--count-visualization-calls"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = json.load(open(out)), json.load(open(oracle_session[0]))
    assert a["records"] == b["records"] and a["poses"] == b["poses"] and a["objects"] == b["objects"] and a.get("long_term_map") == b.get("long_term_map")
    hooks = _runner_hooks(r.stderr)
    assert hooks["ignored_hooks"] == ["ceres_callback_creator"]                       # round 6: the per-factor seam is honoured (test below); only the per-iteration callback has nobody to call it
    assert hooks["creator_calls"] > 10000 and hooks["creator_rejections"] == 0 and hooks["factors_left_out"] == 0 and hooks["refresh_calls"] > 0   # a creator that makes every residual: the same session
    before_any, before_each, after_each, after_pgo, after_all, after_post = hooks["visualization_calls"]
    P = len(prob["poses"])
    n_gba = sum(1 for rec in a["records"] if rec["kind"] == "pgo")
    attempts = sum(1 for rec in a["records"] if rec["kind"].endswith("phase_1"))
    assert before_any == 1 and after_all == 1 and after_post == 1
    assert before_each >= P and after_each == attempts <= before_each                # one runOptimizationIteration per frame from 1 on + the final one (+ re-runs after
                                                                                     # merges); AFTER_EACH only where the visual-feature optimisation ran (:522, :908)
    assert after_pgo == n_gba and n_gba >= 1


def test_a_residual_creator_that_cannot_make_a_residual_leaves_the_factor_out(oracle_driver, oracle_session, scene, tmp_path):
    """The reference's per-factor seam (object_pose_graph_optimizer.h:98-113, :1016-1052; residual_creator.h:347-436): buildPoseGraphOptimization offers every factor
    it selected to the caller's residual_creator, and one that returns false -- "Could not make residual" -- is left out of the problem.  Here: the reference-shaped
    runner with a creator that fails on every observation factor (visual, bounding box) whose id is 6 modulo 7.  Every problem of the session then holds exactly
    the other factors: the first window's residual count drops by the rejected share, every record's factor count does, the trajectory is still recovered, and the
    creator was asked again at every build (refresh_residual_checker says "refresh", as the reference's does, offline_problem_runner.h:273-279)."""
    prob, path, _ = scene
    out = str(tmp_path / "out.json")
    r = subprocess.run([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25", "--ltm", "--creator-rejects-every", "7"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    hooks = _runner_hooks(r.stderr)
    assert hooks["ignored_hooks"] == ["ceres_callback_creator"]
    assert hooks["creator_rejections"] > 1000 and hooks["factors_left_out"] == hooks["creator_rejections"]      # every rejection left a factor out, at every build
    assert 0.10 < hooks["creator_rejections"] / hooks["creator_calls"] < 0.16                                     # about every seventh of the observation factors (1/7 = 0.143, less the other families)
    assert r.stderr.count("Could not make residual for factor type") == hooks["creator_rejections"]              # the reference's message (:1049-1050)
    a, b = json.load(open(out)), json.load(open(oracle_session[0]))
    assert a["ok"] and [(x["kind"], x["min_frame"], x["max_frame"]) for x in a["records"]] == [(x["kind"], x["min_frame"], x["max_frame"]) for x in b["records"]]
    # fewer factors in every visual optimisation: the sum of squared residuals at the start of a window is smaller than the full session's
    pairs = [(x, y) for x, y in zip(a["records"], b["records"]) if x["kind"] == "lba_phase_1" and y["initial_cost"] > 1.0 and x["max_frame"] < 22]
    assert pairs and all(x["initial_cost"] < y["initial_cost"] for x, y in pairs)
    assert a["records"] != b["records"]
    poses = np.array(a["poses"])
    err0 = np.linalg.norm(prob["poses"][:, :3] - prob["gt_poses"][:, :3], axis=1).mean()
    assert np.linalg.norm(poses[:, :3] - prob["gt_poses"][:, :3], axis=1).mean() < 0.7 * err0                  # six sevenths of the data still beat the odometry


def test_planning_the_next_window_beside_the_solve_is_the_same_session(oracle_driver, oracle_session, scene, tmp_path):
    """The runner adds frame f+1's data, builds window f+1 and uploads it with its symbolic plan on a second thread while window f's last solve runs, and hands
    the start values over when they exist (obvi_ba_prepare / obvi_ba_update_state; obvi_runner.h runOptimization).  Nothing of a window's structure depends on the
    previous window's result, so the session is the serial one DIGIT FOR DIGIT (OBVI_HOST_PLAN_AHEAD=0 = serial; the module's oracle_session ran with the default)."""
    prob, path, _ = scene
    out, csv = str(tmp_path / "out.json"), str(tmp_path / "opt.csv")
    r = subprocess.run([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25", "--csv", csv, "--ltm"], capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, OBVI_HOST_PLAN_AHEAD="0", OBVI_HOST_TIMING="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "windows planned ahead" not in r.stderr
    a, b = json.load(open(out)), json.load(open(oracle_session[0]))
    assert a["records"] == b["records"] and a["poses"] == b["poses"] and a["objects"] == b["objects"] and a.get("long_term_map") == b.get("long_term_map")
    strip = lambda text: [",".join(c for i, c in enumerate(ln.split(",")) if i not in TIMING_COLUMNS) for ln in text.strip().split("\n")]
    assert strip(open(csv).read()) == strip(open(oracle_session[1]).read())
    # and the default did plan ahead: every local-BA window that follows a window of the loop
    r = subprocess.run([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25"], capture_output=True, text=True, timeout=1200, env=dict(os.environ, OBVI_HOST_TIMING="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stderr.splitlines() if ln.startswith("windows planned ahead")]
    assert line, r.stderr[-2000:]
    n = int(line[0].split(" x")[1].split(":")[0])
    n_gba = sum(1 for rec in a["records"] if rec["kind"] == "pgo")
    assert len(prob["poses"]) - 2 - n_gba - 6 <= n <= len(prob["poses"]) - 2   # not across global-BA frames, not from the first frames (no visual-feature optimisation there)


def test_sessions_in_one_process_are_the_single_session_each(oracle_driver, oracle_session, scene, tmp_path):
    """`run_offline_ba --sessions-in-process K`: K sessions over the scene at once, a host thread, a runner, a pose graph and device handles each (sessions
    per GPU).  Nothing is shared but the read-only scene and the library's host threads: every one of them is the single session digit for digit."""
    prob, path, _ = scene
    out, csv = str(tmp_path / "out.json"), str(tmp_path / "opt.csv")
    r = subprocess.run([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25", "--csv", csv, "--ltm", "--sessions-in-process", "3"], capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["sessions_in_process"] == 3 and len(line["session_s"]) == 3 and line["frames"] == len(prob["poses"])
    b = json.load(open(oracle_session[0]))
    strip = lambda text: [",".join(c for i, c in enumerate(ln.split(",")) if i not in TIMING_COLUMNS) for ln in text.strip().split("\n")]
    for suffix in ("", ".1", ".2"):
        a = json.load(open(out + suffix))
        assert a["records"] == b["records"] and a["poses"] == b["poses"] and a["objects"] == b["objects"] and a.get("long_term_map") == b.get("long_term_map"), suffix
        assert strip(open(csv + suffix).read()) == strip(open(oracle_session[1]).read()), suffix


def test_limit_on_the_evaluated_trajectory(oracle_driver, scene, tmp_path):
    """LimitTrajectoryEvaluationParams (offline_problem_runner.h:143-147): the session stops at min(max_frame_id_, last frame) -- through either runner shape."""
    prob, path, _ = scene
    outs = []
    for extra in ([], ["--reference-shaped-runner"]):
        out = str(tmp_path / "out.json")
        r = subprocess.run([oracle_driver, path, out, "--window", "20", "--gba-frequency", "25", "--max-frame", "30"] + extra, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.load(open(out)))
    a, b = outs
    assert a["records"] == b["records"] and a["poses"] == b["poses"]
    assert max(rec["max_frame"] for rec in a["records"]) == 30
    assert 26 <= sum(1 for rec in a["records"] if rec["kind"].endswith("phase_1")) <= 40   # frames 1..30 + the final one, less the first frames without a visual-feature optimisation


def test_big_builds_flatten_on_host_threads_into_the_same_arrays(driver, tmp_path):
    """A global-BA frame flattens millions of sightings: above 2^18 records buildPoseGraphOptimization writes the frames' spans on host
    threads.  Same flat problem, array for array, as the plain loop (OBVI_HOST_BUILD_THREADS=1) -- also when frames were filled last
    frame first (factor ids descend: the sorted route) and with every third factor excluded."""
    prob = synth.make_problem(P=260, L=36000, O=3, seed=8, min_obj_obs=6, bbox_noise=5.0, object_classes=("bench",))
    assert len(prob["rp_pose"]) > (1 << 18)
    path = str(tmp_path / "scene.bin")
    scene_io.write_scene_binary(prob, path)
    for extra in ([], ["--frames-reversed"], ["--excluded-every", "3"]):
        outs = []
        for threads in ("1", "6"):
            out = str(tmp_path / ("build_%s.json" % threads))
            subprocess.check_call([driver, path, out, "--dump-build", "0", "259"] + extra, timeout=900, env=dict(os.environ, OBVI_HOST_BUILD_THREADS=threads))
            outs.append(open(out).read())
        assert outs[0] == outs[1] and len(outs[0]) > 1000000, extra


def test_global_ba_mode_and_the_binary_scene(oracle_driver, scene, tmp_path):
    """`run_offline_ba --global-ba` (bench.py's end_to_end_cpp leg): every frame enters the pose graph, then the runner starts at the last frame --
    with a 20-frame window over 80 frames that is a global BA: the pose-graph stage of runPgoPlusEllipsoids, then the two phases.  And the
    binary scene (scene_io.write_scene_binary / loadSceneBinary) is the text scene: same records, same results, digit for digit."""
    prob, path, new_id = scene
    bin_path = str(tmp_path / "scene.bin")
    assert scene_io.write_scene_binary(prob, bin_path) == new_id
    res = []
    for sc in (path, bin_path):
        out = str(tmp_path / "out.json")
        r = subprocess.run([oracle_driver, sc, out, "--global-ba", "--window", "20"], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["ok"] and set(line) >= {"scene_load_ms", "pose_graph_ms", "run_full_optimization_ms", "records"}
        res.append((json.load(open(out)), line))
    (a, la), (b, lb) = res
    assert a["poses"] == b["poses"] and a["objects"] == b["objects"] and la["records"] == lb["records"]
    kinds = [r["kind"] for r in la["records"]]
    assert kinds[0].startswith("pgo") or "pgo" in kinds[0] or kinds[0].startswith("gba"), kinds        # the global-BA branch, not a local window
    assert any(k.endswith("phase_2") for k in kinds) and all(r["n_poses"] == len(prob["poses"]) for r in la["records"] if "phase" in r["kind"])


@pytest.mark.gpu
def test_offline_runner_session(driver, scene, tmp_path):
    """The same session on the HIP path, and against the oracle-driven one: the same sequence of optimisations over the same windows
    and blocks; the first optimisations equal to round-off; later ones as close as two chains of LM runs with loose tolerances
    (1e-3 / 1e-4) and a discontinuous 10 % outlier cut stay -- equally good, and the same map in the end."""
    prob, path, _ = scene
    out, csv = str(tmp_path / "out.json"), str(tmp_path / "ceres_opt_summary.csv")
    subprocess.check_call([driver, path, out, "--window", "20", "--gba-frequency", "25", "--csv", csv, "--ltm"], timeout=600)
    check_session(prob, out, csv)


@pytest.mark.gpu
def test_hip_session_planned_ahead_is_the_serial_session_bit_for_bit(driver, scene, tmp_path):
    """On the device: with fixed-order sums (--deterministic) a session is reproducible, so the session whose windows are planned on the second thread and handle
    beside the solves (the default) must be the serial one (OBVI_HOST_PLAN_AHEAD=0) digit for digit -- every optimisation's iteration count and costs, every pose."""
    prob, path, _ = scene
    res = []
    for mode in ("1", "0"):
        out = str(tmp_path / ("out_%s.json" % mode))
        r = subprocess.run([driver, path, out, "--window", "20", "--gba-frequency", "25", "--deterministic", "--ltm"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, OBVI_HOST_PLAN_AHEAD=mode, OBVI_HOST_TIMING="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("windows planned ahead" in r.stderr) == (mode == "1")
        res.append(json.load(open(out)))
    a, b = res
    assert a["records"] == b["records"] and a["poses"] == b["poses"] and a["objects"] == b["objects"] and a.get("long_term_map") == b.get("long_term_map")


@pytest.mark.gpu
def test_hip_session_against_the_oracle_session(driver, oracle_session, scene, tmp_path):
    prob, path, _ = scene
    out = str(tmp_path / "out.json")
    subprocess.check_call([driver, path, out, "--window", "20", "--gba-frequency", "25", "--ltm"], timeout=600)
    hip, ora = json.load(open(out)), json.load(open(oracle_session[0]))
    rh, ro = hip["records"], ora["records"]
    assert [(r["kind"], r["min_frame"], r["max_frame"]) for r in rh] == [(r["kind"], r["min_frame"], r["max_frame"]) for r in ro]
    # the first windows with something to optimise: identical start, identical problem -> the oracle's LM trajectory
    first = [(a, b) for a, b in zip(rh, ro) if b["initial_cost"] > 1e-3][:2]
    assert len(first) == 2
    for a, b in first:
        assert (a["n_poses"], a["n_features"], a["n_excluded"], a["iterations"]) == (b["n_poses"], b["n_features"], b["n_excluded"], b["iterations"])
        assert abs(a["initial_cost"] - b["initial_cost"]) <= 1e-10 * b["initial_cost"] and abs(a["final_cost"] - b["final_cost"]) <= 1e-8 * b["final_cost"]
    # the whole session: same problem sizes nearly everywhere (a feature at the observation-count threshold may flip with an excluded
    # factor), same LM iteration counts in most windows, costs within the solves' own tolerances
    same_size = sum((a["n_poses"], a["n_features"]) == (b["n_poses"], b["n_features"]) for a, b in zip(rh, ro))
    same_its = sum(a["iterations"] == b["iterations"] for a, b in zip(rh, ro))
    assert same_size >= 0.95 * len(ro) and same_its >= 0.8 * len(ro), (same_size, same_its, len(ro))   # measured over repeated runs: 162 / 162, 151-153 / 162
    rel = [abs(a["final_cost"] - b["final_cost"]) / max(b["final_cost"], 1e-12) for a, b in zip(rh, ro) if a["kind"].endswith(("phase_1", "phase_2"))]
    assert np.median(rel) <= 1e-3 and max(rel) <= 5e-2, (np.median(rel), max(rel))   # measured: 1e-5, 1.6e-3
    # the outcome: trajectory, objects and long-term map
    ph, po = np.array(hip["poses"]), np.array(ora["poses"])
    assert np.abs(ph[:, :3] - po[:, :3]).max() <= 2e-2 and np.abs(ph[:, 3:] - po[:, 3:]).max() <= 2e-3   # measured: 2e-3 m, 1e-4 rad
    err = [np.linalg.norm(x[:, :3] - prob["gt_poses"][:, :3], axis=1).mean() for x in (ph, po)]
    assert abs(err[0] - err[1]) <= 0.25 * err[1]          # (measured over the rounds: 0.5 % ... 10 %, either way round: two chaotic sessions, both 2 cm from the truth)
    assert set(hip["objects"]) == set(ora["objects"]) and set(hip["long_term_map"]) == set(ora["long_term_map"])
    for oid in ora["objects"]:
        a, b = np.array(hip["objects"][oid]), np.array(ora["objects"][oid])
        assert np.abs(np.delete(a - b, 3)).max() <= 0.15   # measured: up to 4 cm (the fp64 atomics make the HIP session differ from run to run in the last digits)
        if abs(b[4] - b[5]) > 0.1:   # the yaw of an ellipsoid with equal horizontal axes is not observable (and drifts freely in both runs)
            assert abs(np.sin(a[3] - b[3])) <= 5e-2
    for oid, e in ora["long_term_map"].items():
        ch, co = np.array(hip["long_term_map"][oid]["covariance"]).reshape(7, 7), np.array(e["covariance"]).reshape(7, 7)
        keep = [0, 1, 2, 4, 5, 6]   # without the yaw (above)
        assert np.abs(np.sqrt(np.diag(ch)[keep]) - np.sqrt(np.diag(co)[keep])).max() <= 0.2 * np.sqrt(np.diag(co)[keep]).max()


@pytest.mark.gpu
def test_hip_session_with_a_rejecting_creator_against_the_oracle_session(driver, oracle_driver, scene, tmp_path):
    """The per-factor seam on the device path: the same creator (fails on every observation factor with id 6 modulo 7) in the HIP-bound and in the oracle-bound
    driver -- the same factors are left out of every problem, the first windows are the oracle's LM runs step for step, the session ends where the oracle's does."""
    prob, path, _ = scene
    outs = []
    for exe, name in ((driver, "hip"), (oracle_driver, "oracle")):
        out = str(tmp_path / (name + ".json"))
        r = subprocess.run([exe, path, out, "--window", "20", "--gba-frequency", "25", "--creator-rejects-every", "7"], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((json.load(open(out)), _runner_hooks(r.stderr)))
    (hip, hh), (ora, ho) = outs
    assert hh["ignored_hooks"] == ho["ignored_hooks"] == ["ceres_callback_creator"] and hh["creator_rejections"] > 1000
    rh, ro = hip["records"], ora["records"]
    assert [(r["kind"], r["min_frame"], r["max_frame"]) for r in rh] == [(r["kind"], r["min_frame"], r["max_frame"]) for r in ro]
    first = [(a, b) for a, b in zip(rh, ro) if b["initial_cost"] > 1e-3][:3]
    for a, b in first:
        assert (a["n_poses"], a["n_features"], a["n_excluded"], a["iterations"]) == (b["n_poses"], b["n_features"], b["n_excluded"], b["iterations"])
        assert abs(a["initial_cost"] - b["initial_cost"]) <= 1e-10 * b["initial_cost"] and abs(a["final_cost"] - b["final_cost"]) <= 1e-8 * b["final_cost"]
    same_size = sum((a["n_poses"], a["n_features"]) == (b["n_poses"], b["n_features"]) for a, b in zip(rh, ro))
    assert same_size >= 0.95 * len(ro)
    ph, po = np.array(hip["poses"]), np.array(ora["poses"])
    assert np.abs(ph[:, :3] - po[:, :3]).max() <= 3e-2 and np.abs(ph[:, 3:] - po[:, 3:]).max() <= 3e-3


@pytest.mark.gpu
def test_phase_two_on_the_phase_one_problem_equals_the_rebuild(driver, scene, tmp_path):
    """Phase II of every window normally runs on the problem phase I left on the device (masks + the kept symbolic plan) instead of
    re-flattening and re-uploading it with the excluded factors as the reference does (offline_problem_runner.h:803-892).  Both
    routes through the same session must give the same records -- window, blocks, excluded factors, LM iterations, costs -- and the same
    trajectory up to the round-off of a different elimination order."""
    import os
    prob, path, _ = scene
    res = []
    for rebuild in ("0", "1"):
        out = str(tmp_path / ("out%s.json" % rebuild))
        env = dict(os.environ, OBVI_HOST_PHASE2_REBUILD=rebuild, OBVI_HOST_TIMING="1")
        p = subprocess.run([driver, path, out, "--window", "20", "--gba-frequency", "25"], timeout=600, env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        res.append((json.load(open(out)), p.stderr))
    (fast, log_fast), (slow, log_slow) = res
    assert "rebuilt x0" in log_fast and "(masks) x0" in log_slow, (log_fast, log_slow)
    assert [r["kind"] for r in fast["records"]] == [r["kind"] for r in slow["records"]]
    # the first phase II starts from identical values on an identical selection: same objective to round-off, same LM trajectory
    k0 = next(k for k, r in enumerate(fast["records"]) if r["kind"] == "lba_phase_2" and r["n_excluded"] > 0)
    a, b = fast["records"][k0], slow["records"][k0]
    assert a["n_excluded"] == b["n_excluded"] and a["iterations"] == b["iterations"]
    assert abs(a["initial_cost"] - b["initial_cost"]) <= 1e-11 * b["initial_cost"] and abs(a["final_cost"] - b["final_cost"]) <= 1e-8 * b["final_cost"]
    # window by window (OBVI_HOST_PHASE2_CHECK=1: every masked phase II is repeated the reference's way -- rebuild, upload, solve --
    # from the same start values on a scratch handle): same problem size, same objective, same LM trajectory up to an iteration that
    # a tolerance test (1e-4) may flip on round-off, values within that tolerance's reach
    out = str(tmp_path / "out_check.json")
    p = subprocess.run([driver, path, out, "--window", "20", "--gba-frequency", "25"], timeout=900, capture_output=True, text=True,
                       env=dict(os.environ, OBVI_HOST_PHASE2_CHECK="1", OBVI_HOST_TIMING="1"))
    assert p.returncode == 0, p.stderr
    line = [ln for ln in p.stderr.split("\n") if ln.startswith("phase2_check")][0].split()
    chk = {line[i]: float(line[i + 1]) for i in range(1, len(line), 2)}
    assert chk["windows"] >= 60 and chk["failures"] == 0 and chk["size_mismatches"] == 0
    assert chk["max_initial_cost_rel"] <= 1e-10
    assert chk["iteration_mismatches"] <= 0.1 * chk["windows"] and chk["max_final_cost_rel"] <= 1e-3 and chk["max_value_diff"] <= 1e-2 and chk["points_apart"] <= 1e-2 * chk["points"] and chk["objects_apart"] <= 0.05 * chk["objects"]
    # the two whole sessions: two chains of LM runs with loose tolerances and a discontinuous 10 % cut drift apart in the last
    # digits and then exclude slightly different factors; they must stay equally good
    for a, b in zip(fast["records"], slow["records"]):
        assert (a["min_frame"], a["max_frame"]) == (b["min_frame"], b["max_frame"])
    err = [np.linalg.norm(np.array(r["poses"])[:, :3] - prob["gt_poses"][:, :3], axis=1).mean() for r in (fast, slow)]
    assert abs(err[0] - err[1]) <= 0.15 * err[1]


@pytest.mark.gpu
def test_pending_object_estimator_mirror(driver, tmp_path):
    """refineInitialEstimateForPendingObjects (pending_object_estimator.cpp:11-151) through the C++ mirror against the same
    problem pushed through the binding: every object of a scene, its boxes, its class prior; poses constant.  Objects of the
    one class with an observable yaw (dx != dy), so that the minimum is a point and the two runs can be compared digit by digit."""
    prob = synth.make_problem(P=60, L=300, O=5, seed=33, min_obj_obs=8, bbox_noise=3.0, object_classes=("bench",))
    path = str(tmp_path / "scene.txt")
    new_id = scene_io.write_scene(prob, path)
    out = str(tmp_path / "pending.json")
    subprocess.check_call([driver, path, out, "--pending-objects"], timeout=300)
    res = json.load(open(out))
    assert res["ok"] and res["final_cost"] < res["initial_cost"]
    # the same problem through the binding: scene objects only (renumbered), poses constant, bounding boxes + shape priors
    old_of = {v: k for k, v in new_id.items()}
    keep = np.array([int(o) in new_id for o in prob["bb_obj"]])
    q = dict(prob)
    q["objects"] = np.array([prob["objects"][old_of[i]] for i in range(len(new_id))])
    q["object_const"] = np.zeros(len(new_id), np.uint8)
    q["bb_obj"] = np.array([new_id[int(o)] for o in prob["bb_obj"][keep]], dtype=np.uint32)
    for k in ("bb_pose", "bb_cam", "bb_corners", "bb_cov"):
        q[k] = prob[k][keep]
    q["sp_obj"] = np.arange(len(new_id), dtype=np.uint32)
    q["sp_mean"] = np.array([prob["sp_mean"][old_of[i]] for i in range(len(new_id))])
    q["sp_cov"] = np.array([prob["sp_cov"][old_of[i]] for i in range(len(new_id))])
    q["pose_const"] = np.ones(len(prob["poses"]), np.uint8)
    g = helpers.product_ba(); synth.upload(g, q, reproj=False, relpose=False)
    s = g.solve(helpers.ba_params(max_it=100, ftol=1e-6, radius=1e4, max_radius=1e16))     # the estimator leaves the radii at Ceres' defaults
    assert s.num_iterations == res["iterations"]
    assert abs(s.initial_cost - res["initial_cost"]) <= 1e-9 * s.initial_cost and abs(s.final_cost - res["final_cost"]) <= 1e-8 * s.final_cost
    est = g.get_objects()
    for i in range(len(new_id)):
        assert np.abs(np.array(res["objects"][str(i)]) - est[i]).max() < 1e-6


@pytest.mark.gpu
def test_session_end_merge_rank_repair_and_iteration_logs(driver, tmp_path, monkeypatch):
    """The rest of runFullOptimization's session end (optimization_runner.h:545-640, offline_problem_runner.h:254-262, 918-958,
    long_term_object_map_extraction.cpp:929-1062) and the per-iteration CSVs (optimization_logger.h:29-147):
      * an object that the front end split in two (same place, same class) is merged after the final global BA, which is re-run;
      * an object whose every box falls in the constant invalid-ellipse branch (the camera is inside it) has zero Jacobian columns
        for its pose: the covariance extraction fails, ParameterPriors are put on the weakest columns, and the retry succeeds;
      * ceres_iterations_<type>.csv holds one row per LM iteration of every optimisation of that type."""
    monkeypatch.setitem(synth.SHAPE_CLASSES, "hangar", ((300.0, 300.0, 300.0), (1.0, 1.0, 1.0)))
    prob = synth.make_problem(P=80, L=1500, O=2, seed=21, min_obj_obs=30, bbox_noise=5.0, object_classes=("bench",), stereo=True)
    n0 = len(prob["objects"])
    assert n0 == 2
    # split object 0: its later observations go to a new object 2 that starts 20 cm away
    mine = np.flatnonzero(prob["bb_obj"] == 0)
    later = mine[len(mine) // 2:]
    assert len(mine) - len(later) >= 12 and len(later) >= 12
    bb_obj = prob["bb_obj"].copy(); bb_obj[later] = n0
    dup = prob["objects"][0].copy(); dup[0] += 0.2
    # a degenerate object 3: the robot drives around inside it
    frames = np.arange(10, 26)
    hangar = np.array([0.0, 0.0, 0.0, 0.3, 300.0, 300.0, 300.0])
    prob["objects"] = np.concatenate([prob["objects"], dup[None], hangar[None]])
    prob["obj_class"] = list(prob["obj_class"]) + [prob["obj_class"][0], "hangar"]
    prob["bb_obj"] = np.concatenate([bb_obj, np.full(len(frames), n0 + 1, np.uint32)]).astype(np.uint32)
    prob["bb_pose"] = np.concatenate([prob["bb_pose"], frames]).astype(np.uint32)
    prob["bb_cam"] = np.concatenate([prob["bb_cam"], np.zeros(len(frames), np.uint16)])
    prob["bb_corners"] = np.concatenate([prob["bb_corners"], np.tile([100.0, 300.0, 80.0, 260.0], (len(frames), 1))])
    prob["bb_cov"] = np.concatenate([prob["bb_cov"], np.tile(prob["bb_cov"][0], (len(frames), 1))])
    scene = str(tmp_path / "scene.txt")
    new_id = scene_io.write_scene(prob, scene)
    logs = tmp_path / "logs"; logs.mkdir()
    out = str(tmp_path / "out.json")
    p = subprocess.run([driver, scene, out, "--window", "20", "--gba-frequency", "25", "--ltm", "--merge-distance", "1.0", "--iteration-log-dir", str(logs)],
                       timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.load(open(out))
    assert res["ok"]
    # ---- merge: one of the two halves is gone, its boxes live on in the other, and the final BA ran once more (attempt 2) ----
    a, b = str(new_id[0]), str(new_id[n0])
    assert res["merge_rounds"] == 1 and (a in res["objects"]) != (b in res["objects"])
    kinds = [r["kind"] for r in res["records"]]
    assert kinds.count("gba_phase_1") == 2                                  # the final global BA and its re-run after the merge
    survivor = res["objects"][a] if a in res["objects"] else res["objects"][b]
    assert np.linalg.norm(np.array(survivor[:3]) - prob["gt_objects"][0][:3]) < 1.0
    # ---- rank repair ----
    g = new_id[n0 + 1]
    rep = res["covariance_rank_repairs"]
    mine = sorted(r["param_idx"] for r in rep if r["block_kind"] == 2 and r["block_id"] == g and r["retry"] == 1)
    assert mine[:4] == [0, 1, 2, 3]                                         # x, y, z, yaw of the hangar: zero columns
    assert all(r["col_sqnorm"] == 0.0 for r in rep if r["block_kind"] == 2 and r["block_id"] == g and r["param_idx"] < 4)
    assert 4 <= len(rep) <= 4 + 51 and all(r["prior_std_dev"] > 0 and np.isfinite(r["prior_std_dev"]) for r in rep)
    assert "Retrying rank deficient jacobian, retry num 1" in p.stderr
    ltm = res["long_term_map"]
    assert set(ltm) == set(res["objects"])
    for oid, e in ltm.items():
        cov = np.array(e["covariance"]).reshape(7, 7)
        assert np.all(np.isfinite(cov)) and np.all(np.linalg.eigvalsh(0.5 * (cov + cov.T)) > 0)
    sd = {r["param_idx"]: r["prior_std_dev"] for r in rep if r["block_kind"] == 2 and r["block_id"] == g}
    cov_g = np.array(ltm[str(g)]["covariance"]).reshape(7, 7)
    assert np.allclose(np.diag(cov_g)[:4], [sd[k] ** 2 for k in range(4)], rtol=1e-6)     # only the priors inform the hangar's pose
    # ---- iteration logs ----
    header = "optimization_id,iteration_num,cost,cost_change,step_norm,step_norm_per_param,is_successful"
    rows = {}
    for kind in ("lba_phase_1", "lba_phase_2", "gba_phase_1", "gba_phase_2", "pgo", "pre_pgo_track", "vf_adjust"):
        lines = open(logs / ("ceres_iterations_%s.csv" % kind)).read().strip().split("\n")
        assert lines[0] == header and len(lines) > 1
        rows[kind] = [ln.split(",") for ln in lines[1:]]
    # the logger calls an optimisation "global" when its window starts at frame 0 (offline_problem_runner.h:403-404, 808-809), whatever the
    # runner's own local / global decision was: the first frames of a session land in the gba files
    # ... and phase I of a visual BA that follows a PGO stage in the same iteration is not logged at all: the PGO stage's
    # writeCurrentOptInfo() resets the logger's type flags (optimization_logger.h:283) and nothing sets them again before phase I
    # (offline_problem_runner.h:403 comes before the PGO stage), so extractOptimizationTimingResults finds no type (:221-224).  Phase II
    # sets them again (:808).  The mirror keeps the reference's behaviour.
    n_iter = {k: 0 for k in ("lba_phase_1", "lba_phase_2", "gba_phase_1", "gba_phase_2")}
    recs = res["records"]
    for i, r in enumerate(recs):
        if r["kind"][4:] not in ("phase_1", "phase_2"):
            continue
        if r["kind"][4:] == "phase_1" and i > 0 and recs[i - 1]["kind"] == "pgo" and recs[i - 1]["max_frame"] == r["max_frame"]:
            continue
        n_iter[("gba_" if r["min_frame"] == 0 else "lba_") + r["kind"][4:]] += r["iterations"]
    assert {k: len(rows[k]) for k in n_iter} == n_iter                       # iterations.size() rows per optimisation, the iteration-0 record included
    assert {"79_1", "79_2"} <= {r[0] for r in rows["gba_phase_2"]} and "1_0" in {r[0] for r in rows["gba_phase_1"]}   # "<max frame>_<attempt>"
    assert {"79_1", "79_2"} <= {r[0] for r in rows["pgo"]}
    assert all(r[0] == "%d_0" % int(r[0].split("_")[0]) for r in rows["lba_phase_1"])
    first = [r for r in rows["lba_phase_1"] if r[0] == rows["lba_phase_1"][0][0]]
    assert [int(r[1]) for r in first] == list(range(len(first))) and first[0][6] == "1"
