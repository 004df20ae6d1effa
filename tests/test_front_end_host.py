"""The stateful half of the visual-feature front end (obvi-slam_amd/host/obvi_visual_feature_front_end.h): which observations and
features of a frame enter the pose graph (visual_feature_front_end.h:262-450, 640-724).

The C++ mirror runs the frame's features in lock step and sends their questions to the device in batches; the reference walks
them one by one.  Here the reference's sequence is restated literally in Python -- one question per call, through the oracle's
vote / parallax entries -- and both must let exactly the same factors into the graph.  CPU: the C++ mirror bound to the oracle
(tests/oracle_abi_shim.h).  GPU: the mirror on the HIP path."""
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
import obvi_ba
import scene_io
import synth
from test_host_mirror import driver, oracle_driver  # noqa: F401  (fixtures)

WINDOW, GBA_FREQ, PAST_N = 20, 25, 5


def window_start(f, max_frame):   # run_opt_utils.h:101-116
    return 0 if (f == max_frame or f % GBA_FREQ == 0 or f < WINDOW) else f - WINDOW


@pytest.fixture(scope="module")
def outlier_scene(tmp_path_factory):
    # one camera: between the two cameras of a stereo rig (same orientation, sideways baseline) the reference's normalised epipolar
    # error has its epipole at infinity and the vote of a same-frame partner depends on the last bit of the extrinsics (7.97 px or
    # -504 px for the same pair) -- nothing two implementations can be compared on
    prob = synth.make_problem(P=60, L=800, O=2, seed=5, min_obj_obs=12, bbox_noise=5.0, object_classes=("bench",), stereo=False)
    rng = np.random.default_rng(0)
    n = len(prob["rp_pose"])
    bad = rng.choice(n, n // 20, replace=False)           # 5 % of the sightings are gross mismatches: 40-90 px off
    prob["rp_pixel"] = prob["rp_pixel"].copy()
    prob["rp_pixel"][bad] += rng.choice([-1, 1], (len(bad), 2)) * rng.uniform(40, 90, (len(bad), 2))
    path = str(tmp_path_factory.mktemp("fe") / "scene.txt")
    scene_io.write_scene(prob, path)
    return prob, path, bad


class SequentialFrontEnd:
    """visual_feature_front_end.h:262-450 statement by statement (enforce_epipolar_error_requirement_ = true, early_votes_return_ = true,
    pixel parallax enforced, robot-pose parallax not: config/base7a_2_fallback.json)."""

    def __init__(self, ba, prob):
        self.ba, self.prob = ba, prob
        self.max_frame = len(prob["poses"]) - 1
        self.by_feature = {}        # the pose graph's reprojection factors: feature -> [(frame, feature, cam, (u, v))]
        self.added, self.pending, self.pending_init = set(), {}, {}
        self.obs_by_frame = [[] for _ in prob["poses"]]
        for k in range(len(prob["rp_pose"])):
            self.obs_by_frame[int(prob["rp_pose"][k])].append((int(prob["rp_point"][k]), int(prob["rp_cam"][k]), tuple(prob["rp_pixel"][k])))

    def is_inlier(self, cand, refs):   # isReprojectionErrorFactorInlier :511-602
        frames = sorted(refs)
        rp, rc, rx, rf, rs = [], [], [], [], []
        for fr in frames:
            for f in refs[fr]:
                rp.append(f[0]); rc.append(f[2]); rx.append(f[3]); rf.append(fr)
                rs.append(1 if (f[0], f[1], f[2]) == (cand[0], cand[1], cand[2]) else 0)     # shouldBeTheSame
        if not rp:
            return False
        _, _, inl = self.ba.epipolar_votes(self.prob["K"], self.prob["ext"], self.prob["poses"], [cand[0]], [cand[2]], [cand[3]], [0, len(rp)], rp, rc, rx, rf, rs)
        return bool(inl[0])

    def cache_add(self, cache, frame, factors):   # addFactorsAndRobotPoseToCache_ :640-697 with the epipolar test
        if cache["cleaned"]:
            to_add = [f for f in factors if self.is_inlier(f, cache["factors"])]
            if to_add:
                cache["factors"][frame] = to_add
        else:
            cache["factors"][frame] = list(factors)
            cleaned = {}
            for fr in sorted(cache["factors"]):
                for f in cache["factors"][fr]:
                    if self.is_inlier(f, cache["factors"]):
                        cleaned.setdefault(fr, []).append(f)
            if cleaned:
                cache["factors"] = cleaned
                cache["cleaned"] = True

    def parallax_ok(self, cache, min_frame):   # checkMinParallaxRequirements_ :726-800
        frames = [fr for fr in sorted(cache["factors"]) if fr >= min_frame]
        if len(frames) <= 1:
            return False
        frame_ptr, obs_ptr, pix = [0, len(frames)], [0], []
        for fr in frames:
            by_cam = {}
            for f in cache["factors"][fr]:
                by_cam[f[2]] = f[3]
            for cam in sorted(by_cam):
                pix.append(by_cam[cam])
            obs_ptr.append(len(pix))
        prm = obvi_ba.ParallaxParams()
        prm.enforce_min_robot_pose_parallax_requirement = 0
        return bool(self.ba.parallax(frame_ptr, np.ones(len(frames)), self.prob["poses"][frames], obs_ptr, pix, prm)[0])

    def to_graph(self, f):
        self.by_feature.setdefault(f[1], []).append(f)

    def add_frame(self, min_frame, frame):
        feats = {}
        for feat, cam, px in self.obs_by_frame[frame]:
            feats.setdefault(feat, []).append((frame, feat, cam, px))
        for feat, factors in feats.items():
            if feat in self.pending_init:                                                    # :322-345
                cache = self.pending_init[feat]
                self.cache_add(cache, frame, factors)
                if cache["cleaned"]:
                    for fr in sorted(cache["factors"]):
                        for f in cache["factors"][fr]:
                            self.to_graph(f)
                del self.pending_init[feat]
            elif feat in self.added:                                                         # :346-380
                for f in factors:
                    lo = frame - PAST_N                                                      # (unsigned in the reference: no references at all before frame 5)
                    refs = {}
                    if lo >= 0:
                        for g in self.by_feature.get(feat, []):
                            if g[0] > lo:
                                refs.setdefault(g[0], []).append(g)
                    if refs and self.is_inlier(f, refs):
                        self.to_graph(f)
                    elif not refs:
                        cache = self.pending_init.setdefault(feat, {"cleaned": False, "factors": {}})
                        self.cache_add(cache, frame, factors)
            else:                                                                            # :381-412
                cache = self.pending.setdefault(feat, {"cleaned": False, "factors": {}})
                self.cache_add(cache, frame, factors)
                if self.parallax_ok(cache, min_frame):
                    self.initialize(feat)
        if frame - window_start(frame, self.max_frame) > WINDOW:                             # gba_checker :415-447
            for feat in [x for x in self.pending if self.parallax_ok(self.pending[x], min_frame)]:
                self.initialize(feat)

    def initialize(self, feat):
        cache = self.pending.pop(feat)
        for fr in sorted(cache["factors"]):
            for f in cache["factors"][fr]:
                self.to_graph(f)
        self.added.add(feat)

    def run(self):
        for f in range(self.max_frame + 1):
            self.add_frame(0 if f == 0 else window_start(f, self.max_frame), f)
        return sorted((g[0], g[1], g[2]) for fs in self.by_feature.values() for g in fs)


def run_front_end(drv, path, out):
    subprocess.check_call([drv, path, out, "--front-end-only", "--window", str(WINDOW), "--gba-frequency", str(GBA_FREQ)], timeout=600)
    return json.load(open(out))


def check_against_the_sequence(res, prob, bad):
    seq = SequentialFrontEnd(helpers.oracle_ba(), prob)
    assert [tuple(x) for x in res["factors"]] == seq.run()
    fe = res["front_end"]
    assert fe["added"] == len(seq.added) and fe["pending"] == len(seq.pending) and fe["pending_initialized"] == len(seq.pending_init)
    # the questions went out in batches: a handful of calls per frame, not one per question
    assert fe["vote_calls"] <= 4 * len(prob["poses"]) and fe["vote_questions"] >= 50 * fe["vote_calls"]
    # what the gate is for: gross mismatches stay out, the rest gets in; nothing enters before it has been seen with parallax
    entered = {tuple(x) for x in res["factors"]}
    key = lambda i: (int(prob["rp_pose"][i]), int(prob["rp_point"][i]), int(prob["rp_cam"][i]))
    bad_in = sum(key(i) in entered for i in bad)
    good_in = len(entered) - bad_in
    assert bad_in <= 0.25 * len(bad) and good_in >= 0.75 * (len(prob["rp_pose"]) - len(bad)), (bad_in, good_in)
    assert res["features_after_frame"][0] == 0 and all(b >= a for a, b in zip(res["features_after_frame"], res["features_after_frame"][1:]))


def test_lock_step_front_end_equals_the_reference_sequence_on_the_oracle(oracle_driver, outlier_scene, tmp_path):
    prob, path, bad = outlier_scene
    check_against_the_sequence(run_front_end(oracle_driver, path, str(tmp_path / "fe.json")), prob, bad)


@pytest.mark.gpu
def test_lock_step_front_end_on_the_device(driver, oracle_driver, outlier_scene, tmp_path):
    prob, path, bad = outlier_scene
    hip = run_front_end(driver, path, str(tmp_path / "fe_hip.json"))
    check_against_the_sequence(hip, prob, bad)
    ora = run_front_end(oracle_driver, path, str(tmp_path / "fe_ora.json"))
    assert hip["factors"] == ora["factors"] and hip["front_end"] == ora["front_end"] and hip["features_after_frame"] == ora["features_after_frame"]


@pytest.mark.gpu
def test_session_with_the_visual_front_end(driver, outlier_scene, tmp_path):
    """The front end inside the sliding-window session (the frame data adder hands it every new frame): the gross mismatches it keeps
    out do not reach the optimiser, and the trajectory comes out better than with every observation let in."""
    prob, path, _ = outlier_scene
    err = []
    for flag in ([], ["--visual-front-end"]):
        out = str(tmp_path / ("s%d.json" % len(flag)))
        subprocess.check_call([driver, path, out, "--window", str(WINDOW), "--gba-frequency", str(GBA_FREQ)] + flag, timeout=900)
        res = json.load(open(out))
        assert res["ok"]
        err.append(np.linalg.norm(np.array(res["poses"])[:, :3] - prob["gt_poses"][:, :3], axis=1).mean())
    assert err[1] <= err[0] * 1.05, err
