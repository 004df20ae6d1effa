#!/usr/bin/env python
"""bench.py -- global bundle adjustment on MI355X: LM iterations / second.

One "step" = one Levenberg-Marquardt iteration of the global BA (linearisation of every active
factor, Schur complement, exact reduced solve, back-substitution, trial-point cost) on the
synthetic problem of BASELINE.json configs[2]: 2 000 keyframes / 200 ellipsoid objects /
300 000 features (seed 20241008+3).  Inputs are uploaded through the C ABI
before the timed region starts, so they are resident in HBM.  Tolerances are set to zero in
the timed solve so exactly K iterations run.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` is honoured: with N > 1 and no launcher environment (WORLD_SIZE unset) the script re-launches itself under
torch.distributed.run with N ranks; under a launcher it refuses to run if WORLD_SIZE != N.

N > 1 (BASELINE.json configs[3]): the workload that shards is independent 500-keyframe local-BA windows, one per
GPU, SHARING object blocks: per LM step an RCCL all-reduce of the shared objects' J^T J blocks / gradients, of the trailing
shared tiles of the reduced system, and of the scalar block (compiled ncclAllReduce forwarder libobvi_rccl.so on the handle's
own stream; `--hook torch` routes the same callback through torch.distributed instead).  That is the default for N > 1
(`--config 4`); weak scaling: value = total LM iterations of all ranks / max-over-ranks time.  A monolithic global BA does not
shard without exchanging the reduced system: `--config 3` with N > 1 runs N independent replicas, no data-path collective.

`--config 5` (BASELINE.json configs[4], SURVEY.md 8e "config #5: 2 sessions per GPU"): `--sessions S` (default 16) sessions of 500 keyframes /
50 000 features over ONE 200-object map, S / N per GPU, solved JOINTLY: every session is a handle (one host thread each), a rank's
handles meet in obvi_rccl_group_* (libobvi_rccl.so: device sum over the rank's handles, ONE ncclAllReduce per collective, fan back), and
all S sessions take the same LM decisions.  Strong scaling: the job is the same S sessions for every N; value = S x LM iterations of the
joint solve / max-over-ranks time.  `--chain` instead runs the reference's own semantics of that config on one GPU: the sessions one
after the other, each starting from the long-term map of its predecessor (ltm_trajectory_sequence_executor.py:45-92).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python"))

import numpy as np  # noqa: E402

CONFIGS = {
    2: dict(name="local-BA 500 KF / 50k features, reprojection only", P=500, L=50000, O=0, const_poses=5),
    3: dict(name="global-BA 2000 KF / 200 objects / 300k features", P=2000, L=300000, O=200, const_poses=1),
    31: dict(name="(diagnostic) config 3 without objects", P=2000, L=300000, O=0, const_poses=1),
    4: dict(name="500-KF local-BA windows, one per GPU, sharing 25 objects (RCCL all-reduce of the shared object blocks)", P=500, L=50000, O=25, const_poses=5, shared=True),
    5: dict(name="config 5: %d concurrent sessions x (500 KF / 50k features) over one 200-object map, %d per GPU, joint solve (RCCL all-reduce of the shared map's blocks)",
            P=500, L=50000, O=200, const_poses=1, shared=True, sessions=16),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MATRIX_PEAK_TF = 78.6   # MI355X datasheet FP64 matrix (== FP64 vector) rate; not in the guide's table


def solver_params(obvi_ba, iters):
    # global_ba_iteration_params of config/base7a_2_fallback.json with zero tolerances
    return obvi_ba.SolverParams(max_num_iterations=iters, allow_non_monotonic_steps=True, function_tolerance=0.0,
                                gradient_tolerance=0.0, parameter_tolerance=0.0, initial_trust_region_radius=100.0,
                                max_trust_region_radius=1e4)


def profile_manifest():
    """profiles/manifest.json (scripts/profile_round.sh): which build the committed profiles measured.  Returns (manifest, stale):
    stale is a reason string when the kernel sources this run uses are not the ones the profiles were taken on."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from source_sha import kernel_source_sha
    path = os.path.join(ROOT, "profiles", "manifest.json")
    try:
        man = json.load(open(path))
    except (OSError, ValueError):
        return None, "profiles/manifest.json missing"
    now = kernel_source_sha()
    if man.get("kernel_source_sha") != now:
        return man, "profiles/ were measured on kernel sources %s, this run uses %s: re-run scripts/profile_round.sh" % (man.get("kernel_source_sha"), now)
    return man, None


def pmc_traffic(kernel, man):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    --pmc runs, corrected as MI355X_MICROARCH.md prescribes; scripts/pmc_summary.py).  None if not measured."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", man["files"]["pmc_traffic"]))).get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch")
    except (ValueError, OSError, KeyError, TypeError):
        return None


def rocprof_avg_us(kernel, man):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this command; kernel only,
    without the launch boundary that a HIP-event bracket includes."""
    try:
        import csv
        import re
        pat = re.compile(r"\bk_" + re.escape(kernel[2:] if kernel.startswith("k_") else kernel) + r"[(<]")   # phase names lack the k_ prefix
        for row in csv.DictReader(open(os.path.join(ROOT, "profiles", man["files"]["kernel_stats"]))):
            if pat.search(row["Name"]):
                return round(float(row["AverageNs"]) / 1e3, 2)
    except (ValueError, OSError, KeyError, TypeError):
        pass
    return None


def pmc_step_bytes(man, table):
    """HBM bytes of one LM step by the committed PMC passes: bytes per launch x launches per step, over every kernel that has both."""
    if man is None:
        return None
    try:
        per = json.load(open(os.path.join(ROOT, "profiles", man["files"]["pmc_traffic"]))).get("kernels", {})
    except (ValueError, OSError, KeyError, TypeError):
        return None
    alias = {"point_pass": "k_point_pass", "pose_pass": "k_pose_pass", "schur_window": "k_schur_window", "schur_blocks": "k_schur_blocks", "point_backsub": "k_backsub_apply",
             "cost": "k_cost", "small_factors": "k_small_lin_lanes", "k_trsm": "k_trsm", "k_update_potrf": "k_update_potrf", "k_backward": "k_backward", "k_potrf": "k_potrf"}
    tot, seen = 0.0, 0
    for name, row in table.items():
        e = per.get(alias.get(name, name))
        if e and e.get("hbm_bytes_per_launch") is not None:
            tot += e["hbm_bytes_per_launch"] * row["launches_per_step"]
            seen += 1
    return tot if seen else None


def sq_counters(kernel, man):
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", man["files"]["sq_counters"])))["kernels"].get(kernel if kernel.startswith("k_") else "k_" + kernel)
        return {k: round(e[k], 4) for k in ("mfma_busy_frac", "active_frac", "wait_frac", "issue_stall_frac") if k in e} if e else None
    except (ValueError, OSError, KeyError, TypeError):
        return None


def parity_vs_oracle(oracle_its, legs):
    """HIP against the oracle on THIS workload, for the LM steps the oracle's bounded run made (cpu_baseline leg: the oracle is the
    checker here, as in tests/test_gpu_parity.py::test_config3_follows_the_oracle_for_two_steps).  Per leg (default / deterministic
    handle) and step: relative difference of cost, step norm, relative decrease, and whether the accept flags agree."""
    def rel(a, b):
        return abs(a - b) / max(abs(b), 1e-300)
    out = {}
    for name, its in legs.items():
        if not its:
            continue
        rows = []
        for k in range(min(len(its), len(oracle_its))):
            g, o = its[k], oracle_its[k]
            row = {"step": k, "cost_rel": rel(g.cost, o.cost)}
            if k > 0:
                row.update(step_norm_rel=rel(g.step_norm, o.step_norm), relative_decrease_abs=abs(g.relative_decrease - o.relative_decrease),
                           same_decision=bool(g.step_is_successful == o.step_is_successful and g.step_is_valid == o.step_is_valid),
                           radius_rel=rel(g.trust_region_radius, o.trust_region_radius))
            rows.append(row)
        out[name] = rows
    out["note"] = ("step 0 = initial cost (same arithmetic: 1e-12); step 1 = one reduced solve from identical values (measured 1e-9 .. 1e-8: the "
                   "conditioning of S times round-off); from step 2 on the difference of step 1 is amplified by this ill-conditioned problem "
                   "(free gauge, zero tolerances, non-monotonic steps) -- DESIGN.md section 6 has the extended-precision arbiter")
    return out


def cpu_baseline(prob, budget_iters=3, legs=None):
    """The CPU oracle on the same problem, on the host cores of this box: min(20, hardware threads) threads -- 20 is the reference's
    own Solver::Options::num_threads (object_pose_graph_optimizer.h:662) -- bounded to a few LM iterations.  The rate is taken
    from the per-iteration records of iterations 1..K (the initial evaluation, which also first-touches the oracle's
    linearisation records, is iteration 0 and is reported separately)."""
    import ctypes
    import obvi_ba
    import synth
    lib = os.path.join(ROOT, "oracle", "libobvi_oracle.so")
    if not os.path.exists(lib):
        return None
    threads = max(1, min(20, os.cpu_count() or 1))
    ctypes.CDLL(lib).oracle_set_threads(ctypes.c_int32(threads))
    o = obvi_ba.BundleAdjuster(library=lib, prefix="oracle_")
    synth.upload(o, prob)
    t0 = time.time()
    s = o.solve(solver_params(obvi_ba, budget_iters))
    dt = time.time() - t0
    its = o.iterations()
    steady = sum(i.iteration_time_in_seconds for i in its[1:])
    n = max(1, len(its) - 1)
    ceres = ceres_harness(prob, budget_iters, threads)
    return {"parity_vs_oracle": parity_vs_oracle(its, legs or {}), "value": n / steady, "unit": "LM iterations/s", "cores": threads, "kind": "port",
            "sample": "%d LM iterations of the same problem (oracle/libobvi_oracle.so, fp64, %d host threads of %d; %.1f s wall incl. %.1f s "
                      "initial evaluation)" % (n, threads, os.cpu_count() or 1, dt, dt - steady),
            "ms_per_step": 1e3 * steady / n, "reference_ceres": ceres}


def end_state_vs_oracle(obvi_ba, synth, device, threads):
    """LM END STATE on the HIP path against the CPU oracle, from a case the oracle finishes in seconds: BASELINE config #2
    (500 keyframes / 50 000 features, reprojection only, first five poses constant) through the reference's two-phase local-BA block (50 it /
    1e-3, the 10 % cut, 100 it / 1e-4).  The oracle is the checker here (as in tests/test_gpu_end_state.py); what the line carries: whether both
    excluded the same factors and took the same LM sequence, and how far apart the two end states are.  A broken solve shows here as a
    difference of many digits; the chaotic trajectory of the ill-conditioned bench workload (zero tolerances) does not."""
    import ctypes
    import end_state
    lib = os.path.join(ROOT, "oracle", "libobvi_oracle.so")
    if not os.path.exists(lib):
        return None
    prob = synth.make_problem(P=500, L=50000, O=0, seed=20241008 + 2, const_poses=5)
    ctypes.CDLL(lib).oracle_set_threads(ctypes.c_int32(threads))
    legs = {}
    for name, make in (("hip", lambda: obvi_ba.BundleAdjuster(device_id=device)), ("oracle", lambda: obvi_ba.BundleAdjuster(library=lib, prefix="oracle_"))):
        ba = make()
        t0 = time.time()
        legs[name] = end_state.run_two_phase(ba, prob, obvi_ba, synth, block=end_state.LOCAL_BA, polish_iterations=0)
        legs[name]["seconds"] = time.time() - t0
        ba.close()
    c = end_state.compare(legs["hip"], legs["oracle"])
    st = c["state_after_phase_2"]
    return {"workload": "BASELINE config #2 (500 KF / 50k features, reprojection only, 5 constant poses), local_ba_iteration_params: 50 it / 1e-3, 10 % cut, 100 it / 1e-4",
            "same_excluded_sets": c["same_excluded_sets"], "same_lm_sequence": bool(c["phase_1"]["same_lm_sequence"] and c["phase_2"]["same_lm_sequence"]),
            "lm_iterations": {"hip": [legs["hip"]["phase_1"]["iterations"], legs["hip"]["phase_2"]["iterations"]], "oracle": [legs["oracle"]["phase_1"]["iterations"], legs["oracle"]["phase_2"]["iterations"]]},
            "final_cost": {"hip": legs["hip"]["phase_2"]["final_cost"], "oracle": legs["oracle"]["phase_2"]["final_cost"]}, "final_cost_rel": c["phase_2"]["final_cost_rel"],
            "pose_translation_max_m": st["pose_translation_max_m"], "pose_rotation_max_rad": st["pose_rotation_max_rad"], "feature_median_m": st["point_median_m"], "feature_max_m": st["point_max_m"],
            "seconds": {"hip": round(legs["hip"]["seconds"], 2), "oracle": round(legs["oracle"]["seconds"], 2)},
            "bar": "BASELINE.md 2.4 (iii): cost 1e-6 relative, poses 1e-6 m / 1e-6 rad (tests/test_gpu_end_state.py asserts 1e-10 / 1e-9)"}


def ceres_harness(prob=None, budget_iters=3, threads=20):
    """if Ceres is discoverable on this box, __graft_entry__.build() has built oracle/_ref/ceres_harness
    (oracle/ceres_harness: find_package(Ceres QUIET)); it times ceres::Solve with the reference's option block on the same problem.
    Otherwise say so."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ceres_harness")
    if not os.path.exists(exe):
        return "Ceres unavailable on this box (oracle/ceres_harness: find_package(Ceres QUIET) found nothing) -- CPU oracle used as baseline"
    if prob is None:
        return "oracle/_ref/ceres_harness present"
    import subprocess
    import tempfile
    import synth
    try:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "problem.flat")
            synth.dump_flat(prob, path, max_it=budget_iters)
            out = subprocess.run([exe, path, str(threads)], capture_output=True, text=True, timeout=900)
        rep = json.loads(out.stdout)
        n = max(1, rep["num_iterations"] - 1)
        steady = sum(i["iteration_time_in_seconds"] for i in rep["iterations"][1:])
        return {"kind": "reference", "ceres_version": rep["ceres_version"], "threads": threads, "value": n / steady, "unit": "LM iterations/s",
                "ms_per_step": 1e3 * steady / n, "final_cost": rep["final_cost"]}
    except Exception as exc:   # the harness is optional: report, never fail the bench
        return "oracle/_ref/ceres_harness failed: %r" % (exc,)


def end_to_end_global_ba(obvi_ba, synth, prob, device):
    """BASELINE config #3 "run as specified" (SURVEY 8d): one global-BA iteration of the reference's runner on the problem as uploaded --
    the pose-graph stage of runPgoPlusEllipsoids (pose_graph_plus_objects_optimizer.h:23-353: relative-pose factors between consecutive
    CURRENT estimates with generateOdomCov(0.1 x 4) and Huber 5, object factors, no visual features; then the features follow their first
    observing pose, then a features-only reprojection BA) and the two-phase BA of runOptimizationIteration (offline_problem_runner.h:541-894:
    phase I, un-robustified residuals, the 10 % largest distinct values of factor types 0 and 2 excluded, values reverted, phase II) with
    pgo_solver_params / global_ba_iteration_params of config/base7a_2_fallback.json (250 iterations, tolerances 1e-6 / 1e-10 / 1e-8,
    radius 100 / 1e4).  Wall-clock of every ABI call, grouped: upload (set_*), symbolic (the host's plan for a new problem), lm (inside
    obvi_ba_solve minus the plan), selection (evaluate + select_outliers + masks), readback."""
    from scipy.spatial.transform import Rotation as Rot
    t = {"upload": 0.0, "symbolic": 0.0, "lm": 0.0, "selection": 0.0, "readback": 0.0, "host_numpy": 0.0}
    stages = []

    def timed(key, fn, *a, **kw):
        t0 = time.perf_counter()
        r = fn(*a, **kw)
        t[key] += 1e3 * (time.perf_counter() - t0)
        return r

    def solve(name, prm):
        t0 = time.perf_counter()
        ba.evaluate(True, False)            # the plan of a new / changed problem is built here (and one evaluation, ~0.3 ms)
        t_plan = 1e3 * (time.perf_counter() - t0)
        t["symbolic"] += t_plan
        s_ = timed("lm", ba.solve, prm)
        stages.append({"stage": name, "lm_iterations": s_.num_iterations - 1, "initial_cost": s_.initial_cost, "final_cost": s_.final_cost,
                       "plan_ms": round(t_plan, 2), "solve_ms": round(1e3 * s_.total_time_in_seconds, 2), "termination": s_.message.decode()})
        return s_

    params = dict(allow_non_monotonic_steps=True, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
                  initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
    prm = obvi_ba.SolverParams(max_num_iterations=250, **params)
    T0 = time.perf_counter()
    ba = obvi_ba.BundleAdjuster(device_id=device)
    # ---- pose-graph stage
    poses0, pts0 = prob["poses"].copy(), prob["points"].copy()
    th = time.perf_counter()
    Ra, Rb = Rot.from_rotvec(poses0[:-1, 3:6]), Rot.from_rotvec(poses0[1:, 3:6])
    rel_t = Ra.inv().apply(poses0[1:, :3] - poses0[:-1, :3])
    rel_aa = (Ra.inv() * Rb).as_rotvec()
    rel_cov = synth.odom_cov(rel_t, rel_aa, 0.1, 0.1, 0.1, 0.1)
    first_obs = np.full(len(pts0), -1, dtype=np.int64)      # first observing frame per feature (getFirstObservedFrameForFeature)
    order = np.argsort(prob["rp_pose"], kind="stable")[::-1]
    first_obs[prob["rp_point"][order]] = prob["rp_pose"][order]
    seen = first_obs >= 0
    R0 = Rot.from_rotvec(poses0[first_obs[seen], 3:6])
    rel_pos = R0.inv().apply(pts0[seen] - poses0[first_obs[seen], :3])
    t["host_numpy"] += 1e3 * (time.perf_counter() - th)
    P = len(poses0)
    timed("upload", ba.set_cameras, prob["K"], prob["ext"])
    timed("upload", ba.set_poses, poses0, prob["pose_const"])
    timed("upload", ba.set_points, pts0, prob["point_const"])
    timed("upload", ba.set_objects, prob["objects"], prob["object_const"])
    if len(prob["objects"]):
        timed("upload", ba.set_bbox, prob["bb_obj"], prob["bb_pose"], prob["bb_cam"], prob["bb_corners"], prob["bb_cov"], prob["bb_huber"], prob["bb_invalid"])
        timed("upload", ba.set_shape_priors, prob["sp_obj"], prob["sp_mean"], prob["sp_cov"], prob["sp_huber"])
    timed("upload", ba.set_relpose, np.arange(P - 1), np.arange(1, P), rel_t, rel_aa, rel_cov, 5.0)
    solve("pgo", prm)
    poses1 = timed("readback", ba.get_poses)
    th = time.perf_counter()
    pts1 = pts0.copy()
    pts1[seen] = Rot.from_rotvec(poses1[first_obs[seen], 3:6]).apply(rel_pos) + poses1[first_obs[seen], :3]      # the features follow their first observing pose
    t["host_numpy"] += 1e3 * (time.perf_counter() - th)
    # ---- features-only BA (poses and objects constant, reprojection factors only)
    timed("upload", ba.update_points, pts1)
    timed("upload", ba.set_reproj, prob["rp_pose"], prob["rp_point"], prob["rp_cam"], prob["rp_pixel"], prob["rp_sigma"], prob["rp_huber"])
    timed("upload", ba.set_const_flags, np.ones(P, np.uint8), None, np.ones(len(prob["objects"]), np.uint8))
    timed("upload", ba.set_relpose, np.zeros(0), np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 36)), 5.0)
    solve("features_only", prm)
    # ---- two-phase global BA: the odometry factors of the problem are back (frames with few sightings only; here: prob["rl_*"] as generated)
    timed("upload", ba.set_const_flags, prob["pose_const"], None, prob["object_const"])
    if "rl_a" in prob:
        timed("upload", ba.set_relpose, prob["rl_a"], prob["rl_b"], prob["rl_t"], prob["rl_aa"], prob["rl_cov"], prob["rl_huber"])
    timed("upload", ba.snapshot)
    solve("phase_1", prm)
    masks = {}
    for ftype in (0, 2):
        masks[ftype] = timed("selection", ba.select_outliers, ftype, 0.1)
    timed("selection", ba.restore)
    for ftype, (m, _) in masks.items():
        timed("selection", ba.set_active_mask, ftype, m)
    solve("phase_2", prm)
    timed("readback", ba.get_poses); timed("readback", ba.get_points); timed("readback", ba.get_objects)
    wall = 1e3 * (time.perf_counter() - T0)
    ba.close()
    return {"wall_ms": round(wall, 1), "split_ms": {k: round(v, 1) for k, v in t.items()}, "stages": stages,
            "excluded": {"reprojection": int(masks[0][1]), "bbox": int(masks[2][1])},
            "lm_iterations_total": sum(x["lm_iterations"] for x in stages),
            "note": "PGO stage + features-only BA + phase I + outlier selection + phase II through the C ABI with the reference's config values; wall clock "
                    "of this process (Python binding included), inputs start on the host"}


def end_to_end_cpp(prob, device):
    """The same "run as specified" global BA through the C++ host mirror instead of the Python re-enactment above: the scene goes to
    obvi-slam_amd/host/run_offline_ba (the mirror of the reference's runner: frame data adder -> pose graph -> OfflineProblemRunner ->
    runOptimizationIteration's global-BA branch = runPgoPlusEllipsoids + two-phase optimisation with base7a_2_fallback values ->
    ObjectPoseGraphOptimizer::buildPoseGraphOptimization / solveOptimization -> C ABI), `--global-ba`: every frame enters the pose graph,
    then the runner starts at the last frame (run_opt_from_pg_state.cpp:160-312 without the checkpoint file).  Wall clock of that process
    stage by stage; OBVI_API_TIMING gives the time inside each ABI entry point."""
    import re
    import subprocess
    import tempfile
    import scene_io
    exe = os.path.join(ROOT, "obvi-slam_amd", "host", "run_offline_ba")
    if not os.path.exists(exe):
        return {"error": "obvi-slam_amd/host/run_offline_ba not built"}
    with tempfile.TemporaryDirectory() as td:
        scene, out = os.path.join(td, "scene.bin"), os.path.join(td, "out.json")
        t0 = time.perf_counter()
        scene_io.write_scene_binary(prob, scene)
        t_write = time.perf_counter() - t0
        t0 = time.perf_counter()
        r = subprocess.run([exe, scene, out, "--global-ba", "--device", str(device), "--merge-distance", "-1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, OBVI_API_TIMING="1"))
        wall = time.perf_counter() - t0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "run_offline_ba --global-ba failed (rc %d): %s" % (r.returncode, r.stderr[-400:])}
    rep = json.loads(lines[-1])
    api = {}
    for m in re.finditer(r"^api timing: (.+?)\s+([0-9.]+) ms in\s+(\d+) calls", r.stderr, re.M):
        api[m.group(1).strip()] = {"ms": float(m.group(2)), "calls": int(m.group(3))}
    lm_ms = api.get("LM step (submit + wait)", {}).get("ms")
    rep.update(process_wall_ms=round(1e3 * wall, 1), scene_write_ms=round(1e3 * t_write, 1), lm_iterations_total=sum(x["iterations"] for x in rep["records"]),
               post_session_merge="off (--merge-distance -1): the synthetic scene's objects are distinct by construction, some of them closer than the 2 m of base7a_2_fallback.json; with merging on, the session end "
                                  "adds merge rounds with a global BA each (the driver's default since round 5; tests/test_host_mirror.py covers it)",
               outside_lm_steps_ms=None if lm_ms is None else round(rep["run_full_optimization_ms"] - lm_ms, 1),   # the LM iteration count of this run varies (chaotic phase I): this part does not
               planned_beside_pgo_stage=os.environ.get("OBVI_HOST_PLAN_AHEAD", "1") != "0",
               api_timing=api or r.stderr[-1500:],
               note="C++ host mirror end to end: scene_load + pose_graph (frame data adder, all frames) + run_full_optimization (build, upload, symbolic, PGO stage, "
                    "features-only BA, phase I, selection, phase II, read-back); process_wall_ms adds process start, HIP context creation and the result file")
    return rep


def sliding_window_session_cpp(synth, device, frames=300, features=30000, objects=20):
    """A sliding-window session through the C++ host mirror (the reference's per-frame shape: two-phase local BA over 50 frames, global BA every 100 frames, final
    global BA; base7a_2_fallback values): 300 keyframes / 30 000 features / 20 objects, the workload of profiles/r0X_session_300_frames.txt.  Wall clock of the driver
    process and what it did; the next window is planned beside the running solve unless OBVI_HOST_PLAN_AHEAD=0."""
    import subprocess
    import tempfile
    import scene_io
    exe = os.path.join(ROOT, "obvi-slam_amd", "host", "run_offline_ba")
    if not os.path.exists(exe):
        return {"error": "obvi-slam_amd/host/run_offline_ba not built"}
    prob = synth.make_problem(P=frames, L=features, O=objects, seed=4, min_obj_obs=10, bbox_noise=5.0, object_classes=("bench",))
    with tempfile.TemporaryDirectory() as td:
        scene, out, csv = os.path.join(td, "scene.bin"), os.path.join(td, "out.json"), os.path.join(td, "opt.csv")
        scene_io.write_scene_binary(prob, scene)
        t0 = time.perf_counter()
        r = subprocess.run([exe, scene, out, "--window", "50", "--gba-frequency", "100", "--csv", csv, "--device", str(device), "--merge-distance", "-1"], capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "run_offline_ba failed (rc %d): %s" % (r.returncode, r.stderr[-400:])}
        rows = [ln.split(",") for ln in open(csv).read().strip().split("\n")[1:]]
    its = sum(int(x[12]) for x in rows)
    return {"workload": "%d keyframes / %d features / %d objects, window 50, global BA every 100 frames" % (frames, features, objects), "process_wall_s": round(wall, 3),
            "optimisations": len(rows), "lm_iterations": its, "frames_per_s": round(frames / wall, 1), "solver_time_s": round(sum(float(x[8]) for x in rows), 3),
            "next_window_planned_beside_the_solve": os.environ.get("OBVI_HOST_PLAN_AHEAD", "1") != "0"}


def concurrent_sessions_cpp(synth, device, k=4, frames=300, features=30000, objects=20):
    """k sliding-window sessions at once on this GPU, as k host threads of ONE driver process (run_offline_ba --sessions-in-process k: a runner, a pose graph and device
    handles per session; VERDICT r4 item 5 for window-sized solves), against one such session alone; serial sessions (OBVI_HOST_PLAN_AHEAD=0: from k = 4 on the host's CPUs
    are the limit and a second busy thread per session does not pay).  Frames / s of all k together, process wall clock including start-up."""
    import subprocess
    import tempfile
    import scene_io
    exe = os.path.join(ROOT, "obvi-slam_amd", "host", "run_offline_ba")
    if not os.path.exists(exe):
        return {"error": "obvi-slam_amd/host/run_offline_ba not built"}
    prob = synth.make_problem(P=frames, L=features, O=objects, seed=4, min_obj_obs=10, bbox_noise=5.0, object_classes=("bench",))
    out = {"workload": "%d keyframes / %d features / %d objects per session, window 50, global BA every 100 frames; serial sessions" % (frames, features, objects)}
    with tempfile.TemporaryDirectory() as td:
        scene = os.path.join(td, "scene.bin")
        scene_io.write_scene_binary(prob, scene)
        # three runs: one session serial (the mode the k sessions run in), one session in the driver's default mode (next window planned beside the solve: the
        # best single-session configuration), k sessions.  The speed-up that counts is against the BEST single session.
        for name, n, ahead in (("sessions_1", 1, "0"), ("sessions_1_planned_ahead", 1, "1"), ("sessions_%d" % k, k, "0")):
            t0 = time.perf_counter()
            r = subprocess.run([exe, scene, os.path.join(td, "out_%s.json" % name), "--window", "50", "--gba-frequency", "100", "--device", str(device), "--merge-distance", "-1"]
                               + (["--sessions-in-process", str(n)] if n > 1 else []), capture_output=True, text=True, timeout=900, env=dict(os.environ, OBVI_HOST_PLAN_AHEAD=ahead))
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "run_offline_ba failed (rc %d): %s" % (r.returncode, r.stderr[-400:])}
            out[name] = {"process_wall_s": round(wall, 3), "frames_per_s": round(n * frames / wall, 1)}
    best_single = max(out["sessions_1"]["frames_per_s"], out["sessions_1_planned_ahead"]["frames_per_s"])
    out["speedup_vs_one_serial_session"] = round(out["sessions_%d" % k]["frames_per_s"] / out["sessions_1"]["frames_per_s"], 3)
    out["speedup_vs_best_single_session"] = round(out["sessions_%d" % k]["frames_per_s"] / best_single, 3)
    return out


def collective_latency(torch, ba, comm, dist, args, prob, world, reps=50):
    """Microseconds per all-reduce of the three per-step sizes of the config-4 exchange (shared objects' blocks 56 doubles each; the shared
    tail tiles + right-hand side; the scalar sums + one slot per rank), on the live communicator / process group, back to back on one stream.
    A window step is ~0.4 ms: this is the latency budget the three collectives take out of it."""
    import ctypes
    n_sh = len(prob["objects"])
    ntail = -(-7 * n_sh // 64)
    sizes = {"shared_blocks": 56 * n_sh, "shared_tail": ntail * (ntail + 1) // 2 * 64 * 64 + ntail * 64, "scalars": 9 + world}
    out = {}
    st = torch.cuda.Stream()
    for name, n in sizes.items():
        buf = torch.zeros(max(1, n), dtype=torch.float64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if comm is not None:
            fn = lambda: comm._lib.obvi_rccl_allreduce(comm._c, ctypes.c_void_p(buf.data_ptr()), ctypes.c_int64(n), ctypes.c_int32(0), ctypes.c_void_p(st.cuda_stream))   # noqa: E731
        elif args.oversubscribe:
            hook = dist_util_staged(dist)
            fn = lambda: hook(buf.data_ptr(), n, 0, st.cuda_stream)   # noqa: E731
        else:
            def fn():
                with torch.cuda.stream(st):
                    dist.all_reduce(buf)
        for _ in range(5):
            fn()
        st.synchronize()
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        st.synchronize()
        out[name] = {"doubles": n, "us_stream": round(1e3 * e0.elapsed_time(e1) / reps, 2), "us_host": round(1e6 * (time.perf_counter() - t0) / reps, 2)}
    return out


def dist_util_staged(dist):
    import dist_util
    return dist_util.staged_allreduce(dist)


def kernel_table(pst, level1, level2, peaks):
    """Per-kernel rows of the JSON line from the two instrumented solves (obvi_ba_set_profiling 1: one HIP event pair per phase, the timed
    schedule; 2: an event after every launch of the tile Cholesky, one stream): average launch, launches per factorising step, algorithmic
    GB/s (SURVEY 8d split of B_step) or TFLOP/s with the fraction of the public and of the measured peak.  level1 / level2 = (before,
    after) snapshots of obvi_ba_get_kernel_times.  Returns (table, phases)."""
    n_r, n_b = pst["reproj_active"], pst["bbox_active"]
    t3 = 64.0 ** 3
    # algorithmic HBM bytes (SURVEY 8d split of B_step) or flops of ONE launch of each kernel
    hbm = {
        "point_pass": n_r * (32.0 + 144.0),            # read observations, write Z (6x3 fp64)
        "pose_pass": n_r * 32.0,
        "schur_window": n_r * 144.0,                   # every Z record once
        "schur_blocks": None,                          # far pairs only: no per-observation figure
        "point_backsub": n_r * 144.0,
        "cost": n_r * 32.0,
        "small_factors": n_b * (168.0 + 672.0),
    }
    # flops of ONE launch = the factorisation's total / the launches of that kernel per factorisation (chol_levels levels: a k_trsm and a
    # k_update_potrf launch for every level but the last; the first level's potrf is its own launch)
    nlaunch = max(pst["chol_levels"] - 1.0, 1.0)
    flops = {
        "k_trsm": pst["trsm_jobs"] * t3 / nlaunch,
        # updates of a level + factor and inverse of the next level's diagonal tiles, one launch
        "k_update_potrf": (pst["update_jobs"] * 2.0 * t3 + pst["tiles_per_dim"] * (2.0 * t3 / 3.0)) / nlaunch,
    }

    def delta(k1_, k0_):
        out = {}
        for name in k1_:
            ms = k1_[name][0] - k0_.get(name, (0.0, 0))[0]
            n = k1_[name][1] - k0_.get(name, (0.0, 0))[1]
            if n > 0:
                out[name] = {"ms_total": ms, "launches": n, "ms_avg": ms / n}
        return out
    phases = delta(level1[1], level1[0])
    kern = delta(level2[1], level2[0])
    kern.pop("cholesky_solve", None)           # replaced by its kernels
    # steps of the instrumented solve that FACTORISE: the submission at the iteration cap linearises only (point pass, pose pass, small factors,
    # diagonal blocks and nothing else), but the library counts a phase record for every submission -- so the factorising steps are counted
    # from the launches of the tile Cholesky itself (levels - 1 launches of k_update_potrf per factorisation), and the phases that exist
    # only in a factorising step are averaged over those (round 4 divided them by the submissions: 5 % low at 20 steps)
    solve_only = ("schur_window", "schur_blocks", "point_backsub", "apply_step", "cost")
    if "k_update_potrf" in kern and pst["chol_levels"] > 1:
        steps_prof = max(1, int(round(kern["k_update_potrf"]["launches"] / (pst["chol_levels"] - 1.0))))
    else:
        steps_prof = max(1, (kern["point_backsub"]["launches"] if "point_backsub" in kern else kern["point_pass"]["launches"]) - 1)
    for tab in (phases, kern):
        for name in solve_only:
            if name in tab and tab[name]["launches"] > steps_prof:
                tab[name]["launches"] = steps_prof
                tab[name]["ms_avg"] = tab[name]["ms_total"] / steps_prof
    table = {}
    for name, v in kern.items():
        row = {"avg_us": round(1e3 * v["ms_avg"], 2), "launches_per_step": round(v["launches"] / steps_prof, 1), "ms_per_step": round(v["ms_total"] / steps_prof, 4)}
        if name in phases and name in ("schur_window", "point_pass", "point_backsub", "cost"):
            # the same kernel in the configuration that is TIMED (main stream of the uninstrumented schedule: the Schur kernel then runs
            # beside the side stream's pose pass / small factors): start-to-next-phase on the main stream, level-1 events
            row["in_situ_us"] = round(1e3 * phases[name]["ms_avg"], 2)
        t_us = row.get("in_situ_us", row["avg_us"])
        if hbm.get(name):
            row.update(bound="hbm", achieved=round(hbm[name] / (t_us * 1e-6) / 1e9, 1), unit="GB/s")
            row["frac"] = round(row["achieved"] / HBM_PEAK_GBS, 4)
            row["frac_measured"] = round(row["achieved"] / peaks["hbm_triad_gbs"], 4)
        elif name in flops:
            row.update(bound="mfma", achieved=round(flops[name] / (v["ms_avg"] * 1e-3) / 1e12, 3), unit="TFLOP/s")
            row["frac"] = round(row["achieved"] / FP64_MATRIX_PEAK_TF, 4)
            row["frac_measured"] = round(row["achieved"] / peaks["mfma_f64_issue_tflops"], 4)
        table[name] = row
    return table, phases


def guarded(fn, seconds, what):
    """Runs fn() on a thread and waits at most `seconds` for it.  A solve whose collective never meets its partner on another rank (ranks that
    issued different sequences, a rank that died) would wait inside the device queue for ever: the bench then says so and exits non-zero
    instead of hanging the launcher (the issue-order comparison used to run only AFTER the timed solve)."""
    import threading
    box = {}

    def run():
        try:
            box["out"] = fn()
        except BaseException as e:             # noqa: BLE001 -- re-raised on the caller's thread
            box["err"] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        print("bench.py: %s did not finish within %.0f s -- a collective without its partner (ranks out of step or a rank gone)?  Giving up." % (what, seconds), file=sys.stderr, flush=True)
        os._exit(3)
    if "err" in box:
        raise box["err"]
    return box.get("out")


def check_issue_order(comm, dist, dist_util, issue_log, when):
    """Every rank must have issued the same collectives in the same host order (one communicator, two streams).  Compared over the launcher's own
    process group / the communicator's host path, at a quiescent point; a difference is an error, reported with where it was seen."""
    if comm is not None:
        calls, same = comm.sequence()[0], comm.same_issue_order()
    else:
        calls, same = issue_log.calls, dist_util.same_issue_order(dist, issue_log.calls, issue_log.digest())
    if not same:
        raise SystemExit("bench.py: the ranks issued different sequences of collectives (%s)" % when)
    return {"collectives_issued": calls, "same_on_every_rank": bool(same), "checked": when}


class c_stdout_to_stderr:
    """RCCL prints a version banner on the C library's stdout when a communicator is formed; this script's stdout carries ONE JSON line and nothing else.
    Inside the block file descriptor 1 is the process's stderr (C stdio flushed on both sides)."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def numa_node_of_caller():
    """NUMA node of the CPU this thread runs on (the library's pool threads are the caller's neighbours with OBVI_HOST_AFFINITY=1), or None."""
    try:
        cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else None
        if cpu is None:
            import ctypes
            cpu = ctypes.CDLL(None).sched_getcpu()
        for node in sorted(os.listdir("/sys/devices/system/node")):
            if node.startswith("node") and os.path.exists("/sys/devices/system/node/%s/cpu%d" % (node, cpu)):
                return int(node[4:])
    except Exception:      # noqa: BLE001 -- a diagnostic
        pass
    return None


def form_rccl_comm(args, torch, dist, dist_util, rank, world, local_rank, ddev):
    """The job's communicator of libobvi_rccl.so (`--hook rccl`): rank 0's ncclUniqueId travels over the launcher's process group, the data
    path then never touches Python.  Every rank must succeed, or every rank falls back to `--hook torch` (args.hook is rewritten): one rank
    without the compiled hook must not leave the others waiting in a collective.  Returns (comm or None, {"libobvi_rccl": v, "torch": v})."""
    def all_ranks_ok(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=ddev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)
    comm, rccl_versions = None, None
    if args.hook == "rccl":
        my_id, err = None, None
        try:
            # libobvi_rccl.so resolves /opt/rocm/lib/librccl while torch has mapped its own librccl.so: only the same build on both sides
            ok, ours, theirs = dist_util.RcclComm.check_against_torch()
            rccl_versions = {"libobvi_rccl": ours, "torch": theirs}
            if not ok:
                raise RuntimeError("libobvi_rccl.so resolves RCCL %s, torch.distributed uses %s" % (ours, theirs))
            my_id = dist_util.RcclComm.unique_id()      # (every rank: shows that the library loads; rank 0's id is the job's)
        except Exception as e:                          # noqa: BLE001 -- reported below, the run goes on with the other hook
            err = e
        if not all_ranks_ok(err is None):
            if rank == 0:
                print("bench.py: libobvi_rccl.so unusable on some rank (%r): falling back to --hook torch" % (err,), file=sys.stderr)
            args.hook = "torch"
    if args.hook == "rccl":
        ids = [my_id if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            with c_stdout_to_stderr():
                comm = dist_util.RcclComm(rank, world, local_rank, unique_id=ids[0])
        except Exception as e:                          # noqa: BLE001
            comm, err = None, e
        if not all_ranks_ok(comm is not None):
            if comm is not None:
                comm.close()
            comm = None
            if rank == 0:
                print("bench.py: the RCCL communicator of libobvi_rccl.so could not be formed on every rank (%r): falling back to --hook torch" % (err,), file=sys.stderr)
            args.hook = "torch"
    return comm, rccl_versions


def run_sessions(args, torch, obvi_ba, synth, dist_util, rank, local_rank, world, dist, ddev):
    """`--config 5`: S concurrent sessions over one object map, S / world per rank, one joint solve (module docstring).
    `--config 4 --windows-per-gpu K` takes the same route with config 4's windows (500 keyframes / 50 000 features, 25 shared objects): K windows
    per GPU (fused into one problem, or behind the group hook with --group) -- what k window-sized solves together buy on one device."""
    import threading
    cfg = dict(CONFIGS[args.config])
    if args.config == 4:
        cfg.update(name="config 4 with %d windows (500 KF / 50k features, 25 shared objects), %d per GPU, joint solve", sessions=world * args.windows_per_gpu)
    S = args.sessions or cfg["sessions"]
    if args.chain:
        if world != 1:
            raise SystemExit("bench.py --config 5 --chain: the chain is sequential by construction (session s starts from the map of s - 1): one GPU")
        return run_session_chain(args, torch, obvi_ba, synth, S)
    if S % world != 0 or S < world:
        raise SystemExit("bench.py --config 5: --sessions %d is not a multiple of --gpus %d" % (S, world))
    k = S // world
    base = 20241008
    mine = []
    for m in range(k):
        g = rank * k + m                                                # global session index = contributor index of the job
        q = synth.make_problem(P=cfg["P"], L=cfg["L"], O=cfg["O"], seed=dist_util.rank_seed(base, args.config, g), const_poses=cfg["const_poses"], object_seed=base + args.config, min_obj_obs=10)
        if g != 0:                                                      # object-only factors of the shared map: contributor 0 alone
            for key in ("sp_obj", "sp_mean", "sp_cov"):
                q[key] = q[key][:0]
        mine.append(q)
    for q in mine[1:]:
        assert np.array_equal(q["objects"], mine[0]["objects"])
    n_obj = len(mine[0]["objects"])
    first_session = mine[0]
    # How a rank runs its k sessions.  FUSED (default): the k sessions are ONE problem on ONE handle (poses and features of all of them, the map's
    # objects once): every kernel of an LM step covers all k sessions -- the level-scheduled tile Cholesky eliminates the k session subtrees in
    # the same launches, so the dependent chain of a step is as long as ONE session's -- and the rank is one contributor of the exchange.
    # --group: k handles (one host thread, one stream pair each) behind obvi_rccl_group_*: the general mechanism (sessions that arrive as
    # separate handles), k chains side by side and two cross-stream hops per collective (measured: profiles/r05_config5_*.txt).
    fused = not args.group
    k_sessions = k
    if fused:
        mine = [synth.join_problems(mine)] if k > 1 else mine
        k = 1
    issue_log = dist_util.IssueLog()
    comm, rccl_versions, hook = None, None, "group"
    if world > 1:
        comm, rccl_versions = form_rccl_comm(args, torch, dist, dist_util, rank, world, local_rank, ddev)
        hook = "group+" + args.hook
    if comm is not None:
        group = dist_util.RcclGroup(k, comm=comm)
    elif world > 1:
        inner = dist_util.staged_allreduce(dist, issue_log) if args.oversubscribe else dist_util.torch_allreduce(dist, issue_log)
        group = dist_util.RcclGroup(k, inner=inner, rank=rank, world=world, device=local_rank)
    else:
        group = dist_util.RcclGroup(k, device=local_rank)
    if fused:
        hook = hook.replace("group", "fused")
    handles, upload_ms = [], 0.0
    for m, q in enumerate(mine):
        ba = obvi_ba.BundleAdjuster(device_id=local_rank)
        t0 = time.perf_counter()
        synth.upload(ba, q)
        upload_ms += 1e3 * (time.perf_counter() - t0)
        handles.append(ba)

    def in_threads(fn):
        out, err = [None] * k, [None] * k

        def run(m):
            try:
                out[m] = fn(m)
            except Exception as e:                                      # noqa: BLE001 -- re-raised on the main thread
                err[m] = e
        th = [threading.Thread(target=run, args=(m,)) for m in range(k)]
        [t.start() for t in th]
        [t.join() for t in th]
        for e in err:
            if e is not None:
                raise e
        return out

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # what one session costs alone on this GPU (no exchange attached: the shared objects are ordinary objects): the serial yardstick of the rank
    alone = None
    if rank == 0 and not (fused and k_sessions > 1):
        handles[0].evaluate(True, False)
        if args.warmup > 0:
            handles[0].solve(solver_params(obvi_ba, args.warmup))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sa = handles[0].solve(solver_params(obvi_ba, args.steps))
        torch.cuda.synchronize()
        alone = 1e3 * (time.perf_counter() - t0) / max(1, sa.num_iterations - 1)
        synth.upload(handles[0], mine[0])
    if rank == 0 and fused and k_sessions > 1:
        # the yardstick is ONE session: a handle of its own for it
        one = obvi_ba.BundleAdjuster(device_id=local_rank)
        synth.upload(one, first_session)
        one.evaluate(True, False)
        if args.warmup > 0:
            one.solve(solver_params(obvi_ba, args.warmup))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sa = one.solve(solver_params(obvi_ba, args.steps))
        torch.cuda.synchronize()
        alone = 1e3 * (time.perf_counter() - t0) / max(1, sa.num_iterations - 1)
        one.close()
    is_shared = np.ones(n_obj, np.uint8)
    for m, ba in enumerate(handles):
        group.attach(m, ba, is_shared)
    # symbolic phase of the rank's k sessions, all at once (HostPool runs the k plans concurrently)
    t0 = time.perf_counter()
    in_threads(lambda m: handles[m].evaluate(True, False))
    symbolic_ms = 1e3 * (time.perf_counter() - t0)
    limit = 180.0 + 0.5 * S * (args.steps + args.warmup)
    if args.warmup > 0:
        guarded(lambda: in_threads(lambda m: handles[m].solve(solver_params(obvi_ba, args.warmup))), limit, "the warm-up solve")
    if world > 1:
        check_issue_order(comm, dist, dist_util, issue_log, "after the warm-up solve")
    barrier()
    c0 = group.stats()
    t0 = time.perf_counter()
    summ = guarded(lambda: in_threads(lambda m: handles[m].solve(solver_params(obvi_ba, args.steps))), limit, "the timed solve")
    barrier()
    dt = time.perf_counter() - t0
    c1 = group.stats()
    steps_done = min(sm.num_iterations for sm in summ) - 1
    dt, steps_done = dist_util.reduce_timing(dist, ddev, dt, steps_done)
    issue_order = check_issue_order(comm, dist, dist_util, issue_log, "after the warm-up solve and after the timed solve") if world > 1 else None
    # device timings: the same steps twice more with events on (every handle runs them: the collectives need all contributors); rank 0 reads session 0
    for ba in handles:
        ba.set_profiling(1)
    k0 = handles[0].kernel_times()
    in_threads(lambda m: handles[m].solve(solver_params(obvi_ba, args.steps)))
    k1 = handles[0].kernel_times()
    for ba in handles:
        ba.set_profiling(2)
    p0 = handles[0].kernel_times()
    in_threads(lambda m: handles[m].solve(solver_params(obvi_ba, args.steps)))
    p1 = handles[0].kernel_times()
    for ba in handles:
        ba.set_profiling(0)
    if rank == 0:
        pst = handles[0].problem_stats()
        peaks = handles[0].measure_peaks()
        table, phases = kernel_table(pst, (k0, k1), (p0, p1), peaks)
        dom = max(table, key=lambda kk: table[kk]["ms_per_step"])
        d = table[dom]
        ntail = -(-7 * n_obj // 64)
        submissions = steps_done + 1                                    # every LM step + the linearisation-only submission behind the last one
        sizes = {"shared_blocks": 8 * 56 * n_obj, "shared_tail": 8 * (ntail * (ntail + 1) // 2 * 64 * 64 + ntail * 64), "scalars": 8 * (9 + S)}
        stats = [synth.problem_stats(q) for q in mine]
        ms_step = 1e3 * dt / max(steps_done, 1)
        out = {
            "metric": "local-BA LM iterations/s (concurrent sessions sharing one object map, joint solve)", "value": S * steps_done / dt, "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"] % (S, k_sessions), "sessions": S, "sessions_per_rank": k_sessions, "handles_per_rank": k,
                       "mode": "fused: a rank's sessions are one problem on one handle" if fused else "group: one handle per session behind obvi_rccl_group_*",
                       "keyframes": cfg["P"], "features": cfg["L"], "objects": n_obj,
                       "reprojection_obs_per_handle": stats[0]["N_r"], "bbox_obs_per_handle": stats[0]["N_b"], "reduced_rows_per_handle": int(pst["reduced_rows"]),
                       "parallelism": ("%d sessions per GPU (%s) + all-reduce across GPUs" if world > 1 else "%d sessions on one GPU (%s)") % (k_sessions, "one fused problem" if fused else "group sum over k handles"),
                       "rccl_ranks": (comm.world() if comm is not None else (dist.get_world_size() if dist is not None else 1)), "allreduce_hook": hook,
                       "oversubscribed": bool(args.oversubscribe), "rccl_versions": rccl_versions, "collective_issue_order": issue_order, "steps_done": steps_done,
                       "collectives_per_lm_step": round((c1[0] - c0[0] - 1) / max(1, submissions), 2),
                       "collective_bytes": {"per_lm_step": int(8 * (c1[1] - c0[1]) / max(1, submissions)), "by_collective": sizes,
                                            "note": "bytes a rank hands to the inter-rank all-reduce per LM step (the group's sum over its own sessions travels ONCE, whatever "
                                                    "k is); the tail is the dense lower triangle of the shared map's reduced block: every session observes the map, so after "
                                                    "eliminating a session's own poses every pair of map objects is coupled on every rank -- no tile of it is structurally "
                                                    "zero on all ranks (DESIGN.md section 8)"},
                       "final_cost_per_session": [sm.final_cost for sm in summ] * (k_sessions if fused else 1), "termination": summ[0].message.decode()},
            "value_note": "value = sessions x LM iterations of the joint solve / max-over-ranks seconds (every session advances one iteration per joint step; config 4 counts "
                          "its windows the same way); ms_per_step = one JOINT LM step of all %d sessions" % S,
            "concurrency": {"one_session_alone_ms_per_step": round(alone, 4), "sessions_on_this_gpu": k_sessions, "joint_ms_per_step": round(ms_step, 4),
                            "speedup_vs_serial": round(k_sessions * alone / ms_step, 3),
                            "note": "the k sessions of this GPU solved jointly (fused into one problem, or k handles in k host threads with --group) against k x ONE session solved "
                                    "alone, no exchange: what putting window-sized solves together buys on a device that one of them leaves mostly idle"},
            "roofline": {"kernel": dom, "bound": d.get("bound", "hbm"), "achieved": d.get("achieved"), "peak": HBM_PEAK_GBS if d.get("bound", "hbm") == "hbm" else FP64_MATRIX_PEAK_TF,
                         "unit": d.get("unit", "GB/s"), "frac": d.get("frac"), "frac_measured": d.get("frac_measured"), "traffic": None, "avg_launch_us": d["avg_us"],
                         "peaks_measured": {kk: round(v, 2) for kk, v in peaks.items()},
                         "note": "dominant kernel of session 0 on rank 0 by device time per LM step, HIP events in an instrumented solve of the same steps while the rank's other "
                                 "sessions run beside it (their kernels share the device: launch durations include that); traffic: no PMC pass for this workload"},
            "kernels": dict(sorted(table.items(), key=lambda kv: -kv[1]["ms_per_step"])),
            "host": {"upload_ms_all_sessions": round(upload_ms, 1), "symbolic_phase_ms_all_sessions_concurrently": round(symbolic_ms, 1)},
        }
        if not args.no_cpu_baseline:
            q0 = dict(mine[0])
            base_line = cpu_baseline(q0, legs={})
            if base_line is not None:
                base_line["sample"] = "ONE of the %d sessions alone (the joint step is %d of these plus the shared tail): " % (S, S) + base_line["sample"]
                base_line.pop("parity_vs_oracle", None)
            out["cpu_baseline"] = base_line
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
    for ba in handles:
        ba.close()
    group.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


def run_session_chain(args, torch, obvi_ba, synth, S):
    """`--config 5 --chain`: the reference's semantics of the multi-session config (ltm_trajectory_sequence_executor.py:45-92): the sessions
    one after the other on one GPU, session s starting from the long-term map (ellipsoid estimates + marginal covariances as
    IndependentObjectMapFactor priors) that session s - 1 extracted.  A step here = one session (upload, two-phase BA, map extraction)."""
    cfg = CONFIGS[5]
    prm = obvi_ba.SolverParams(max_num_iterations=50, allow_non_monotonic_steps=True, function_tolerance=1e-4, gradient_tolerance=1e-10,
                               parameter_tolerance=1e-8, initial_trust_region_radius=100.0, max_trust_region_radius=1e4)
    g = obvi_ba.BundleAdjuster(device_id=0)
    ltm, rows, iters, t_all = None, [], 0, 0.0
    for sidx in range(S):
        prob = synth.make_problem(P=cfg["P"], L=cfg["L"], O=cfg["O"], seed=1000 + sidx, object_seed=77, const_poses=1, min_obj_obs=10, object_classes=("bench",))
        if ltm is not None:
            prob["objects"][ltm[0]] = ltm[1]
            prob.update(lt_obj=ltm[0].astype(np.uint32), lt_mean=ltm[1], lt_cov=ltm[2].reshape(-1, 49), lt_huber=1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        synth.upload(g, prob)
        s1 = g.solve(prm)
        mask, _ = g.select_outliers(0, 0.1)
        g.set_active_mask(0, mask)
        s2 = g.solve(prm)
        g.set_active_mask(0, np.ones_like(mask))
        ids = np.arange(len(prob["objects"]), dtype=np.uint32)
        cov = g.object_covariances(ids)
        est = g.get_objects()
        dt = time.perf_counter() - t0
        seen = np.abs(cov).max(axis=(1, 2)) > 0
        err = np.linalg.norm(est[seen, :3] - prob["gt_objects"][seen, :3], axis=1)
        rows.append({"session": sidx, "ms": round(1e3 * dt, 2), "lm_iterations": int(s1.num_iterations + s2.num_iterations - 2), "objects_mapped": int(seen.sum()),
                     "centre_error_median_m": round(float(np.median(err)), 4)})
        iters += s1.num_iterations + s2.num_iterations - 2
        t_all += dt
        ltm = (ids[seen], est[seen], cov[seen])
    print(json.dumps({"metric": "local-BA LM iterations/s (sessions chained through the long-term map)", "value": iters / t_all, "unit": "LM iterations/s", "n_gpus": 1,
                      "steps": S, "warmup": 0, "ms_per_step": 1e3 * t_all / S, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "config 5 (chain): %d sessions x (500 KF / 50k features) over one 200-object map, one after the other through the long-term map" % S,
                                 "sessions": S, "step": "one session: upload + two-phase BA + map extraction"}, "sessions": rows}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS), help="default: 3 on one GPU, 4 (windows sharing objects, RCCL all-reduce) on several")
    ap.add_argument("--hook", choices=("rccl", "torch"), default="rccl", help="all-reduce callback of config 4: libobvi_rccl.so (compiled) or torch.distributed")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="N > 1 on ONE GPU: every rank uses device 0, the process group is gloo and the exchange goes through dist_util.staged_allreduce "
                         "(device -> host -> gloo -> device).  Not a measurement of scaling: it exercises the whole N > 1 code path of this script where "
                         "only one GPU exists (tests/test_gpu_shared_objects.py runs it)")
    ap.add_argument("--sessions", type=int, default=None, help="config 5: total number of sessions of the job (default 16; must be a multiple of --gpus)")
    ap.add_argument("--windows-per-gpu", type=int, default=0, help="config 4: K windows per GPU, all sharing the object set, behind the group hook (0: the one-window-per-GPU path)")
    ap.add_argument("--group", action="store_true", help="config 5 / config 4 --windows-per-gpu: one handle per session behind obvi_rccl_group_* instead of fusing a rank's sessions into one problem")
    ap.add_argument("--chain", action="store_true", help="config 5 the reference's way: sessions one after the other through the long-term map, one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-deterministic-leg", action="store_true", help="skip the deterministic-mode timing of the same steps")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end two-phase global BA (config 3, one GPU; about 2 s)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one.  One process per GPU, rendezvous on the loopback address.
        import socket
        import subprocess
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import obvi_ba
    import synth

    import dist_util
    rank, local_rank, world = dist_util.rank_info()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    if args.oversubscribe:
        local_rank = 0
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit("bench.py: rank %d needs device %d but only %d visible (one process per GPU)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.oversubscribe:
            dist.init_process_group(backend="gloo")
            args.hook = "staged-gloo"
        else:
            with c_stdout_to_stderr():
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                warm = torch.zeros(1, device="cuda")
                dist.all_reduce(warm)             # the communicator is formed lazily: here, where its banner goes to stderr
                torch.cuda.synchronize()
    ddev = "cpu" if args.oversubscribe else "cuda"      # where the launcher group's own small tensors live
    if args.config is None:
        args.config = 3 if world == 1 else 4

    if args.config == 5 or (args.config == 4 and args.windows_per_gpu > 0):
        run_sessions(args, torch, obvi_ba, synth, dist_util, rank, local_rank, world, dist, ddev)
        return
    cfg = CONFIGS[args.config]
    shared = bool(cfg.get("shared")) and world > 1
    prob = synth.make_problem(P=cfg["P"], L=cfg["L"], O=cfg["O"], seed=dist_util.rank_seed(20241008, args.config, rank), const_poses=cfg["const_poses"],
                              object_seed=(20241008 + args.config) if cfg.get("shared") else None, min_obj_obs=10)
    if shared and rank != 0:        # object-only factors of a shared object are uploaded by exactly one rank
        for k in ("sp_obj", "sp_mean", "sp_cov"):
            prob[k] = prob[k][:0]
    stats = synth.problem_stats(prob)
    ba = obvi_ba.BundleAdjuster(device_id=local_rank)
    t_up = time.perf_counter()
    synth.upload(ba, prob)          # inputs now resident in HBM
    upload_ms = 1e3 * (time.perf_counter() - t_up)
    rccl_ranks = None
    comm = None
    scaling_baseline = None
    rccl_versions = None
    issue_log = dist_util.IssueLog()
    if shared:
        # the N = 1 point of THIS workload (the driver's N = 1 run is the config-3 headline, another problem): every rank solves its own
        # window alone -- no exchange attached yet, shared objects are ordinary objects -- for the same steps, before the group solve
        ba.evaluate(True, False)
        if args.warmup > 0:
            ba.solve(solver_params(obvi_ba, args.warmup))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s1 = ba.solve(solver_params(obvi_ba, args.steps))
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        tt = torch.tensor([dt1, float(s1.num_iterations - 1)], dtype=torch.float64, device=ddev)
        lst = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(lst, tt)
        per_rank = [(float(x[1]) / float(x[0])) for x in lst]
        scaling_baseline = {"workload": cfg["name"] + ", ONE window on one GPU, no exchange", "steps": int(s1.num_iterations - 1), "ms_per_step": 1e3 * dt1 / max(1, s1.num_iterations - 1),
                            "value_one_gpu": float(np.mean(per_rank)), "per_rank_value": [round(v, 2) for v in per_rank], "unit": "LM iterations/s"}
        synth.upload(ba, prob)      # back to the initial values for the group solve
    if shared:
        is_shared = np.ones(len(prob["objects"]), np.uint8)
        comm, rccl_versions = form_rccl_comm(args, torch, dist, dist_util, rank, world, local_rank, ddev)
        if args.hook == "rccl":
            comm.attach(ba, is_shared)
            rccl_ranks = comm.world()
        else:
            ba.set_shared_objects(is_shared, rank, world)
            ba.set_allreduce(dist_util.staged_allreduce(dist, issue_log) if args.oversubscribe else dist_util.torch_allreduce(dist, issue_log))
            rccl_ranks = dist.get_world_size()
    t_sym = time.perf_counter()
    ba.evaluate(True, False)        # builds the reduced-program bookkeeping / symbolic plan (not timed: the reference times "build" separately)
    t_ev = time.perf_counter()
    ba.evaluate(True, False)
    symbolic_ms = 1e3 * max(0.0, (t_ev - t_sym) - (time.perf_counter() - t_ev))   # first evaluate = symbolic phase + an evaluation

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    limit = 120.0 + 0.25 * (args.steps + args.warmup)      # generous: a config-3 step is 2 ms; what it must catch is a wait that never ends
    if args.warmup > 0:
        guarded(lambda: ba.solve(solver_params(obvi_ba, args.warmup)), limit, "the warm-up solve")
    if shared:
        # BEFORE the timed region: the warm-up went through every collective of the protocol once per step -- ranks that are out of step are
        # found here (and a rank that waits for ever in the warm-up or in the timed solve is found by the time limit), not after the fact
        check_issue_order(comm, dist, dist_util, issue_log, "after the warm-up solve")
    barrier()
    t0 = time.perf_counter()
    summ = guarded(lambda: ba.solve(solver_params(obvi_ba, args.steps)), limit, "the timed solve")
    barrier()
    dt = time.perf_counter() - t0
    ba_timed_iterations = ba.iterations()
    steps_done = summ.num_iterations - 1
    dt, steps_done = dist_util.reduce_timing(dist, ddev, dt, steps_done)
    # one communicator, two streams (first collective of a step on the side stream, the other two on the main stream): legal only if every rank
    # issues the same collectives in the same host order -- compared here, after the timed solve
    issue_order = check_issue_order(comm, dist, dist_util, issue_log, "after the warm-up solve and after the timed solve") if shared else None
    collectives_us = collective_latency(torch, ba, comm, dist, args, prob, world) if shared else None
    if world == 1 and rank == 0:
        # the latency budget of the exchange is needed BEFORE eight ranks exist: the three per-step sizes of config 4 and of
        # config 5 on a ONE-rank communicator of libobvi_rccl.so -- what RCCL's launch path costs per collective before any byte crosses xGMI
        try:
            with c_stdout_to_stderr():
                comm1 = dist_util.RcclComm(0, 1, local_rank, unique_id=dist_util.RcclComm.unique_id())
                collectives_us = {"communicator": "one rank: what RCCL's enqueue costs before any byte crosses xGMI (an in-place all-reduce of one rank moves nothing)"}
                for label, nobj in (("config_4_25_shared_objects", 25), ("config_5_200_shared_objects", 200)):
                    collectives_us[label] = collective_latency(torch, ba, comm1, None, args, {"objects": np.zeros((nobj, 7))}, 1)
                comm1.close()
        except Exception as e:                  # noqa: BLE001 -- a diagnostic: never fails the bench
            collectives_us = "unavailable: %r" % (e,)

    # Device timings come from two more solves of the same K steps, outside the timed region (the timed solve records no
    # events at all): level 1 = one HIP event pair per phase of an LM step, same schedule as the timed solve (side stream on);
    # level 2 = additionally an event after every launch of the tile Cholesky (single stream, costs a few percent).
    ba.set_profiling(1)
    k0 = ba.kernel_times()
    ba.solve(solver_params(obvi_ba, args.steps))
    k1 = ba.kernel_times()
    ba.set_profiling(2)
    p0 = ba.kernel_times()
    ba.solve(solver_params(obvi_ba, args.steps))
    p1 = ba.kernel_times()
    ba.set_profiling(0)

    det_first = None
    if rank == 0:
        pst = ba.problem_stats()
        n_r, n_b = pst["reproj_active"], pst["bbox_active"]
        peaks = ba.measure_peaks()   # triad / copy / read GB/s and the fp64 MFMA rates of THIS device (obvi_ba_measure_peaks)
        table, phases = kernel_table(pst, (k0, k1), (p0, p1), peaks)
        dom = max(table, key=lambda k: table[k]["ms_per_step"])
        d = table[dom]
        man, stale = profile_manifest()
        fresh = man if stale is None else None
        roof = {"kernel": dom, "bound": d.get("bound", "hbm"), "achieved": d.get("achieved"), "peak": HBM_PEAK_GBS if d.get("bound", "hbm") == "hbm" else FP64_MATRIX_PEAK_TF,
                "unit": d.get("unit", "GB/s"), "frac": d.get("frac"),
                "peak_measured": round(peaks["hbm_triad_gbs"] if d.get("bound", "hbm") == "hbm" else peaks["mfma_f64_issue_tflops"], 2), "frac_measured": d.get("frac_measured"),
                "peaks_measured": {k: round(v, 2) for k, v in peaks.items()},
                "traffic": pmc_traffic(dom, fresh),
                "avg_launch_us": d["avg_us"], "launches_per_step": d["launches_per_step"], "rocprof_avg_us": rocprof_avg_us(dom, fresh), "sq": sq_counters(dom, fresh),
                "profiles": {"tag": man.get("tag") if man else None, "kernel_source_sha": man.get("kernel_source_sha") if man else None, "stale": stale},
                "note": "peak = public figure (HBM3E 8 TB/s; fp64 matrix 78.6 TFLOP/s), peak_measured = obvi_ba_measure_peaks in THIS run (HBM: triad over 3 x 1 GiB; "
                        "MFMA: issue rate of v_mfma_f64_16x16x4_f64 with register operands; peaks_measured also has copy / read and the LDS-fed 64x64x64 tile product). "
                        "dominant kernel by device time per LM step; achieved / frac from HIP events around every launch (they include the launch "
                        "boundary, about 3 us) in an instrumented solve of the same steps in THIS run; traffic, rocprof_avg_us and sq come from the "
                        "rocprofv3 passes committed under profiles/ and are null when profiles/manifest.json was not measured on the kernel sources "
                        "this run uses (profiles.stale says why)"}
        ms_step = 1e3 * dt / max(steps_done, 1)
        alg_step = n_r * 496.0 + n_b * 1008.0             # B_step without B_S
        pmc_step = pmc_step_bytes(fresh, table)
        roof["step"] = {"alg_bytes": alg_step, "pmc_bytes": pmc_step, "ms": round(ms_step, 4),
                        "achieved_gbs": round(alg_step / (ms_step * 1e-3) / 1e9, 1),
                        "frac_of_peak": round(alg_step / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "frac_of_measured": round(alg_step / (ms_step * 1e-3) / 1e9 / peaks["hbm_triad_gbs"], 4),
                        "note": "whole LM step: algorithmic bytes N_r 496 + N_b 1008 (SURVEY 8d) over the TIMED ms per step, against the 8 TB/s public figure and the "
                                "triad measured in this run; pmc_bytes = sum over the step's launches of the committed FETCH/WRITE passes (null when stale)"}
        out = {
            "metric": "global-BA LM iterations/s" if not cfg.get("shared") else "local-BA LM iterations/s (windows sharing objects)", "value": dist_util.aggregate_throughput(world, steps_done, dt), "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(steps_done, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"], "keyframes": stats["P"], "features": stats["L"], "objects": stats["O"],
                       "reprojection_obs": stats["N_r"], "bbox_obs": stats["N_b"], "reduced_rows": int(pst["reduced_rows"]),
                       "parallelism": ("windows+allreduce" if shared else "replicas") if world > 1 else "single",
                       "rccl_ranks": rccl_ranks, "allreduce_hook": (args.hook if shared else None), "oversubscribed": bool(args.oversubscribe),
                       "rccl_versions": rccl_versions, "collective_issue_order": issue_order, "collectives_us": collectives_us, "steps_done": steps_done,
                       "final_cost": summ.final_cost, "termination": summ.message.decode(),
                       # the mix of the timed steps: a rejected step costs a full linearisation here -- with the stored Z = rho' Jp^T Jl C^-T the
                       # point pass IS the cheapest re-damp, what a rejected step could skip is the side stream (0.1 ms of contention at this size; EXPERIMENTS round 6)
                       # a hiccup in the timed region shows here, not only in the average: the slowest LM step of the timed solve, and whether the fused level kernel of the
                       # tile Cholesky ever timed out waiting for its jobs (the step is then re-run on the two-launch schedule, which the handle keeps: DESIGN section 4)
                       "slowest_step_ms": round(1e3 * max(it.iteration_time_in_seconds for it in ba_timed_iterations[1:]), 4) if len(ba_timed_iterations) > 1 else None,
                       "median_step_ms": round(1e3 * float(np.median([it.iteration_time_in_seconds for it in ba_timed_iterations[1:]])), 4) if len(ba_timed_iterations) > 1 else None,
                       "potrf_wait_timeouts": int(ba.problem_stats().get("potrf_wait_timeouts", 0)), "fused_potrf_schedule": bool(ba.problem_stats().get("fused_potrf", 1)),
                       "accepted_steps": int(sum(1 for it in ba_timed_iterations[1:] if it.step_is_successful)),
                       "rejected_or_invalid_steps": int(sum(1 for it in ba_timed_iterations[1:] if not it.step_is_successful))},
            "roofline": roof,
            "kernels": dict(sorted(table.items(), key=lambda kv: -kv[1]["ms_per_step"])),
            "phases_ms_avg": {k: round(v["ms_avg"], 4) for k, v in phases.items()},
            # once per problem, outside the timed region: host -> device upload through the binding, and the host's symbolic phase
            "host": {"upload_ms": round(upload_ms, 1), "symbolic_phase_ms": round(symbolic_ms, 1), "host_threads_used": int(pst.get("host_threads", 0)),
                     "usable_cpus": int(pst.get("usable_cpus", 0)), "cpu_count": os.cpu_count(), "numa_node_of_caller": numa_node_of_caller()},
        }
        if scaling_baseline is not None:
            out["scaling_baseline"] = scaling_baseline
            out["weak_scaling_efficiency"] = round(out["value"] / (world * scaling_baseline["value_one_gpu"]), 4)
        if world == 1 and not args.no_deterministic_leg:
            # the same steps in deterministic mode (obvi_ba_options.deterministic: fixed-order sums, one stream; for parity runs)
            bd = obvi_ba.BundleAdjuster(device_id=local_rank, deterministic=True)
            synth.upload(bd, prob)
            bd.solve(solver_params(obvi_ba, 3))      # from the uploaded values, like the oracle's bounded run: the records parity_vs_oracle compares
            bd_first = bd.iterations()
            synth.upload(bd, prob)
            if args.warmup > 0:
                bd.solve(solver_params(obvi_ba, args.warmup))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sd = bd.solve(solver_params(obvi_ba, args.steps))
            torch.cuda.synchronize()
            its_d, its_0 = bd.iterations(), ba_timed_iterations
            det_first = bd_first
            out["deterministic_mode"] = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / max(1, sd.num_iterations - 1), 4), "steps": sd.num_iterations - 1,
                                         # the two modes differ by the order of their sums only: the first steps agree to round-off, later ones as far as this
                                         # ill-conditioned problem amplifies it (zero tolerances, non-monotonic steps: a chaotic trajectory)
                                         "cost_rel_diff_vs_default": {"after_step_%d" % k: abs(its_d[k].cost - its_0[k].cost) / its_0[k].cost for k in (1, 2, 4, 8, len(its_0) - 1) if k < min(len(its_d), len(its_0))}}
            bd.close()
        if world == 1 and args.config == 3 and not args.no_end_to_end:
            out["end_to_end"] = end_to_end_global_ba(obvi_ba, synth, prob, local_rank)
            out["end_to_end_cpp"] = end_to_end_cpp(prob, local_rank)
            out["sliding_window_session_cpp"] = sliding_window_session_cpp(synth, local_rank)
            out["concurrent_sessions_cpp"] = concurrent_sessions_cpp(synth, local_rank)
        if not args.no_cpu_baseline:
            legs = {}
            if world == 1:
                synth.upload(ba, prob)
                ba.solve(solver_params(obvi_ba, 3))
                legs["default"] = ba.iterations()
                if det_first is not None:
                    legs["deterministic"] = det_first
            out["cpu_baseline"] = cpu_baseline(prob, legs=legs)
            if world == 1 and out["cpu_baseline"] is not None:
                out["cpu_baseline"]["end_state_vs_oracle"] = end_state_vs_oracle(obvi_ba, synth, local_rank, out["cpu_baseline"]["cores"])
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        ba.close()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
