"""GPU (-m gpu): the contract of `python bench.py` on one GPU -- ONE JSON line on stdout and nothing else, the fields the round driver reads, the
two objects the measurement rules ask for (`roofline` for the dominant kernel, `cpu_baseline` with the oracle as the timed CPU port), and the
round-5 additions (end state against the oracle, host thread accounting, collective latencies on a one-rank communicator)."""
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_default_workload_line():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-end-to-end", "--no-deterministic-leg"],
                         capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["metric"] == "global-BA LM iterations/s" and b["unit"] == "LM iterations/s" and b["n_gpus"] == 1 and b["steps"] == 4 and b["warmup"] == 1
    assert b["higher_is_better"] is True and b["scaling"] == "weak" and b["vs_baseline"] is None and b["dtype"] == "f64" and b["data"] == "synthetic"
    assert b["value"] > 0 and abs(b["value"] * b["ms_per_step"] / 1e3 - 1.0) < 1e-6                     # iterations / s and ms per iteration of the same timed region
    cfg = b["config"]
    assert cfg["workload"].startswith("global-BA 2000 KF / 200 objects / 300k features") and cfg["keyframes"] == 2000 and cfg["objects"] == 200 and cfg["steps_done"] == 4
    r = b["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["kernel"] in b["kernels"] and r["step"]["alg_bytes"] > 1e9
    k = b["kernels"]["k_update_potrf"]
    assert abs(k["launches_per_step"] - (b["config"].get("chol_levels", 31) - 1)) <= 1.0 or k["launches_per_step"] == pytest.approx(30.0, abs=0.01)   # levels - 1 launches per factorising step
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "LM iterations/s" and c["value"] > 0 and c["cores"] >= 1 and "LM iterations of the same problem" in c["sample"]
    e = c["end_state_vs_oracle"]
    assert e["same_excluded_sets"] and e["same_lm_sequence"] and e["final_cost_rel"] < 1e-10 and e["pose_translation_max_m"] < 1e-9 and e["pose_rotation_max_rad"] < 1e-9
    h = b["host"]
    assert 1 <= h["host_threads_used"] <= 16 and 1 <= h["usable_cpus"] <= h["cpu_count"]
    cu = cfg["collectives_us"]
    assert isinstance(cu, dict) and {"shared_blocks", "shared_tail", "scalars"} <= set(cu["config_5_200_shared_objects"]) and cu["config_5_200_shared_objects"]["shared_tail"]["doubles"] == 1037696


def test_the_host_mirror_legs_of_the_bench_line():
    """`end_to_end_cpp` and `sliding_window_session_cpp` (bench.py, config 3 only; the contract test above skips them for time): the functions themselves on small
    scenes -- the driver runs, the records parse, the fields the docs quote are there."""
    import importlib.util
    sys.path[:0] = [os.path.join(helpers.ROOT, "obvi-slam_amd", "python")]
    import synth
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(helpers.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = bench.sliding_window_session_cpp(synth, 0, frames=60, features=4000, objects=4)
    assert "error" not in s, s
    assert s["optimisations"] >= 100 and s["lm_iterations"] > s["optimisations"] and s["frames_per_s"] > 0 and s["next_window_planned_beside_the_solve"] is True
    c = bench.concurrent_sessions_cpp(synth, 0, k=3, frames=60, features=4000, objects=4)
    assert "error" not in c, c
    assert c["sessions_3"]["frames_per_s"] > 0 and c["sessions_1"]["frames_per_s"] > 0 and c["sessions_1_planned_ahead"]["frames_per_s"] > 0 and c["speedup_vs_one_serial_session"] > 0.5 and c["speedup_vs_best_single_session"] > 0.3
    prob = synth.make_problem(P=120, L=6000, O=6, seed=9, const_poses=1, min_obj_obs=10)
    e = bench.end_to_end_cpp(prob, 0)
    assert "error" not in e, e
    assert e["ok"] and e["run_full_optimization_ms"] > 0 and e["outside_lm_steps_ms"] is not None and 0 < e["outside_lm_steps_ms"] < e["run_full_optimization_ms"] and e["planned_beside_pgo_stage"] is True
