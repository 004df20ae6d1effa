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
