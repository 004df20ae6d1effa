#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05h
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
timeout 900 python -m pytest tests/test_gpu_shared_objects.py tests/test_gpu_session_groups.py -q -m gpu > $O/t_shared.log 2>&1; echo "shared rc=$?"; tail -3 $O/t_shared.log | head -1
for S in 16 4 2; do
  timeout 600 python bench.py --config 5 --sessions $S --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_fused_s$S.json 2> $O/cfg5_fused_s$S.err
  OBVI_TAIL_SPATIAL=0 timeout 600 python bench.py --config 5 --sessions $S --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_fused_s${S}_indexorder.json 2> $O/cfg5_fused_s${S}_indexorder.err
done
for K in 4 8; do
  timeout 600 python bench.py --config 4 --windows-per-gpu $K --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg4_fused_k$K.json 2> $O/cfg4_fused_k$K.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05h/c*.json")):
    try:
        b = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); k = b["kernels"]
        print("%-36s joint ms %7.3f speedup %.2f value %7.1f | k_update_potrf %6.1f us x %2.0f  trsm %5.1f  backward %5.1f x %2.0f" % (os.path.basename(f), b["ms_per_step"], b["concurrency"]["speedup_vs_serial"], b["value"],
              k["k_update_potrf"]["avg_us"], k["k_update_potrf"]["launches_per_step"], k["k_trsm"]["avg_us"], k["k_backward"]["avg_us"], k["k_backward"]["launches_per_step"]))
    except Exception as e:
        print(f, "ERR", e)
PY
