#!/bin/bash
# round-5 measurement batch 3 (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
# converged config #3 with a second fp64 oracle run (one thread fewer) on the host cores, beside everything else
(timeout 3000 python scripts/end_state_table.py 3 64 0 0 1 > $O/end_state_3.txt 2> $O/end_state_3.err) &
BG=$!
# A/B: k_point_pass at 4 wavefronts per SIMD (LDS image capped at 1264 doubles) against the round-4 layout (variants/libobvi_ba_img1408.so)
LEGS="--no-cpu-baseline --no-deterministic-leg --no-end-to-end"
for REP in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 $LEGS > $O/ab_new_$REP.json 2> $O/ab_new_$REP.err
  OBVI_BA_LIBRARY=$R/obvi-slam_amd/csrc/variants/libobvi_ba_img1408.so timeout 300 python bench.py --steps 20 --warmup 5 $LEGS > $O/ab_old_$REP.json 2> $O/ab_old_$REP.err
done
python - <<'PY' > $O/ab_point_pass.txt
import json, glob, os
O = os.environ.get("O", "gpurun_out/r05c")
for tag in ("new", "old"):
    for f in sorted(glob.glob("gpurun_out/r05c/ab_%s_*.json" % tag)):
        try:
            b = json.loads(open(f).read().strip().splitlines()[-1]); k = b["kernels"]
            print(tag, os.path.basename(f), "ms/step %.4f" % b["ms_per_step"], "point_pass avg %.1f in situ %.1f us" % (k["point_pass"]["avg_us"], k["point_pass"].get("in_situ_us", 0)), "schur in situ %.1f" % k["schur_window"].get("in_situ_us", 0))
        except Exception as e:
            print(tag, f, "ERR", e)
PY
cat $O/ab_point_pass.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"; tail -6 $O/t_all.log
timeout 900 python bench.py --config 5 --sessions 16 --steps 10 --warmup 2 > $O/cfg5_fused_s16.json 2> $O/cfg5_fused_s16.err; echo "cfg5 rc=$?"
wait $BG
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"; tail -3 $O/bench_cfg3.err
