#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05k
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
LEGS="--no-cpu-baseline --no-deterministic-leg --no-end-to-end"
for REP in 1 2; do
  for P in none 1 -1; do
    if [ $P = none ]; then unset OBVI_SIDE_PRIORITY; else export OBVI_SIDE_PRIORITY=$P; fi
    timeout 300 python bench.py --steps 20 --warmup 5 $LEGS > $O/prio_${P}_$REP.json 2> $O/prio_${P}_$REP.err
  done
done
unset OBVI_SIDE_PRIORITY
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05k/prio_*.json")):
    b = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); k = b["kernels"]; p = b["phases_ms_avg"]
    print("%-22s ms/step %.4f strips in situ %.1f  side: pose %.3f small %.3f" % (os.path.basename(f), b["ms_per_step"], k["schur_window"]["in_situ_us"], p["pose_pass"], p["small_factors"]))
PY
