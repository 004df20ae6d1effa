#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_structure.py tests/test_host_mirror.py tests/test_lockstep_session.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do
  for mode in 0 1; do
    echo "== plan ahead $mode, run $rep"
    OBVI_HOST_PLAN_AHEAD=$mode OBVI_HOST_TIMING=1 python scripts/session_time.py 2>&1 | grep -E "wall|planned ahead|planned beside|solveOptimization|driver:"
  done
done 2>&1 | tee gpurun_out/plan_ahead_session.txt
for mode in 0 1; do
  echo "== e2e config 3, plan ahead $mode"
  OBVI_HOST_PLAN_AHEAD=$mode python scripts/e2e_cpp.py 2000 300000 200 3 2>&1 | grep -E "^run|planned|beside|driver:|obvi_ba_create|pose-graph \+ object"
done 2>&1 | tee gpurun_out/plan_ahead_e2e.txt
