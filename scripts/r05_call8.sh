#!/bin/bash
# the next window planned beside the solve: the 300-frame session with and without it
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_structure.py -x -q -m gpu -k "planned_ahead or update_state" 2>&1 | tail -5
for rep in 1 2 3; do
  for mode in 0 1; do
    echo "== plan ahead $mode, run $rep"
    OBVI_HOST_PLAN_AHEAD=$mode OBVI_HOST_TIMING=1 python scripts/session_time.py 2>&1 | grep -E "wall|planned ahead|solveOptimization|runOptimizationIteration|driver:"
  done
done 2>&1 | tee gpurun_out/plan_ahead_session.txt
echo "== spin 0"
OBVI_HOST_BESIDE_SPIN_US=0 OBVI_HOST_TIMING=1 python scripts/session_time.py 2>&1 | grep -E "wall|planned ahead"
echo "== 8 host threads"
OBVI_HOST_THREADS=8 OBVI_HOST_TIMING=1 python scripts/session_time.py 2>&1 | grep -E "wall|planned ahead"
OBVI_HOST_PLAN_AHEAD=1 OBVI_HOST_TIMING=1 OBVI_API_TIMING=1 python scripts/session_time.py > gpurun_out/plan_ahead_session_api.txt 2>&1
grep "api timing" gpurun_out/plan_ahead_session_api.txt | cut -c1-200
