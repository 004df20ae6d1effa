#!/bin/bash
mkdir -p gpurun_out
for mode in 0 1; do
  OBVI_HOST_PLAN_AHEAD=$mode python scripts/e2e_cpp.py 2000 300000 200 2 > gpurun_out/e2e_plan_ahead_$mode.txt 2>&1
done
grep -E "^run|runPgo|solveOptimization|runOptimizationIteration|planned|driver|prepare|obvi_ba_solve|LM step|obvi_ba_set_reproj |obvi_ba_create|get_state|update_state" gpurun_out/e2e_plan_ahead_0.txt gpurun_out/e2e_plan_ahead_1.txt | cut -c1-330
