#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python - <<'PY' > /tmp/scene.txt
import os, sys
sys.path[:0] = ["obvi-slam_amd/python"]
import synth, scene_io
prob = synth.make_problem(P=300, L=30000, O=20, seed=4, min_obj_obs=10, bbox_noise=5.0, object_classes=("bench",))
scene_io.write_scene_binary(prob, "/tmp/scene.bin")
PY
for k in 1 4 8; do
  echo "== K=$k threads in one process, serial sessions"
  OBVI_HOST_PLAN_AHEAD=0 OBVI_API_TIMING=1 obvi-slam_amd/host/run_offline_ba /tmp/scene.bin /tmp/out.json --window 50 --gba-frequency 100 $( [ $k -gt 1 ] && echo --sessions-in-process $k ) 2>&1 | grep -E "LM step|sessions_in_process|obvi_ba_solve |prepare \(sym|set_reproj   "
done 2>&1 | tee gpurun_out/concurrent_submit_time.txt
