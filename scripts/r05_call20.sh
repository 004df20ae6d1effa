#!/bin/bash
mkdir -p gpurun_out
{
for q in 4 8 16 32; do
  echo "== GPU_MAX_HW_QUEUES=$q, K threads in one process, serial sessions"
  GPU_MAX_HW_QUEUES=$q OBVI_SESSIONS_IN_PROCESS=1 OBVI_HOST_PLAN_AHEAD=0 python scripts/concurrent_sessions.py 300 30000 20 1,4,8
done
echo "== GPU_MAX_HW_QUEUES=16, planned ahead"
GPU_MAX_HW_QUEUES=16 OBVI_SESSIONS_IN_PROCESS=1 python scripts/concurrent_sessions.py 300 30000 20 1,4,8
} 2>&1 | tee gpurun_out/concurrent_sessions_hw_queues.txt
