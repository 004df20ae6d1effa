#!/bin/bash
mkdir -p gpurun_out
{
for t in 16 4 2 1; do
  echo "== in one process, serial sessions, OBVI_HOST_THREADS=$t"
  OBVI_HOST_THREADS=$t OBVI_SESSIONS_IN_PROCESS=1 OBVI_HOST_PLAN_AHEAD=0 python scripts/concurrent_sessions.py 300 30000 20 4,8,12
done
echo "== in one process, planned ahead, OBVI_HOST_THREADS=2"
OBVI_HOST_THREADS=2 OBVI_SESSIONS_IN_PROCESS=1 python scripts/concurrent_sessions.py 300 30000 20 4,8
} 2>&1 | tee gpurun_out/concurrent_sessions_threads.txt
