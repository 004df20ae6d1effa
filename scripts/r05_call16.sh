#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_host_mirror.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
{
echo "== K threads in one process, windows planned ahead"; OBVI_SESSIONS_IN_PROCESS=1 python scripts/concurrent_sessions.py 300 30000 20 1,2,4,8
echo "== K threads in one process, serial sessions"; OBVI_SESSIONS_IN_PROCESS=1 OBVI_HOST_PLAN_AHEAD=0 python scripts/concurrent_sessions.py 300 30000 20 1,2,4,8
echo "== K threads in one process, planned ahead, no spinning beside thread"; OBVI_SESSIONS_IN_PROCESS=1 OBVI_HOST_BESIDE_SPIN_US=0 python scripts/concurrent_sessions.py 300 30000 20 2,4,8
} 2>&1 | tee gpurun_out/concurrent_sessions_in_process.txt
