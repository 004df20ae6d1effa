#!/bin/bash
mkdir -p gpurun_out
{
echo "== planned ahead, beside thread does not spin"; OBVI_HOST_BESIDE_SPIN_US=0 python scripts/concurrent_sessions.py 300 30000 20 1,2,4,6
echo "== serial sessions"; OBVI_HOST_PLAN_AHEAD=0 python scripts/concurrent_sessions.py 300 30000 20 1,2,4,6
echo "== planned ahead (default)"; python scripts/concurrent_sessions.py 300 30000 20 1,2,4,6
} 2>&1 | tee gpurun_out/concurrent_sessions_modes.txt
