#!/bin/bash
mkdir -p gpurun_out/r06
{
echo "# round 6: the 300-frame two-phase sliding-window session through the C++ host mirror (scripts/session_time.py), three runs, next window planned beside the solve (default)"
for i in 1 2 3; do python scripts/session_time.py 2>&1 | tail -1; done
echo "# K sessions at once as K host threads of one process, serial sessions (scripts/concurrent_sessions.py)"
OBVI_SESSIONS_IN_PROCESS=1 OBVI_HOST_PLAN_AHEAD=0 python scripts/concurrent_sessions.py 300 30000 20 1,4,8 2>&1 | tail -4
} 2>&1 | tee gpurun_out/r06/session_300_frames.txt
