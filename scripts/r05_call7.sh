#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
OBVI_HOST_TIMING=2 timeout 600 python scripts/e2e_cpp.py 2000 300000 200 2 > $O/e2e_cpp.txt 2>&1
OBVI_DEBUG_PREPARE=1 timeout 300 python scripts/window_iter.py > $O/window_plan.txt 2>&1
OBVI_HOST_TIMING=1 OBVI_API_TIMING=1 timeout 600 python scripts/session_time.py > $O/session.txt 2>&1
tail -60 $O/e2e_cpp.txt
