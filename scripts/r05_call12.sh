#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host_mirror.py -x -q -m gpu 2>&1 | tail -4
OBVI_HOST_TIMING=1 OBVI_API_TIMING=1 python scripts/session_time.py > gpurun_out/session_planned_ahead_api.txt 2>&1
tail -3 gpurun_out/session_planned_ahead_api.txt | cut -c1-250
python scripts/e2e_cpp.py 2000 300000 200 3 > gpurun_out/e2e_planned_ahead.txt 2>&1
grep "^run" gpurun_out/e2e_planned_ahead.txt
