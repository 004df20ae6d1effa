#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
for i in 1 2 3; do OBVI_HOST_TIMING=1 OBVI_API_TIMING=1 timeout 600 python scripts/session_time.py > $O/session_$i.txt 2>&1; tail -1 $O/session_$i.txt; grep "prepare (symbolic\|set_reproj: gather\|LM step" $O/session_$i.txt; done
OBVI_HOST_TIMING=1 timeout 600 python scripts/e2e_cpp.py 2000 300000 200 3 > $O/e2e_cpp.txt 2>&1; head -3 $O/e2e_cpp.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"; tail -4 $O/t_all.log | head -2
bash scripts/profile_round.sh r05a > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -2 $O/profile_round.log | cut -c1-200
