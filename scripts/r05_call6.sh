#!/bin/bash
# the round's full check: every -m gpu test, the profile set of scripts/profile_round.sh (copy gpurun_out/r05a/{manifest.json,r05a_*} into profiles/ afterwards), smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"; tail -6 $O/t_all.log
bash scripts/profile_round.sh r05a > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -15 $O/profile_round.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
