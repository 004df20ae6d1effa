#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05m
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
timeout 900 python -m pytest tests/test_gpu_structure.py -q -m gpu -x > $O/t_structure.log 2>&1; echo "structure rc=$?"; tail -3 $O/t_structure.log | head -2
for H in 0 1; do
  echo "== OBVI_PLAN_SLOTS_ON_HOST=$H" >> $O/prepare.txt
  OBVI_PLAN_SLOTS_ON_HOST=$H OBVI_DEBUG_PREPARE=1 timeout 300 python scripts/prepare_time.py 2000 300000 200 >> $O/prepare.txt 2>&1
  echo "== window OBVI_PLAN_SLOTS_ON_HOST=$H" >> $O/prepare.txt
  OBVI_PLAN_SLOTS_ON_HOST=$H OBVI_DEBUG_PREPARE=1 timeout 300 python scripts/window_iter.py 2>&1 | grep "prepare:" | tail -9 >> $O/prepare.txt
done
grep -E "^==|schur batches|upload \+ alloc|symbolic phase about" $O/prepare.txt | head -60
for H in 0 1; do for i in 1 2; do OBVI_PLAN_SLOTS_ON_HOST=$H OBVI_API_TIMING=1 timeout 600 python scripts/session_time.py > $O/session_h${H}_$i.txt 2>&1; echo "host=$H: $(tail -1 $O/session_h${H}_$i.txt | cut -c1-160)"; grep "prepare (symbolic" $O/session_h${H}_$i.txt; done; done
