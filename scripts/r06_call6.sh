#!/bin/bash
mkdir -p gpurun_out/r06
{
echo "== the internal feature numbering: its own tests"
timeout 900 python -m pytest tests/test_gpu_point_order.py -q -m gpu -x 2>&1 | tail -15
echo "== the parity / structure / deterministic / shared-object tests with EVERY problem renumbered (OBVI_POINT_RENUMBER_MIN=1)"
OBVI_POINT_RENUMBER_MIN=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_structure.py tests/test_gpu_deterministic.py tests/test_gpu_shared_objects.py tests/test_gpu_nine_dof.py tests/test_gpu_datasets.py -q -m gpu 2>&1 | tail -12
echo "== A/B config 3: caller's numbering (OBVI_POINT_RENUMBER_MIN=0) vs internal numbering (default)"
bash scripts/ab_env.sh "OBVI_POINT_RENUMBER_MIN=0" "X=0"
} 2>&1 | tee gpurun_out/r06/point_renumber.txt
