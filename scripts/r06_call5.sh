#!/bin/bash
mkdir -p gpurun_out/r06
{
echo "== feature ids random (generator) vs in order of first sighting (OBVI_SYNTH_SORT_POINTS=1), config 3"
bash scripts/ab_env.sh "X=0" "OBVI_SYNTH_SORT_POINTS=1"
timeout 900 python -m pytest tests/test_gpu_nine_dof.py tests/test_gpu_shared_objects.py -q -m gpu -k "nine or eight_ranks" 2>&1 | tail -4
} 2>&1 | tee gpurun_out/r06/point_order.txt
