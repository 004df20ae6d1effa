#!/bin/bash
# round 6, call 3: the oracle's end state on config 3w (fixture), both HIP modes against it, the new shared-object / bench-contract tests
mkdir -p gpurun_out/r06
{
nproc
mkdir -p gpurun_out/r06; ( time python tests/golden/gen_config3w_end_state.py 20 ) 2>&1 | tail -8
cp tests/golden/config3w_end_state.npz gpurun_out/r06/
timeout 1200 python -m pytest tests/test_gpu_end_state.py -q -m gpu -s -k config3w > gpurun_out/r06/config3w_test.txt 2>&1; grep -v "^E   \|^    \|^>" gpurun_out/r06/config3w_test.txt | cut -c1-400 | tail -30
timeout 900 python -m pytest tests/test_gpu_shared_objects.py -q -m gpu -k "planned_ahead or refused" 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r06/config3w.txt
