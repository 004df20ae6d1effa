#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_end_state.py -q -m gpu -s -k config3w > gpurun_out/r06/config3w_test.txt 2>&1
grep -v "^E   \|^    \|^>" gpurun_out/r06/config3w_test.txt | cut -c1-400 | tail -40
