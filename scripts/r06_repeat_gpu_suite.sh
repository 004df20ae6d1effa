#!/bin/bash
# the -m gpu suite several times on one box: which tests are flaky (chaotic sessions, stopping rules decided in the last bits)
mkdir -p gpurun_out/r06
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r06/gpu_repeat_$i.txt 2>&1
  echo "run $i: $(tail -1 gpurun_out/r06/gpu_repeat_$i.txt)"; grep "^FAILED" gpurun_out/r06/gpu_repeat_$i.txt
done
