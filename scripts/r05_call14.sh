#!/bin/bash
mkdir -p gpurun_out/lock
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python -m pytest "tests/test_lockstep_session.py::test_every_optimisation_of_a_session_follows_the_oracle[False]" -q -m gpu 2>&1 | grep -E "^E  |passed|failed" | cut -c1-1500 | head -6
  cp gpurun_out/lockstep_default.jsonl gpurun_out/lock/run_$i.jsonl
done
