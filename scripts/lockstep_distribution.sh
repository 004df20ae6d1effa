#!/bin/bash
# N lock-step sessions (tests/test_lockstep_session.py, default mode): the distribution behind the test's bars -> gpurun_out/lockstep_runs/run_*.jsonl
N=${1:-40}
mkdir -p gpurun_out/lockstep_runs
for i in $(seq 1 $N); do
  timeout 600 python -m pytest "tests/test_lockstep_session.py::test_every_optimisation_of_a_session_follows_the_oracle[False]" -q -m gpu 2>&1 | grep -E "passed|failed" | head -1
  cp gpurun_out/lockstep_default.jsonl gpurun_out/lockstep_runs/run_$i.jsonl
done
