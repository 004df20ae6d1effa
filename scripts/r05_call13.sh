#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4; done
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_lockstep_session.py tests/test_host_mirror.py -q -m gpu -x 2>&1 | tail -2; done
