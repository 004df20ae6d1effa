#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host_mirror.py tests/test_reference_files.py -q -m gpu 2>&1 | grep -E "passed|failed|^E " | head
for rep in 1 2 3; do for mode in 0 1; do
  OBVI_HOST_PLAN_AHEAD=$mode python scripts/e2e_cpp.py 2000 300000 200 1 > gpurun_out/e2e_vf_${mode}_${rep}.txt 2>&1
  python - "$mode" gpurun_out/e2e_vf_${mode}_${rep}.txt <<'PY'
import re,sys
t=open(sys.argv[2]).read()
run=re.search(r"runFullOptimization (\d+) ms", t); lm=re.search(r"LM step \(submit \+ wait\)\s+([0-9.]+) ms in\s+(\d+)", t)
print("plan ahead %s: runFullOptimization %s ms, LM steps %s ms in %s, outside %.0f ms" % (sys.argv[1], run.group(1), lm.group(1), lm.group(2), float(run.group(1)) - float(lm.group(1))))
PY
done; done 2>&1 | tee gpurun_out/e2e_vf_beside.txt
grep -E "runPgo|planned|beside" gpurun_out/e2e_vf_1_3.txt | cut -c1-200
