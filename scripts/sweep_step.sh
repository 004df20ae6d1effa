#!/bin/bash
# usage (GPU box): scripts/sweep_step.sh VAR "v1 v2 ..."  -> ms per LM iteration, Schur strip kernel alone / in situ, reduced solve per value (config #3)
VAR=$1; VALS=$2
for v in $VALS; do
  export $VAR=$v
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_avg']; k=d['kernels']
print(sys.argv[1], 'ms/step %.4f' % d['ms_per_step'], 'schur %.1f / %.1f us' % (k['schur_window']['avg_us'], k['schur_window'].get('in_situ_us') or 0), 'chol %.4f' % p['cholesky_solve'])" "$VAR=$v"
done
