#!/bin/bash
VAR=$1; VALS=$2
for v in $VALS; do
  export $VAR=$v
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_avg']
print(sys.argv[1], 'ms/step %.4f' % d['ms_per_step'], 'chol %.4f' % p['cholesky_solve'])" "$VAR=$v"
done
