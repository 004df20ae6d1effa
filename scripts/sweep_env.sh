#!/bin/bash
# usage: scripts/sweep_env.sh VAR "v1 v2 ..." [kernel names...]   -- runs bench.py per value, prints ms/step and the chosen kernel rows
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  export $VAR=$v
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
names=sys.argv[2:] or list(k)
print(sys.argv[1], 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s=%.1fus' % (n, k[n]['avg_us']) for n in names if n in k))
" "$VAR=$v" "$@"
done
