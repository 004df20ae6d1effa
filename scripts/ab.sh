#!/bin/bash
# usage: scripts/ab.sh VAR  -- alternates runs with and without VAR=1, prints ms/step
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $1=1; else unset $1; fi
    timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1=$v ms/step %.4f' % d['ms_per_step'])"
  done
done
