#!/bin/bash
# usage: scripts/ab_env.sh "A=1 B=2" "A=3" ...   -- alternates bench runs under each environment setting (three rounds), prints ms/step and the
# phase times of the level-1 profiling solve (main-stream phases start-to-next-phase; side-stream phases start-to-end).  "" = the defaults.
for i in 1 2 3; do
  for setting in "$@"; do
    env $setting timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; p=d['phases_ms_avg']
print('[%s] ms/step %.4f | ' % ('$setting', d['ms_per_step']) + ' '.join('%s %.0f' % (n.replace('_window','').replace('cholesky_solve','chol').replace('point_','pt_').replace('small_factors','small').replace('reduced_','')[:10], 1e3*v) for n, v in p.items()) + ' | schur alone %.0f' % k['schur_window']['avg_us'])"
  done
done
