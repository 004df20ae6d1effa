#!/bin/bash
# usage: scripts/ab_env.sh "A=1 B=2" "A=3" ...   -- alternates bench runs under each environment setting (three rounds), prints ms/step and the
# main kernels' in-situ times.  "" = the defaults.
for i in 1 2 3; do
  for setting in "$@"; do
    env $setting timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('[%s] ms/step %.4f  schur %.1f (alone %.1f)  point_pass %.1f  chol %.1f' % ('$setting', d['ms_per_step'], k['schur_window'].get('in_situ_us', 0), k['schur_window']['avg_us'], k['point_pass'].get('in_situ_us', 0), 1e3*d['phases_ms_avg'].get('cholesky_solve', 0)))"
  done
done
