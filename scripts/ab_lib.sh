#!/bin/bash
# usage (GPU box, repo root): scripts/ab_lib.sh kernel [kernel...]  -- alternates obvi-slam_amd/csrc/libA.bin and libB.bin as libobvi_ba.so (same box, same call:
# boxes differ by +-1.5 %), prints ms/step and the named kernels' average launch time
cd obvi-slam_amd/csrc
for i in 1 2 3; do for v in A B; do
  cp lib$v.bin libobvi_ba.so
  (cd ../..; python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$v ms/step %.3f' % d['ms_per_step'], ' '.join('%s=%.1fus' % (n, k[n]['avg_us']) for n in sys.argv[1:] if n in k))" "$@")
done; done
cp libB.bin libobvi_ba.so
