#!/bin/bash
# round 6, call 1: matrix-free Schur prototype -- parity of the MF strip kernel, then same-box A/B of the three modes
mkdir -p gpurun_out/r06
{
echo "== parity of the matrix-free strip kernel (OBVI_SCHUR_MF=1: Z still stored, the strips form their own)"
OBVI_SCHUR_MF=1 timeout 1200 python -m pytest tests/test_gpu_structure.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
echo "== OBVI_SCHUR_MF=2 (nothing stored; far pairs / covariances not covered by the prototype): structure tests, failures expected only where far pairs exist"
OBVI_SCHUR_MF=2 timeout 900 python -m pytest tests/test_gpu_structure.py -q -m gpu 2>&1 | tail -8
echo "== A/B, config 3"
for i in 1 2 3; do
  for setting in "X=0" "OBVI_SCHUR_MF=1" "OBVI_SCHUR_MF=2"; do
    env $setting timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; p=d['phases_ms_avg']
print('[%s] ms/step %.4f final_cost %.6e | in situ: ' % ('$setting', d['ms_per_step'], d['config']['final_cost']) + ' '.join('%s %.0f' % (n[:12], 1e3*v) for n, v in p.items()) + ' | alone: ' + ' '.join('%s %.0f' % (n, k[n]['avg_us']) for n in ('point_pass','schur_window','point_backsub','cost')))"
  done
done
} 2>&1 | tee gpurun_out/r06/mf_prototype.txt
