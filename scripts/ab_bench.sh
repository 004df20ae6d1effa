#!/bin/bash
# usage (GPU box): scripts/ab_bench.sh [label]  -> one line: ms/step and the Schur / Cholesky phase times of the default bench
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; p=d['phases_ms_avg']
print('$1', 'ms/step', round(d['ms_per_step'],4), 'schur', k['schur_window']['avg_us'], k['schur_window'].get('in_situ_us'), 'chol', p['cholesky_solve'], 'point', p['point_pass'], 'backsub', p['point_backsub'], 'cost', p['cost'], 'update_potrf', k['k_update_potrf']['avg_us'], 'trsm', k['k_trsm']['avg_us'])"
