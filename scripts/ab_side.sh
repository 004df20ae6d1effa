python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; p=d['phases_ms_avg']
print('$1', 'ms/step', round(d['ms_per_step'],4), 'schur', k['schur_window']['avg_us'], k['schur_window'].get('in_situ_us'), 'pose_pass', k.get('pose_pass',{}).get('avg_us'), k.get('pose_pass',{}).get('in_situ_us'), 'small', k.get('small_factors',{}).get('avg_us'))"
