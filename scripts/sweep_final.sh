#!/bin/bash
# usage (GPU box, repo root): bash scripts/sweep_final.sh  -- one bench line per value of the plan / schedule knobs (ms per LM step, reduced solve, strip kernel)
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-deterministic-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_avg']
print(sys.argv[1], 'ms/step %.4f' % d['ms_per_step'], 'chol', p['cholesky_solve'], 'schur', p['schur_window'])" "$1"; }
run base
for v in 48 80 96; do OBVI_ND_LEAF=$v run ND_LEAF=$v; done
for v in 2 8; do OBVI_ND_G=$v run ND_G=$v; done
for v in 1024 2048 3072; do OBVI_SCHUR_WGS=$v run SCHUR_WGS=$v; done
for v in 1 3; do OBVI_UPD_CHUNK=$v run UPD_CHUNK=$v; done
for v in 256 768; do OBVI_SLICE_MAX=$v run SLICE_MAX=$v; done
for v in 3 5 6; do OBVI_BACKWARD_LEVELS=$v run BW_LEVELS=$v; done
run base
