#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" OBVI_HOST_TIMING=1 OBVI_API_TIMING=1 python scripts/session_time.py 2>&1 | grep -E "wall|planned ahead|obvi_ba_set_reproj|sort by point|by pose  |gather|prepare \(symbolic"; }
for rep in 1 2; do
run "default" X=1
run "arena max 1" MALLOC_ARENA_MAX=1
run "no trim, no mmap" MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=67108864
run "arena 1 + no trim" MALLOC_ARENA_MAX=1 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=67108864
done 2>&1 | tee gpurun_out/plan_ahead_malloc.txt
