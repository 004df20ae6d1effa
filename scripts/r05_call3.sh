#!/bin/bash
# round-5 measurement batch (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd /tmp && export TMPDIR=/tmp
# A. do the kernels of two / four concurrent sessions overlap on the device?
for S in 2 4; do
  rm -rf /tmp/ktr_s$S
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr_s$S -o run -- python $R/bench.py --config 5 --sessions $S --steps 4 --warmup 1 --no-cpu-baseline > $O/ktr_s${S}_bench.json 2> $O/ktr_s${S}.err
  F=$(find /tmp/ktr_s$S -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/overlap.py "$F" 2.5 > $O/overlap_s$S.txt 2>&1
done
rm -rf /tmp/ktr_q
GPU_MAX_HW_QUEUES=16 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr_q -o run -- python $R/bench.py --config 5 --sessions 4 --steps 4 --warmup 1 --no-cpu-baseline > $O/ktr_s4q16_bench.json 2> $O/ktr_s4q16.err
python3 $R/scripts/overlap.py "$(find /tmp/ktr_q -name '*kernel_trace.csv' | head -1)" 2.5 > $O/overlap_s4q16.txt 2>&1
cd $R
# config 4 windows per GPU (25 shared objects: a 3-column tail)
for K in 1 2 4; do
  timeout 600 python bench.py --config 4 --windows-per-gpu $K --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg4_k$K.json 2> $O/bench_cfg4_k$K.err
done
# B. end state with the arbiter
timeout 900 python scripts/end_state.py 2 20 100 1 > $O/end_state_2.txt 2> $O/end_state_2.err
timeout 1200 python scripts/end_state.py 2o 20 100 1 > $O/end_state_2o.txt 2> $O/end_state_2o.err
# C. where the symbolic phase goes
for T in 1 4 16 32; do
  echo "== OBVI_HOST_THREADS=$T" >> $O/prepare_stages.txt
  OBVI_HOST_THREADS=$T OBVI_DEBUG_PREPARE=1 timeout 300 python scripts/prepare_time.py 2000 300000 200 >> $O/prepare_stages.txt 2>&1
done
echo "== OBVI_HOST_THREADS=16 OBVI_HOST_AFFINITY=1" >> $O/prepare_stages.txt
OBVI_HOST_THREADS=16 OBVI_HOST_AFFINITY=1 OBVI_DEBUG_PREPARE=1 timeout 300 python scripts/prepare_time.py 2000 300000 200 >> $O/prepare_stages.txt 2>&1
nproc >> $O/prepare_stages.txt; lscpu | grep -E "Model name|Socket|NUMA|Thread" >> $O/prepare_stages.txt
# D. side stream on a CU subset
for C in 0 64 128 160 192 224; do
  OBVI_SIDE_CUS=$C timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deterministic-leg --no-end-to-end > $O/side_cus_$C.json 2> $O/side_cus_$C.err
done
# E. the whole GPU suite
timeout 1200 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"
tail -15 $O/t_all.log
