#!/bin/bash
# round-5 measurement batch 2 (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
export PYTHONPATH=$R/obvi-slam_amd/python:$R/tests
cd $R
# converged config #3 on the host cores, beside everything else (the oracle's two-phase global BA takes minutes)
(timeout 2400 python scripts/end_state_table.py 3 64 0 0 > $O/end_state_3.txt 2> $O/end_state_3.err) &
BG=$!
timeout 900 python -m pytest tests/test_gpu_end_state.py tests/test_gpu_session_groups.py -q -m gpu > $O/t_new.log 2>&1; echo "new tests rc=$?"; tail -4 $O/t_new.log
for S in 16 4 2; do
  timeout 600 python bench.py --config 5 --sessions $S --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_fused_s$S.json 2> $O/cfg5_fused_s$S.err; echo "cfg5 fused $S rc=$?"
done
timeout 600 python bench.py --config 5 --sessions 16 --group --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_group_s16.json 2> $O/cfg5_group_s16.err; echo "cfg5 group 16 rc=$?"
for K in 1 2 4 8; do
  timeout 600 python bench.py --config 4 --windows-per-gpu $K --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg4_fused_k$K.json 2> $O/cfg4_fused_k$K.err; echo "cfg4 fused $K rc=$?"
done
timeout 600 python bench.py --config 4 --windows-per-gpu 4 --group --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg4_group_k4.json 2> $O/cfg4_group_k4.err
timeout 600 python bench.py --config 5 --chain --sessions 4 > $O/cfg5_chain_s4.json 2> $O/cfg5_chain_s4.err; echo "chain rc=$?"
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_end_state.py --deselect tests/test_gpu_session_groups.py > $O/t_all.log 2>&1; echo "all rc=$?"; tail -4 $O/t_all.log
wait $BG
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"
tail -c 400 $O/end_state_3.txt
