#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
# Writes under gpurun_out/<tag>/: the bench JSON line, the rocprofv3 kernel stats of the same command, and HBM read / write
# bytes per kernel from two separate counter-only passes (FETCH_SIZE, WRITE_SIZE).  Copy what should be kept to profiles/.
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cfg3 -- python $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o cfg3 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err
done
ls $OUT $OUT/trace | head -30
grep -h metric $OUT/bench.json | cut -c1-400
