#!/bin/bash
# usage (GPU box, repo root): scripts/window_trace.sh <tag> [P L O]  -> gpurun_out/<tag>_window_timeline.txt: the launches of one LM iteration of a
# local-BA window (rocprofv3 --kernel-trace of scripts/window_iter.py; start offset, duration, gap on the same queue)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wtr_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/wtr_$TAG -o run -- python $R/scripts/window_iter.py "$@" > $R/gpurun_out/${TAG}_window_iter.txt 2> $R/gpurun_out/${TAG}_window_rocprof.err
F=$(find /tmp/wtr_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/level_timeline.py "$F" 60 > $R/gpurun_out/${TAG}_window_timeline.txt
cat $R/gpurun_out/${TAG}_window_iter.txt
