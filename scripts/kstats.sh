#!/bin/bash
# usage (GPU box, repo root): scripts/kstats.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$TAG -o run -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_rocprof.err
F=$(find /tmp/kst_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    import re; m = re.search(r"(k_\w+)", r["Name"]); print("%-60s calls %6s avg %10.2f us total %10.3f ms  %5s%%" % ((m.group(1) if m else r["Name"][:60]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
