#!/bin/bash
# usage (GPU box, repo root): scripts/ktrace.sh <tag> [bench args]  -> gpurun_out/<tag>_kernel_trace.csv (rocprofv3 --kernel-trace: one row per dispatch with start / end)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktr_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr_$TAG -o run -- python $R/bench.py --no-cpu-baseline --no-deterministic-leg --no-end-to-end "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_rocprof.err
F=$(find /tmp/ktr_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$F" $R/gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); return m.group(1) if m else r["Kernel_Name"][:30]
# the last complete LM iteration: from the last k_point_pass backwards to the previous one
pp = [i for i, r in enumerate(rows) if name(r) == "k_point_pass"]
a, b = pp[-3], pp[-2]
t0 = int(rows[a]["Start_Timestamp"])
out = open(sys.argv[2], "w")
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write("%-22s start %9.2f us  dur %8.2f us  gap %7.2f us  grid %s\n" % (name(r), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "")))
    prev_end = max(prev_end, e)
out.write("iteration: %.2f us\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
head -5 $R/gpurun_out/${TAG}_timeline.txt
