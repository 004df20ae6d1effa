#!/bin/bash
# usage: scripts/pmc_kernel.sh <kernel-regex> <out-prefix> "<counters pass 1>" ["<counters pass 2>" ...]
# Collects PMC counters (one rocprofv3 run per pass, counters only -- no tracing) for one kernel of bench.py.
K=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "$K" --output-format csv -d $R/gpurun_out/$OUT -o pass$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-deterministic-leg --no-end-to-end > $R/gpurun_out/$OUT.pass$i.log 2>&1
done
python3 - "$R/gpurun_out/$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print("   %-32s %16.1f per launch (%d launches)" % (c, tot[k][c] / n[k][c], n[k][c]))
PY
