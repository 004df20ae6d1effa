#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_levels.sh <tag>  -> gpurun_out/<tag>_levels.txt: FETCH_SIZE / WRITE_SIZE of every k_update_potrf and k_trsm launch of one LM step
# (two counter-only rocprofv3 passes; raw counter values in KiB, see scripts/pmc_summary.py for the calibration)
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pl_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pl_$C -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-deterministic-leg --no-end-to-end > /dev/null 2> $R/gpurun_out/${TAG}_pmc_$C.err
done
python3 - "$(find /tmp/pl_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find /tmp/pl_WRITE_SIZE -name '*counter_collection.csv' | head -1)" > $R/gpurun_out/${TAG}_levels.txt <<'PY'
import csv, re, sys
def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(re.search(r"(k_\w+)", r["Kernel_Name"]).group(1) if re.search(r"(k_\w+)", r["Kernel_Name"]) else r["Kernel_Name"][:20], float(r["Counter_Value"])) for r in rows]
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
# the last complete step: between the last two k_point_pass dispatches
pp = [i for i, (n, _) in enumerate(f) if n == "k_point_pass"]
a, b = pp[-2], pp[-1]
lvl = 0
for (n, fv), (n2, wv) in zip(f[a:b], w[a:b]):
    if n in ("k_update_potrf", "k_trsm", "k_potrf", "k_schur_window", "k_backward"):
        print("%-16s fetch raw %9.2f MB   write %9.2f MB" % (n, fv * 1024 / 1e6, wv * 1024 / 1e6))
PY
cat $R/gpurun_out/${TAG}_levels.txt | head -70
