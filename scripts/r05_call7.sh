#!/bin/bash
# stage times of the symbolic phase over the windows of the 300-frame session
mkdir -p gpurun_out
OBVI_DEBUG_PREPARE=1 python scripts/session_time.py 2> gpurun_out/prepare_stages.txt | tail -2
python - <<'PY'
import re, collections
tot = collections.defaultdict(float); n = collections.Counter()
for ln in open("gpurun_out/prepare_stages.txt"):
    m = re.match(r"prepare: (.*?)\s+([0-9.]+) ms", ln)
    if m: tot[m.group(1).strip()] += float(m.group(2)); n[m.group(1).strip()] += 1
for k in tot: print("%-28s %8.1f ms over %4d calls = %.4f ms each" % (k, tot[k], n[k], tot[k] / n[k]))
print("sum %.1f ms" % sum(tot.values()))
PY
