#!/usr/bin/env python3
"""GPU idle time between LM steps of the timed (un-instrumented) solve of bench.py, from a rocprofv3 kernel trace:
per step, the gap between the end of the trial-cost kernel (+ the scalar publish) and the start of the next point pass, and what runs in it.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python bench.py --no-cpu-baseline; step_gaps.py DIR/**/run_kernel_trace.csv"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); return m.group(1) if m else r["Kernel_Name"][:24]
pp = [i for i, r in enumerate(rows) if name(r) == "k_point_pass"]
gaps, spans = [], []
for a, b in zip(pp[5:30], pp[6:31]):   # steps of the timed solve (after the warm-up)
    seg = rows[a:b]
    cost = [r for r in seg if name(r) == "k_cost"]
    if not cost:
        continue
    t_cost_end = int(cost[-1]["End_Timestamp"])
    t_next = int(rows[b]["Start_Timestamp"])
    busy = sum(min(int(r["End_Timestamp"]), t_next) - max(int(r["Start_Timestamp"]), t_cost_end) for r in seg if int(r["End_Timestamp"]) > t_cost_end)
    gaps.append((t_next - t_cost_end) / 1e3); spans.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    inside = [name(r) for r in seg if int(r["Start_Timestamp"]) >= t_cost_end]
print("steps %d: span %.1f us (median), cost-end -> next point pass %.1f us (median), min %.1f max %.1f; kernels in the gap: %s" % (len(gaps), sorted(spans)[len(spans) // 2], sorted(gaps)[len(gaps) // 2], min(gaps), max(gaps), inside))
