#!/usr/bin/env python
"""Print the launch sequence of one LM iteration from a rocprofv3 kernel-trace CSV: start offset, duration, gap to the previous
launch on the same queue.  usage: level_timeline.py <kernel_trace.csv> [iteration_index_from_end]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("obvi::", "")).replace("void ", ""), r.get("Queue_Id", "")) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_zero_tiles" in e[2]]
a, b = starts[-back - 1], starts[-back]
t0 = ev[a][0]
last_end = {}
for s, e, n, q in ev[a:b]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print("%9.1f us  dur %7.2f  gap %6.2f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n[:50]))
    last_end[q] = e
print("iteration span %.1f us" % ((ev[b][0] - t0) / 1e3))
