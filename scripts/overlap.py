#!/usr/bin/env python3
"""How much do the kernels of different HIP streams overlap on the device?  Reads a rocprofv3 --kernel-trace CSV: per queue the busy time,
the wall span, the time during which >= 2 / >= 3 queues have a kernel running, and the launch sequence of a slice of the trace.
usage: overlap.py <kernel_trace.csv> [tail_ms=3]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("obvi::", "")).replace("void ", "")) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = {}
for s, e, q, n in ev:
    busy[q] = busy.get(q, 0) + (e - s)
print("trace: %d dispatches on %d queues, span %.2f ms" % (len(ev), len(busy), (t1 - t0) / 1e6))
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("  queue %-4s busy %9.3f ms (%4.1f %% of the span), %6d dispatches" % (q, b / 1e6, 100.0 * b / (t1 - t0), sum(1 for x in ev if x[2] == q)))
# sweep: time with k queues active (a queue counts once however many of its kernels overlap)
pts = []
for s, e, q, n in ev:
    pts.append((s, 1, q)); pts.append((e, -1, q))
pts.sort()
active, level_time, last = {}, {}, pts[0][0]
for t, d, q in pts:
    k = sum(1 for v in active.values() if v > 0)
    level_time[k] = level_time.get(k, 0) + (t - last)
    last = t
    active[q] = active.get(q, 0) + d
tot = sum(level_time.values())
print("queues with a kernel running:  " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(level_time.items())))
print("sum of kernel durations / span = %.2f" % (sum(busy.values()) / (t1 - t0)))
cut = t1 - int(tail_ms * 1e6)
print("last %.1f ms:" % tail_ms)
for s, e, q, n in ev:
    if s >= cut:
        print("%10.1f us  dur %7.2f  q%-3s %s" % ((s - cut) / 1e3, (e - s) / 1e3, q, n[:40]))
