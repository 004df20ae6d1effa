#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-60s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for name, n, t, mn, mx in rows:
    name = re.sub(r"\(.*", "", name)
    print("%-60s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name[-60:], n, t / 1e3, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
