#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes).  Units: the counters are in KiB.  gfx950 correction from the guide: FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced streaming read, so the read side is reported both raw and doubled; WRITE_SIZE
is uncalibrated on gfx950 and reported raw.  usage: pmc_summary.py fetch.csv write.csv out.json"""
import collections, csv, json, re, sys
def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:40]
        d[name][0] += 1; d[name][1] += float(r["Counter_Value"])
    return d
f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 3 --warmup 1`; KiB -> bytes; "
                  "hbm_bytes_per_launch = 2*FETCH (gfx950 half-count correction for wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE (raw, uncalibrated)",
       "kernels": {}}
print("%-24s %6s %16s %16s %18s" % ("kernel", "calls", "fetch MB/launch", "write MB/launch", "2*fetch+write MB"))
for k in sorted(f, key=lambda k: -f[k][1]):
    fb = f[k][1] / f[k][0] * 1024.0
    wb = (w[k][1] / w[k][0] * 1024.0) if k in w and w[k][0] else 0.0
    out["kernels"][k] = {"calls": f[k][0], "fetch_bytes_raw": fb, "write_bytes_raw": wb, "hbm_bytes_per_launch": 2 * fb + wb}
    print("%-24s %6d %16.3f %16.3f %18.3f" % (k, f[k][0], fb / 1e6, wb / 1e6, (2 * fb + wb) / 1e6))
json.dump(out, open(sys.argv[3], "w"), indent=1)
