#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes).  Counter unit: KiB.
gfx950 correction: the guide says FETCH_SIZE reports half the bytes of a wide coalesced streaming read.  Two kernels of this
library read a known number of bytes and pin the factor for the two access patterns that occur here:
  * k_copy3 streams 3 arrays with full-line coalesced loads: 7.31 MB read (same as it writes; WRITE_SIZE reports 7.31 MB) and
    raw FETCH_SIZE reports 3.67 MB                      -> factor 2 for full-line streaming reads
  * k_point_backsub (today the point part of k_backsub_apply) reads every Z record once (2 990 848 x 144 B = 430.7 MB) through per-lane 144-byte-strided loads; with
    8-byte loads raw FETCH_SIZE reported 451.8 MB          -> factor 1 for gathers / partial lines
    (the current kernel uses 16-byte loads and reports 322.9 MB for the same >= 430.7 MB: part of its requests are counted at
    half, so for gather kernels with 16-byte loads the factor-1 figure is a LOWER bound, up to 1.4x low)
The table below assigns a factor per kernel by its dominant read pattern; WRITE_SIZE is used raw (it matches k_copy3).
usage: pmc_summary.py fetch.csv write.csv out.json"""
import collections, csv, json, re, sys
STREAMING = {"k_copy3", "k_trsm", "k_update_potrf", "k_update", "k_potrf", "k_backward", "k_schur_window"}   # tiles / Z staged with full-line loads
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
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE counter-only passes of `bench.py --steps 4 --warmup 1 --no-cpu-baseline` (config #3); "
                  "KiB -> bytes; hbm_bytes_per_launch = fetch_calibration * FETCH + WRITE.  fetch_calibration: 2.0 for kernels that stream full cache lines "
                  "(anchor: k_copy3 reads 7.31 MB, raw FETCH_SIZE 3.67 MB), 1.0 for gathers / partial lines (anchor: k_point_backsub read 430.7 MB of Z "
                  "records with 8-byte loads, raw FETCH_SIZE 451.8 MB; with 16-byte loads it reports 322.9 MB, so factor-1 figures of gather kernels are lower bounds); see scripts/pmc_summary.py",
       "kernels": {}}
print("%-24s %6s %16s %16s %6s %18s" % ("kernel", "calls", "fetch MB/launch", "write MB/launch", "cal", "HBM MB/launch"))
for k in sorted(f, key=lambda k: -f[k][1]):
    fb = f[k][1] / f[k][0] * 1024.0
    wb = (w[k][1] / w[k][0] * 1024.0) if k in w and w[k][0] else 0.0
    cal = 2.0 if k in STREAMING else 1.0
    out["kernels"][k] = {"calls": f[k][0], "fetch_bytes_raw": fb, "write_bytes_raw": wb, "fetch_calibration": cal, "hbm_bytes_per_launch": cal * fb + wb}
    print("%-24s %6d %16.3f %16.3f %6.1f %18.3f" % (k, f[k][0], fb / 1e6, wb / 1e6, cal, (cal * fb + wb) / 1e6))
json.dump(out, open(sys.argv[3], "w"), indent=1)
