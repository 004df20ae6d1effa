#!/usr/bin/env python3
"""Dev tool: global loads that are waited for on their own (s_waitcnt vmcnt(0) right behind a single load, no other load near it) in the
ISA of every kernel of a file -- each is a full memory latency on the wavefront's path.  usage: lonely_loads.py file.s [kernel-substring]
(hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only x.hip -o file.s)"""
import re, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else None
for f in re.split(r'\n(?=_Z[^\n]*:\s*; @)', txt):
    name = f.split(':')[0]
    m = re.search(r'(k_[a-z_0-9]+?)(?:I|E)', name)
    if not m:
        continue
    lines = f.split('\n')
    loads = [i for i, l in enumerate(lines) if re.search(r'\b(global|buffer)_load', l)]
    lonely = []
    for idx, i in enumerate(loads):
        nxt = loads[idx + 1] if idx + 1 < len(loads) else 10 ** 9
        prv = loads[idx - 1] if idx > 0 else -10 ** 9
        w = [j for j in range(i + 1, min(i + 8, len(lines))) if 's_waitcnt vmcnt(0)' in lines[j]]
        if w and w[0] < nxt and i - prv > 6:
            lonely.append(i)
    print("%-26s loads %4d  waited for on their own %3d" % (m.group(1), len(loads), len(lonely)))
    if want and want in name:
        for i in lonely:
            print("   ", i, lines[i].strip()[:60], "|", " ; ".join(l.strip()[:36] for l in lines[i + 1:i + 4]))
