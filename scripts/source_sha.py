#!/usr/bin/env python3
"""sha256 over the kernel / library sources (obvi-slam_amd/csrc): the stamp that ties a profile under profiles/ to the build it
measured.  bench.py refuses to quote profile-derived numbers (roofline.traffic, roofline.rocprof_avg_us) when the stamp in
profiles/manifest.json differs from the sources it runs."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha(root=ROOT):
    d = os.path.join(root, "obvi-slam_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode()); h.update(b"\0"); h.update(open(os.path.join(d, name), "rb").read()); h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_sha())
