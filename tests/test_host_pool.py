"""CPU: the library's host thread pool (obvi-slam_amd/csrc/host_util.h: HostPool) under concurrent callers.  Round 5 lets calls from different
threads run side by side -- one handle per thread plans at the same time (config #5's sessions) -- instead of queueing them behind one mutex; the
contract stays: run(parts, fn) calls fn(0) ... fn(parts - 1) exactly once each and returns when all are done, whoever else is calling.
tests/hostpool_shim.cpp drives it (also clean under ThreadSanitizer: hipcc -fsanitize=thread on the shim)."""
import ctypes
import os
import subprocess

import pytest

import helpers

LIB = os.path.join(helpers.ROOT, "tests", "libhostpool.so")


@pytest.fixture(scope="module")
def pool():
    src = os.path.join(helpers.ROOT, "tests", "hostpool_shim.cpp")
    hdr = os.path.join(helpers.ROOT, "obvi-slam_amd", "csrc", "host_util.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-w", "-o", LIB, src])
    return ctypes.CDLL(LIB)


@pytest.mark.parametrize("workers,callers,runs,max_parts", [(7, 1, 3000, 16), (7, 4, 3000, 16), (3, 8, 1500, 40), (15, 16, 500, 16), (0, 3, 200, 8), (1, 2, 2000, 3)])
def test_every_part_runs_exactly_once_whoever_else_is_calling(pool, workers, callers, runs, max_parts):
    assert pool.hostpool_stress(workers, callers, runs, max_parts) == 0
