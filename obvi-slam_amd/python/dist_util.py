"""Rank plumbing shared by bench.py and the multi-process tests: one process per GPU.  Independent windows per rank;
either replicas (no data-path collective) or windows that share object blocks, whose per-step all-reduces go
through libobvi_rccl.so (compiled ncclAllReduce forwarder, RcclComm below), through torch.distributed (torch_allreduce),
or -- tests without RCCL -- through a host bounce over any torch.distributed backend (staged_allreduce)."""
import ctypes as C
import os


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def rank_seed(base_seed, config, rank):
    """Each rank solves its own window: distinct deterministic seed per (config, rank)."""
    return base_seed + config + 100 * rank


def reduce_timing(dist, device, seconds, steps_done):
    """(max seconds over ranks, min completed steps over ranks); dist None -> identity."""
    if dist is None:
        return seconds, steps_done
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(steps_done)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.MIN)
    return float(t.item()), int(s.item())


def aggregate_throughput(world, steps_done, seconds):
    """Whole-job LM iterations per second: every rank completed `steps_done` iterations in `seconds`."""
    return world * steps_done / seconds


class _DeviceArray:
    """Zero-copy view of a device buffer handed out by the C ABI (obvi_ba_set_allreduce callback)."""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n),), "typestr": "<f8", "version": 2}


def device_tensor(ptr, n):
    import torch
    return torch.as_tensor(_DeviceArray(ptr, n), device="cuda")


class IssueLog:
    """Issue order of a rank's data-path collectives: (count, op, ordinal of the stream among the streams seen so far) per call, the same
    record libobvi_rccl.so folds into obvi_rccl_sequence.  One communicator is driven from two streams of a handle, which is legal only if
    every rank enqueues the same collectives in the same host order: ranks compare `calls` and `digest()` at a quiescent point."""

    def __init__(self):
        self.calls, self._hash, self._streams, self.records = 0, 1469598103934665603, [], []

    def note(self, count, op, stream):
        if stream not in self._streams:
            self._streams.append(stream)
        rec = (int(count), int(op), self._streams.index(stream))
        for w in rec:
            self._hash = ((self._hash ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        self.calls += 1
        self.records.append(rec)

    def digest(self):
        return self._hash


def same_issue_order(dist, calls, digest):
    """True iff every rank of the torch.distributed group reports the same (calls, digest) -- any backend (values travel as float64 halves)."""
    import torch
    mine = torch.tensor([float(calls), float(digest >> 32), float(digest & 0xFFFFFFFF)], dtype=torch.float64)
    lo, hi = mine.clone(), mine.clone()
    if dist.get_backend() == "nccl":
        lo, hi = lo.cuda(), hi.cuda()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool((lo == hi).all().item())


def torch_allreduce(dist, log=None):
    """obvi_ba all-reduce hook on top of torch.distributed (backend nccl == RCCL over xGMI): the collective is enqueued
    behind the library's own HIP stream, no host synchronisation."""
    import torch
    dev = torch.cuda.current_device()          # the hook may be called from another host thread (a solve on a worker thread): its device is the creator's

    def fn(ptr, count, op, stream):
        if log is not None:
            log.note(count, op, stream)
        with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
            dist.all_reduce(device_tensor(ptr, count), op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
        return 0
    return fn


def staged_allreduce(dist, log=None):
    """The same hook over any torch.distributed backend (gloo in the tests and in `bench.py --oversubscribe`: several ranks on ONE GPU, where
    RCCL refuses to form a communicator): device -> host on the library's stream, all_reduce on the host, host -> device on the same stream."""
    import torch
    dev = torch.cuda.current_device()

    def fn(ptr, count, op, stream):
        if log is not None:
            log.note(count, op, stream)
        with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
            d = device_tensor(ptr, count)
            h = d.cpu()                                     # synchronises the stream: everything before the exchange is done
            dist.all_reduce(h, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
            d.copy_(h)
        return 0
    return fn


def host_allreduce(dist, log=None):
    """The hook for a library whose exchange buffers live in HOST memory -- the CPU oracle, which runs the same exchange protocol on the host
    (oracle_ba_set_allreduce; stream is NULL): torch.distributed.all_reduce in place on the buffer, any backend that takes CPU tensors (gloo)."""
    import numpy as np
    import torch

    def fn(ptr, count, op, stream):
        if log is not None:
            log.note(count, op, stream)
        t = torch.from_numpy(np.ctypeslib.as_array((C.c_double * int(count)).from_address(int(ptr))))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
        return 0
    return fn


def torch_rccl_version():
    """NCCL_VERSION_CODE-style integer of the librccl torch.distributed uses (torch.cuda.nccl.version()), or None."""
    try:
        import torch
        v = torch.cuda.nccl.version()
        if isinstance(v, tuple):
            major, minor, patch = (list(v) + [0, 0])[:3]
            return int(major) * 10000 + int(minor) * 100 + int(patch) if major >= 2 and (major > 2 or minor >= 9) else int(major) * 1000 + int(minor) * 100 + int(patch)
        return int(v)
    except Exception:      # noqa: BLE001 -- a torch without the nccl bindings: nothing to compare with
        return None


class RcclComm:
    """ctypes binding of include/obvi_rccl.h (libobvi_rccl.so): the compiled RCCL all-reduce callback of a C/C++ host."""
    ID_BYTES = 128

    def __init__(self, rank, world, device, unique_id=None, id_file=None, timeout_s=120.0, library=None):
        here = os.path.dirname(os.path.abspath(__file__))
        path = library or os.path.join(os.path.dirname(here), "csrc", "libobvi_rccl.so")
        if not os.path.exists(path):
            raise RuntimeError("%s not found: build it with __graft_entry__.build()" % path)
        self._lib = C.CDLL(path)
        self._c = C.c_void_p()
        if id_file is not None:
            rc = self._lib.obvi_rccl_comm_create_from_file(id_file.encode(), C.c_int32(rank), C.c_int32(world), C.c_int32(device), C.c_double(timeout_s), C.byref(self._c))
        else:
            if unique_id is None or len(unique_id) != self.ID_BYTES:
                raise ValueError("unique_id: %d bytes from RcclComm.unique_id() on rank 0" % self.ID_BYTES)
            rc = self._lib.obvi_rccl_comm_create(C.c_char_p(bytes(unique_id)), C.c_int32(rank), C.c_int32(world), C.c_int32(device), C.byref(self._c))
        if rc != 0:
            raise RuntimeError("obvi_rccl_comm_create failed: status %d" % rc)
        self.rank, self.world_requested = rank, world

    @staticmethod
    def library_path(library=None):
        here = os.path.dirname(os.path.abspath(__file__))
        return library or os.path.join(os.path.dirname(here), "csrc", "libobvi_rccl.so")

    @staticmethod
    def nccl_version(library=None):
        """ncclGetVersion as libobvi_rccl.so resolves it in THIS process (obvi_rccl_nccl_version)."""
        lib = C.CDLL(RcclComm.library_path(library))
        lib.obvi_rccl_nccl_version.restype = C.c_int32
        return int(lib.obvi_rccl_nccl_version())

    @staticmethod
    def check_against_torch(library=None):
        """libobvi_rccl.so links /opt/rocm/lib/librccl, torch maps its own librccl.so into the same process.  Two different RCCL builds
        behind one set of symbols is not a configuration anybody has run with eight ranks: refuse (the caller falls back to the
        torch.distributed hook) unless both report the same version.  Returns (ok, ours, torchs)."""
        ours, theirs = RcclComm.nccl_version(library), torch_rccl_version()
        return (theirs is None or ours == theirs), ours, theirs

    def sequence(self):
        """(calls, hash) of the data-path collectives issued on this communicator so far (obvi_rccl_sequence)."""
        calls, h = C.c_uint64(), C.c_uint64()
        if self._lib.obvi_rccl_sequence(self._c, C.byref(calls), C.byref(h)) != 0:
            raise RuntimeError("obvi_rccl_sequence failed")
        return int(calls.value), int(h.value)

    def same_issue_order(self):
        """True iff every rank of the communicator has issued the same sequence of collectives (min == max of calls / hash halves)."""
        calls, h = self.sequence()
        v = [float(calls), float(h >> 32), float(h & 0xFFFFFFFF)]
        return self.host_allreduce(v, op=2) == self.host_allreduce(v, op=1)

    @staticmethod
    def unique_id(library=None):
        here = os.path.dirname(os.path.abspath(__file__))
        lib = C.CDLL(library or os.path.join(os.path.dirname(here), "csrc", "libobvi_rccl.so"))
        buf = C.create_string_buffer(RcclComm.ID_BYTES)
        if lib.obvi_rccl_unique_id(buf) != 0:
            raise RuntimeError("obvi_rccl_unique_id failed")
        return buf.raw

    def world(self):
        """ncclCommCount of the live communicator."""
        self._lib.obvi_rccl_comm_world.restype = C.c_int32
        return int(self._lib.obvi_rccl_comm_world(self._c))

    def attach(self, ba, is_shared):
        """obvi_ba_set_shared_objects + obvi_ba_set_allreduce(h, obvi_rccl_allreduce, comm): no Python in the solve's exchange."""
        import numpy as np
        m = None if is_shared is None else np.ascontiguousarray(is_shared, dtype=np.uint8)
        rc = self._lib.obvi_rccl_attach(ba._h, self._c, None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            raise RuntimeError("obvi_rccl_attach failed: status %d" % rc)
        self._keep = m

    def host_allreduce(self, values, op=0):
        """op 0 sum, 1 max, 2 min over a short list of doubles."""
        n = len(values)
        buf = (C.c_double * n)(*values)
        rc = self._lib.obvi_rccl_host_allreduce(self._c, buf, C.c_int32(n), C.c_int32(op))
        if rc != 0:
            raise RuntimeError("obvi_rccl_host_allreduce failed: status %d" % rc)
        return [buf[i] for i in range(n)]

    def barrier(self):
        if self._lib.obvi_rccl_barrier(self._c) != 0:
            raise RuntimeError("obvi_rccl_barrier failed")

    def close(self):
        if self._c:
            self._lib.obvi_rccl_comm_destroy.restype = None
            self._lib.obvi_rccl_comm_destroy(self._c)
            self._c = C.c_void_p()


class RcclGroup:
    """ctypes binding of the obvi_rccl_group_* part of include/obvi_rccl.h: k handles of THIS process (one host thread each) behind one
    inter-rank all-reduce per collective (config #5: sessions over one object map, 16 / N of them per GPU).  The inter-rank step is a
    communicator of libobvi_rccl.so (`comm`), any Python hook of the obvi_ba_set_allreduce shape (`inner`: torch_allreduce /
    staged_allreduce above), or nothing (a job of one rank)."""

    def __init__(self, n_members, comm=None, inner=None, rank=0, world=1, device=0, library=None):
        import obvi_ba
        self._lib = C.CDLL(RcclComm.library_path(library))
        self._g = C.c_void_p()
        self._keep = []
        self.n_members = n_members
        if comm is not None:
            rc = self._lib.obvi_rccl_group_create_on_comm(comm._c, C.c_int32(n_members), C.byref(self._g))
            self.rank, self.world = comm.rank, comm.world_requested
        else:
            cb = C.cast(None, obvi_ba.ALLREDUCE_FN) if inner is None else obvi_ba.ALLREDUCE_FN(lambda user, buf, count, op, stream: int(inner(buf or 0, count, op, stream or 0)))
            self._keep.append(cb)
            rc = self._lib.obvi_rccl_group_create(cb, None, C.c_int32(rank), C.c_int32(world), C.c_int32(n_members), C.c_int32(device), C.byref(self._g))
            self.rank, self.world = rank, world
        if rc != 0:
            raise RuntimeError("obvi_rccl_group_create failed: status %d" % rc)

    def attach(self, member, ba, is_shared):
        """The handle becomes contributor rank * k + member of world * k (obvi_ba_set_shared_objects) and exchanges through the group."""
        import numpy as np
        m = None if is_shared is None else np.ascontiguousarray(is_shared, dtype=np.uint8)
        rc = self._lib.obvi_rccl_group_attach(self._g, C.c_int32(member), ba._h, None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            raise RuntimeError("obvi_rccl_group_attach failed: status %d" % rc)
        self._keep.append(m)

    def set_timeout(self, seconds):
        self._lib.obvi_rccl_group_set_timeout.restype = None
        self._lib.obvi_rccl_group_set_timeout(self._g, C.c_double(seconds))

    def stats(self):
        """(collectives completed, doubles carried between ranks)."""
        a, b = C.c_uint64(), C.c_uint64()
        if self._lib.obvi_rccl_group_stats(self._g, C.byref(a), C.byref(b)) != 0:
            raise RuntimeError("obvi_rccl_group_stats failed")
        return int(a.value), int(b.value)

    def close(self):
        if self._g:
            self._lib.obvi_rccl_group_destroy.restype = None
            self._lib.obvi_rccl_group_destroy(self._g)
            self._g = C.c_void_p()


class HostGroup:
    """The group for libraries whose exchange buffers live in HOST memory (the CPU oracle; tests/test_distributed_gloo.py): k handles of this
    process, one thread each; per collective the last one to arrive sums / maximises the k buffers, runs ONE inter-rank all-reduce
    (`inner`, e.g. host_allreduce(dist); None = one rank) and copies the result back into every buffer.  Test plumbing."""

    def __init__(self, n_members, inner=None, timeout_s=300.0):
        import threading
        self.n, self.inner, self.timeout_s = n_members, inner, timeout_s
        self._bar = threading.Barrier(n_members)
        self._slots = [None] * n_members
        self.collectives, self.doubles = 0, 0

    def hook(self, member):
        import numpy as np

        def fn(ptr, count, op, stream):
            self._slots[member] = np.ctypeslib.as_array((C.c_double * int(count)).from_address(int(ptr)))
            self._bar.wait(self.timeout_s)
            if member == 0:
                st = np.stack(self._slots)
                acc = st.max(0) if op else st.sum(0)
                if self.inner is not None:
                    acc = np.ascontiguousarray(acc)
                    self.inner(acc.ctypes.data, count, op, stream)
                for s in self._slots:
                    s[:] = acc
                self.collectives += 1
                self.doubles += int(count)
            self._bar.wait(self.timeout_s)
            return 0
        return fn
