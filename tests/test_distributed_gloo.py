"""CPU, world_size 2 over gloo: the N>1 path of bench.py (independent windows per rank, max-over-ranks timing).
The per-rank compute is stood in for by the CPU oracle on a tiny window so the test needs no GPU; what is
under test is the rank plumbing (seeds, reductions, aggregate value)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(helpers.ROOT, "obvi-slam_amd", "python")); sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    import dist_util, synth
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    r, lr, w = dist_util.rank_info()
    prob = synth.make_problem(P=10, L=60, O=0, seed=dist_util.rank_seed(20241008, 2, r))
    o = helpers.oracle_ba(); synth.upload(o, prob, relpose=False)
    s = o.solve(helpers.ba_params(max_it=3 + r, ftol=0, ptol=0, gtol=0))       # rank 1 runs one more iteration
    secs, steps = dist_util.reduce_timing(dist, "cpu", 0.5 * (r + 1), s.num_iterations - 1)
    costs = [torch.zeros(1, dtype=torch.float64) for _ in range(w)]
    dist.all_gather(costs, torch.tensor([s.final_cost], dtype=torch.float64))
    out[rank] = (secs, steps, dist_util.aggregate_throughput(w, steps, secs), [float(c) for c in costs], len(prob["rp_pose"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_over_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert set(out.keys()) == {0, 1}
    for r in (0, 1):
        secs, steps, value, costs, n_obs = out[r]
        assert secs == 1.0            # max over ranks of (0.5, 1.0)
        assert steps == 3             # min over ranks of (3, 4)
        assert value == pytest.approx(2 * 3 / 1.0)
        assert costs[0] != costs[1]   # different seeds -> different windows
    assert out[0][4] != out[1][4]
