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


def _scene():
    import synth
    return synth.make_problem(P=60, L=900, O=3, seed=33, min_obj_obs=6, object_classes=("bench",), bbox_noise=5.0)   # the scene of tests/test_gpu_shared_objects.py


def _shared_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(helpers.ROOT, "obvi-slam_amd", "python")); sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    import numpy as np
    import dist_util, synth
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    wins, joint, keep_pts = helpers.split_problem(_scene(), 30)
    q = wins[rank][0]
    o = helpers.oracle_ba(); synth.upload(o, q)
    o.set_shared_objects(np.ones(len(q["objects"]), np.uint8), rank, world)
    log = dist_util.IssueLog()
    o.set_allreduce(dist_util.host_allreduce(dist, log))
    s = o.solve(helpers.ba_params(max_it=15))
    same = dist_util.same_issue_order(dist, log.calls, log.digest())
    # a rank that issued one collective more than the others is noticed
    skew = dist_util.same_issue_order(dist, log.calls + rank, log.digest())
    out[rank] = dict(iterations=s.num_iterations, termination=s.termination_type, initial=s.initial_cost, final=s.final_cost,
                     its=[(i.step_is_successful, i.cost, i.relative_decrease, i.step_norm, i.gradient_max_norm, i.trust_region_radius) for i in o.iterations()],
                     poses=o.get_poses(), points=o.get_points(), objects=o.get_objects(), calls=log.calls, records=list(log.records), same=same, skew=skew)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_objects_over_gloo_land_on_the_joint_solve():
    """SURVEY 8e on the CPU: two windows that share every object, one process each, the CPU oracle running the exchange protocol of
    include/obvi_ba.h on host buffers through dist_util.host_allreduce over gloo -- against the oracle's solve of the joint problem in
    one process (the same comparison tests/test_gpu_shared_objects.py makes for the HIP path with two ranks on one device).  Every rank
    reports the job-wide costs and takes the same decisions; poses, points and objects land on the joint solution; the ranks issue the
    same collectives in the same order (IssueLog / same_issue_order, which also notices a rank that is one call ahead)."""
    import numpy as np
    import synth
    world, port = 2, _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_shared_worker, args=(world, port, out), nprocs=world, join=True)
    wins, joint, keep_pts = helpers.split_problem(_scene(), 30)
    ref = helpers.oracle_ba(); synth.upload(ref, joint)
    sref = ref.solve(helpers.ba_params(max_it=15))
    jits = ref.iterations()
    a, b = out[0], out[1]
    assert a["its"] == b["its"] and a["iterations"] == b["iterations"] == sref.num_iterations and a["termination"] == sref.termination_type
    assert a["same"] and b["same"] and not a["skew"] and not b["skew"] and a["records"] == b["records"] and a["calls"] >= 1 + 4 * (sref.num_iterations - 1)
    for o in (a, b):
        assert abs(o["initial"] - sref.initial_cost) <= 1e-12 * sref.initial_cost and abs(o["final"] - sref.final_cost) <= 1e-8 * sref.final_cost
        for (ok, cost, rho, step, gmax, radius), j in zip(o["its"], jits):
            assert ok == j.step_is_successful and abs(cost - j.cost) <= 1e-8 * j.cost and abs(step - j.step_norm) <= 1e-6 * max(j.step_norm, 1e-12) and abs(gmax - j.gradient_max_norm) <= 1e-6 * j.gradient_max_norm
    jp, jo, jpts = ref.get_poses(), ref.get_objects(), ref.get_points()
    assert np.abs(a["poses"] - jp[:30]).max() < 1e-8 and np.abs(b["poses"] - jp[30:]).max() < 1e-8
    assert np.abs(a["objects"] - jo).max() < 1e-7 and np.array_equal(a["objects"], b["objects"])
    pos = {int(p): i for i, p in enumerate(keep_pts)}
    for o, (q, pts, rng) in zip((a, b), wins):
        assert np.abs(o["points"] - jpts[np.array([pos[int(p)] for p in pts])]).max() < 1e-7


def test_shared_objects_on_one_rank_are_the_plain_solve():
    """world = 1: the exchanges are identities (the hook leaves the buffers alone), the shared objects are merely eliminated last."""
    import numpy as np
    import synth
    prob = _scene()
    plain, shared = helpers.oracle_ba(), helpers.oracle_ba()
    for o in (plain, shared):
        synth.upload(o, prob)
    calls = []
    shared.set_shared_objects((np.arange(len(prob["objects"])) % 2 == 0).astype(np.uint8), 0, 1)
    shared.set_allreduce(lambda buf, count, op, stream: calls.append((count, op)) or 0)
    prm = helpers.ba_params(max_it=10)
    sp, ss = plain.solve(prm), shared.solve(prm)
    assert calls and all(op == 0 for _, op in calls)
    assert ss.num_iterations == sp.num_iterations and abs(ss.final_cost - sp.final_cost) <= 1e-9 * sp.final_cost
    assert np.abs(shared.get_poses() - plain.get_poses()).max() < 1e-8 and np.abs(shared.get_objects() - plain.get_objects()).max() < 1e-7


# ---- config #5 in small: sessions over ONE object map, several handles per rank ("config #5: 2 sessions per GPU") -------------

SESSIONS = dict(n_sessions=4, P=60, L=900, O=3, seed0=500, object_seed=33, min_obj_obs=6, object_classes=("bench",), bbox_noise=5.0)


def _sessions():
    import synth
    return synth.make_sessions(**SESSIONS)


def _session_group_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(helpers.ROOT, "obvi-slam_amd", "python")); sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    import threading
    import numpy as np
    import dist_util, synth
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sessions = _sessions()
    k = len(sessions) // world                                     # sessions per rank
    log = dist_util.IssueLog()
    group = dist_util.HostGroup(k, inner=dist_util.host_allreduce(dist, log))
    handles, res = [], [None] * k
    for m in range(k):
        q = sessions[rank * k + m]
        o = helpers.oracle_ba(); synth.upload(o, q)
        o.set_shared_objects(np.ones(len(q["objects"]), np.uint8), rank * k + m, world * k)   # contributor rank * k + m of world * k
        o.set_allreduce(group.hook(m))
        handles.append(o)

    def run(m):
        res[m] = handles[m].solve(helpers.ba_params(max_it=15))
    th = [threading.Thread(target=run, args=(m,)) for m in range(k)]
    [t.start() for t in th]; [t.join(timeout=600) for t in th]
    same = dist_util.same_issue_order(dist, log.calls, log.digest())
    out[rank] = dict(ok=all(r is not None for r in res), same=same, inter_rank_calls=log.calls, group_collectives=group.collectives,
                     members=[dict(iterations=r.num_iterations, termination=r.termination_type, initial=r.initial_cost, final=r.final_cost,
                                   its=[(i.step_is_successful, i.cost, i.step_norm, i.gradient_max_norm) for i in h.iterations()],
                                   poses=h.get_poses(), points=h.get_points(), objects=h.get_objects()) for r, h in zip(res, handles)])
    dist.barrier()
    dist.destroy_process_group()


def test_four_sessions_two_per_rank_over_gloo_land_on_the_joint_solve():
    """Config #5 as SURVEY 8e states it, on the CPU: four sessions over one object map, TWO per rank (two gloo ranks, two oracle handles each,
    one thread per handle), every object shared.  Per collective a rank first sums its own handles' buffers (dist_util.HostGroup, the host
    twin of obvi_rccl_group_*), then ONE inter-rank all-reduce carries the sum.  Every handle must follow the oracle's solve of the JOINT
    problem (all four sessions in one problem) step for step and land on its poses, features and objects."""
    import numpy as np
    import synth
    world, port = 2, _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_session_group_worker, args=(world, port, out), nprocs=world, join=True)
    sessions = _sessions()
    joint = synth.join_problems(sessions)
    ref = helpers.oracle_ba(); synth.upload(ref, joint)
    sref = ref.solve(helpers.ba_params(max_it=15))
    jits = ref.iterations()
    jp, jo, jpts = ref.get_poses(), ref.get_objects(), ref.get_points()
    po, lo = joint["session_pose_offsets"], joint["session_point_offsets"]
    assert set(out.keys()) == {0, 1} and all(out[r]["ok"] and out[r]["same"] for r in (0, 1))
    # one inter-rank collective per group collective: two handles per rank do not double the traffic between ranks
    assert out[0]["inter_rank_calls"] == out[0]["group_collectives"] == out[1]["inter_rank_calls"] >= 1 + 3 * (sref.num_iterations - 1)
    for rank in (0, 1):
        for m, o in enumerate(out[rank]["members"]):
            s = rank * 2 + m
            assert o["iterations"] == sref.num_iterations and o["termination"] == sref.termination_type
            assert abs(o["initial"] - sref.initial_cost) <= 1e-12 * sref.initial_cost and abs(o["final"] - sref.final_cost) <= 1e-8 * sref.final_cost
            for (ok, cost, step, gmax), j in zip(o["its"], jits):
                assert ok == j.step_is_successful and abs(cost - j.cost) <= 1e-8 * j.cost
                assert abs(step - j.step_norm) <= 1e-6 * max(j.step_norm, 1e-12) and abs(gmax - j.gradient_max_norm) <= 1e-6 * j.gradient_max_norm
            assert np.abs(o["poses"] - jp[po[s]:po[s + 1]]).max() < 1e-8 and np.abs(o["points"] - jpts[lo[s]:lo[s + 1]]).max() < 1e-7
            assert np.abs(o["objects"] - jo).max() < 1e-7 and np.array_equal(o["objects"], out[0]["members"][0]["objects"])
