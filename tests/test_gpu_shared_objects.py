"""GPU (-m gpu): independent windows that share object blocks.  One GPU is enough to check the algebra: two
handles are driven from two threads and the all-reduce hook is emulated by summing their exchange buffers; the result
must equal one handle solving the joint problem."""
import threading

import numpy as np
import pytest
import torch

import dist_util
import helpers
import obvi_ba
import synth

pytestmark = pytest.mark.gpu


def split_problem(prob, cut):
    """Two windows [0,cut) and [cut,P) sharing every object; points seen from both sides are dropped."""
    P = len(prob["poses"])
    side = (prob["rp_pose"] >= cut).astype(int)
    lo = np.full(len(prob["points"]), 2); hi = np.full(len(prob["points"]), -1)
    np.minimum.at(lo, prob["rp_point"], side); np.maximum.at(hi, prob["rp_point"], side)
    pt_side = np.where(lo == hi, lo, -1)
    wins = []
    for w, (a, b) in enumerate(((0, cut), (cut, P))):
        pts = np.nonzero(pt_side == w)[0]
        pmap = -np.ones(len(prob["points"]), int); pmap[pts] = np.arange(len(pts))
        r = (side == w) & (pt_side[prob["rp_point"]] == w)
        bb = (prob["bb_pose"] >= a) & (prob["bb_pose"] < b)
        rl = (prob["rl_a"] >= a) & (prob["rl_b"] < b)
        q = dict(K=prob["K"], ext=prob["ext"], poses=prob["poses"][a:b].copy(), pose_const=np.zeros(b - a, np.uint8), points=prob["points"][pts].copy(),
                 point_const=np.zeros(len(pts), np.uint8), objects=prob["objects"].copy(), object_const=np.zeros(len(prob["objects"]), np.uint8),
                 rp_pose=prob["rp_pose"][r] - a, rp_point=pmap[prob["rp_point"][r]], rp_cam=prob["rp_cam"][r], rp_pixel=prob["rp_pixel"][r],
                 rp_sigma=prob["rp_sigma"], rp_huber=prob["rp_huber"], bb_obj=prob["bb_obj"][bb], bb_pose=prob["bb_pose"][bb] - a, bb_cam=prob["bb_cam"][bb],
                 bb_corners=prob["bb_corners"][bb], bb_cov=prob["bb_cov"][bb], bb_huber=prob["bb_huber"], bb_invalid=prob["bb_invalid"],
                 sp_obj=prob["sp_obj"] if w == 0 else prob["sp_obj"][:0], sp_mean=prob["sp_mean"] if w == 0 else prob["sp_mean"][:0],
                 sp_cov=prob["sp_cov"] if w == 0 else prob["sp_cov"][:0], sp_huber=prob["sp_huber"],
                 rl_a=prob["rl_a"][rl] - a, rl_b=prob["rl_b"][rl] - a, rl_t=prob["rl_t"][rl], rl_aa=prob["rl_aa"][rl], rl_cov=prob["rl_cov"][rl], rl_huber=prob["rl_huber"])
        q["pose_const"][0] = 1
        wins.append((q, pts, (a, b)))
    # joint problem: both windows in one handle
    keep_pts = np.nonzero(pt_side >= 0)[0]
    jm = -np.ones(len(prob["points"]), int); jm[keep_pts] = np.arange(len(keep_pts))
    r = pt_side[prob["rp_point"]] >= 0
    r &= (side == pt_side[prob["rp_point"]])
    rl = ~((prob["rl_a"] < cut) & (prob["rl_b"] >= cut))
    joint = dict(prob)
    joint.update(points=prob["points"][keep_pts].copy(), point_const=np.zeros(len(keep_pts), np.uint8), rp_pose=prob["rp_pose"][r], rp_point=jm[prob["rp_point"][r]],
                 rp_cam=prob["rp_cam"][r], rp_pixel=prob["rp_pixel"][r], rl_a=prob["rl_a"][rl], rl_b=prob["rl_b"][rl], rl_t=prob["rl_t"][rl], rl_aa=prob["rl_aa"][rl],
                 rl_cov=prob["rl_cov"][rl], pose_const=np.zeros(P, np.uint8))
    joint["pose_const"][[0, cut]] = 1
    return wins, joint, keep_pts


class EmulatedAllReduce:
    """Stands in for RCCL: sums / maximises the exchange buffers of `world` handles living on one GPU."""
    def __init__(self, world):
        self.world, self.bar, self.slots, self.res, self.calls = world, threading.Barrier(world), [None] * world, None, 0

    def hook(self, rank):
        def fn(ptr, count, op, stream):
            torch.cuda.synchronize()
            t = dist_util.device_tensor(ptr, count)
            self.slots[rank] = t
            self.bar.wait()
            if rank == 0:
                st = torch.stack(self.slots)
                self.res = st.max(0).values if op else st.sum(0)
                self.calls += 1
            self.bar.wait()
            t.copy_(self.res)
            torch.cuda.synchronize()
            self.bar.wait()
            return 0
        return fn


@pytest.fixture(scope="module")
def scene():
    prob = synth.make_problem(P=60, L=900, O=3, seed=33, min_obj_obs=6, object_classes=("bench",), bbox_noise=5.0)
    return prob


def test_single_rank_identity_hook(scene):
    """world = 1: shared objects are merely ordered last and the three exchanges are identities -> same solve."""
    a, b = helpers.product_ba(), helpers.product_ba()
    for ba in (a, b):
        synth.upload(ba, scene)
    calls = []
    b.set_shared_objects(np.ones(len(scene["objects"]), np.uint8), 0, 1)
    b.set_allreduce(lambda ptr, n, op, stream: calls.append((n, op)) or 0)
    prm = helpers.ba_params(max_it=12)
    sa, sb = a.solve(prm), b.solve(prm)
    assert calls and sb.num_iterations == sa.num_iterations
    assert abs(sb.final_cost - sa.final_cost) <= 1e-9 * sa.final_cost
    assert np.abs(b.get_poses() - a.get_poses()).max() < 1e-9 and np.abs(b.get_objects() - a.get_objects()).max() < 1e-8
    # per LM submission: shared blocks (56 per object, sum), tail (sum), scalars (sum), gradient max (max)
    assert (56 * len(scene["objects"]), 0) in calls and (1, 1) in calls


def test_two_windows_sharing_objects_equal_the_joint_solve(scene):
    wins, joint, keep_pts = split_problem(scene, 30)
    ref = helpers.product_ba()
    synth.upload(ref, joint)
    prm = helpers.ba_params(max_it=15)
    sref = ref.solve(prm)
    emu = EmulatedAllReduce(2)
    handles, out = [], [None, None]
    for rank, (q, pts, rng) in enumerate(wins):
        ba = helpers.product_ba()
        synth.upload(ba, q)
        ba.set_shared_objects(np.ones(len(q["objects"]), np.uint8), rank, 2)
        ba.set_allreduce(emu.hook(rank))
        handles.append(ba)

    def run(rank):
        out[rank] = handles[rank].solve(prm)
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert all(o is not None for o in out) and emu.calls > 0
    for o in out:
        assert o.num_iterations == sref.num_iterations and o.termination_type == sref.termination_type
        assert abs(o.final_cost - sref.final_cost) <= 1e-8 * sref.final_cost        # every rank reports the job-wide cost
    jp, jo, jpts = ref.get_poses(), ref.get_objects(), ref.get_points()
    assert np.abs(handles[0].get_poses() - jp[:30]).max() < 1e-8 and np.abs(handles[1].get_poses() - jp[30:]).max() < 1e-8
    assert np.abs(handles[0].get_objects() - jo).max() < 1e-7 and np.abs(handles[1].get_objects() - handles[0].get_objects()).max() == 0.0
    pos = {int(p): i for i, p in enumerate(keep_pts)}
    for rank, (q, pts, rng) in enumerate(wins):
        idx = np.array([pos[int(p)] for p in pts])
        assert np.abs(handles[rank].get_points() - jpts[idx]).max() < 1e-7


def test_rccl_hook_through_torch_distributed(scene):
    """The production hook (dist_util.torch_allreduce: RCCL via torch.distributed, enqueued on the library's own stream
    through torch.cuda.ExternalStream) on a one-rank process group: the plumbing must leave the solve unchanged."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        a, b = helpers.product_ba(), helpers.product_ba()
        for ba in (a, b):
            synth.upload(ba, scene)
        b.set_shared_objects(np.ones(len(scene["objects"]), np.uint8), 0, 1)
        b.set_allreduce(dist_util.torch_allreduce(dist))
        prm = helpers.ba_params(max_it=10)
        sa, sb = a.solve(prm), b.solve(prm)
        assert sb.num_iterations == sa.num_iterations and abs(sb.final_cost - sa.final_cost) <= 1e-9 * sa.final_cost
        assert np.abs(b.get_poses() - a.get_poses()).max() < 1e-9
    finally:
        dist.destroy_process_group()
