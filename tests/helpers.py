"""Shared test plumbing: load the CPU oracle (checker) and the HIP product through the same binding."""
import os
import subprocess

import numpy as np

import obvi_ba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "libobvi_oracle.so")
PRODUCT_LIB = os.path.join(ROOT, "obvi-slam_amd", "csrc", "libobvi_ba.so")


def ensure_oracle():
    if not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ORACLE_LIB


def oracle_ba(**options):
    return obvi_ba.BundleAdjuster(library=ensure_oracle(), prefix="oracle_", **options)


def product_ba(device=0, **options):
    return obvi_ba.BundleAdjuster(device_id=device, library=PRODUCT_LIB, prefix="obvi_", **options)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def ba_params(max_it=50, nonmono=True, ftol=1e-6, radius=100.0, max_radius=1e4, gtol=1e-10, ptol=1e-8):
    """global_ba / local_ba style parameter block."""
    return obvi_ba.SolverParams(max_num_iterations=max_it, allow_non_monotonic_steps=nonmono, function_tolerance=ftol,
                                gradient_tolerance=gtol, parameter_tolerance=ptol, initial_trust_region_radius=radius,
                                max_trust_region_radius=max_radius)


def numpy_robust_residuals(prob):
    """Independent numpy / scipy restatement of the objective 1/2 sum rho(|r_b|^2) over every factor family of the path
    (synth.project_points, synth.project_ellipsoids, scipy Rotation, eigen-decomposition square roots; Huber per residual BLOCK,
    handed out as pre-robustified residuals r~ = r sqrt(rho(s)/s)).  Returns f(poses, points, objects) -> flat r~."""
    from scipy.spatial.transform import Rotation as Rot
    import synth

    def inv_sqrt(S):
        w, V = np.linalg.eigh(S)
        return (V / np.sqrt(w)) @ V.T
    W_bb = [inv_sqrt(c.reshape(4, 4)) for c in prob["bb_cov"]]
    W_sp = [inv_sqrt(c.reshape(3, 3)) for c in prob["sp_cov"]]
    W_rl = [inv_sqrt(c.reshape(6, 6)) for c in prob.get("rl_cov", [])]

    def robustified(r, delta):
        if len(r) == 0:
            return np.zeros(0)
        sq = (r * r).sum(axis=1)
        rho = np.where(sq > delta * delta, 2 * delta * np.sqrt(sq) - delta * delta, sq)
        return (r * np.sqrt(rho / np.maximum(sq, 1e-300))[:, None]).ravel()

    def f(poses, pts, objs):
        px, _ = synth.project_points(poses[prob["rp_pose"]], pts[prob["rp_point"]], prob["K"][0], prob["ext"][0])
        out = [robustified((px - prob["rp_pixel"]) / prob["rp_sigma"], prob["rp_huber"])]
        if len(prob["bb_obj"]):
            corners, valid, _ = synth.project_ellipsoids(objs[prob["bb_obj"]], poses[prob["bb_pose"]], prob["K"][0], prob["ext"][0])
            assert valid.all()
            out.append(robustified(np.stack([W @ d for W, d in zip(W_bb, corners - prob["bb_corners"])]), prob["bb_huber"]))
        if len(prob["sp_obj"]):
            out.append(robustified(np.stack([W @ d for W, d in zip(W_sp, objs[prob["sp_obj"], 4:7] - prob["sp_mean"])]), prob["sp_huber"]))
        if "rl_a" in prob and len(prob["rl_a"]):
            a, b = prob["rl_a"], prob["rl_b"]
            Ra = Rot.from_rotvec(poses[a, 3:6])
            t_rel = Ra.inv().apply(poses[b, :3] - poses[a, :3])
            rot = ((Ra.inv() * Rot.from_rotvec(poses[b, 3:6])) * Rot.from_rotvec(prob["rl_aa"]).inv()).as_rotvec()
            out.append(robustified(np.stack([W @ d for W, d in zip(W_rl, np.concatenate([t_rel - prob["rl_t"], rot], axis=1))]), prob["rl_huber"]))
        return np.concatenate(out)
    return f


def first_order_optimality_on_the_numpy_restatement(prob, poses, points, objects):
    """How good a minimum (poses, points, objects) is, judged by arithmetic that shares nothing with the solver: the numpy / scipy restatement of the objective
    (numpy_robust_residuals), its Jacobian by sparse central differences (scipy's grouped columns), and scipy.optimize.least_squares started AT the point.
    Returns (cost of the restatement at the point, largest |g_i| / (|J_i| |r|) over the parameters, relative cost gain scipy finds from there in 30 evaluations)."""
    from scipy.optimize import least_squares
    from scipy.optimize._numdiff import approx_derivative, group_columns
    from scipy.sparse import lil_matrix
    objective = numpy_robust_residuals(prob)
    P = len(prob["poses"])
    pv = np.flatnonzero(prob["pose_const"] == 0); pidx = -np.ones(P, int); pidx[pv] = np.arange(len(pv))
    nP, nL, nO = len(pv), len(prob["points"]), len(prob["objects"])

    def unpack(x):
        p = prob["poses"].copy(); p[pv] = x[:6 * nP].reshape(nP, 6)
        return p, x[6 * nP:6 * nP + 3 * nL].reshape(nL, 3), x[6 * nP + 3 * nL:].reshape(nO, 7)

    def f(x):
        return objective(*unpack(x))
    x = np.concatenate([np.asarray(poses)[pv].ravel(), np.asarray(points).ravel(), np.asarray(objects).ravel()])
    fam = [(2, [("p", prob["rp_pose"]), ("l", prob["rp_point"])]), (4, [("p", prob["bb_pose"]), ("o", prob["bb_obj"])]), (3, [("o", prob["sp_obj"])])]
    if "rl_a" in prob and len(prob["rl_a"]):
        fam.append((6, [("p", prob["rl_a"]), ("p", prob["rl_b"])]))
    m = sum(d * len(blocks[0][1]) for d, blocks in fam)
    S = lil_matrix((m, len(x)), dtype=np.int8)
    row = 0
    for d, blocks in fam:
        for i in range(len(blocks[0][1])):
            cols = []
            for kind, idx in blocks:
                j = int(idx[i])
                if kind == "p":
                    cols += list(range(6 * pidx[j], 6 * pidx[j] + 6)) if pidx[j] >= 0 else []
                elif kind == "l":
                    cols += list(range(6 * nP + 3 * j, 6 * nP + 3 * j + 3))
                else:
                    cols += list(range(6 * nP + 3 * nL + 7 * j, 6 * nP + 3 * nL + 7 * j + 7))
            for a in range(d):
                S[row + a, cols] = 1
            row += d
    assert row == m
    J = approx_derivative(f, x, method="3-point", sparsity=(S, group_columns(S)))
    r = f(x)
    g = J.T @ r
    colnorm = np.sqrt(np.asarray(J.multiply(J).sum(axis=0)).ravel())
    scaled = float((np.abs(g) / (colnorm * np.linalg.norm(r) + 1e-300)).max())
    cost = 0.5 * float(r @ r)
    ref = least_squares(f, x, method="trf", jac_sparsity=S, xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=30, tr_solver="lsmr")
    first_order_optimality_on_the_numpy_restatement.last_gradient = (g, colnorm * np.linalg.norm(r))      # order: variable poses, features, objects; and the scale of each entry
    return cost, scaled, (cost - float(ref.cost)) / cost


def map_rule(sq, active, fraction):
    """offline_problem_runner.h:769-800 stated in numpy: the active values as keys of a map in descending order (equal values are one
    entry; its member here: the highest index), the first floor(entries * fraction) entries go."""
    sq, active = np.asarray(sq, np.float64), np.asarray(active, bool)
    mask = active.astype(np.uint8)
    entries = {}
    for i in np.flatnonzero(active):
        entries[sq[i]] = i                      # ascending i: the highest index stays
    keys = sorted(entries, reverse=True)
    n_out = int(len(keys) * fraction)
    for k in keys[:n_out]:
        mask[entries[k]] = 0
    return mask, n_out


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
