"""CPU: the oracle's linear algebra and LM bookkeeping against brute-force numpy on small problems."""
import numpy as np

import helpers
import obvi_ba
import synth

FACTOR_BLOCKS = {0: (("pose", "rp_pose"), ("point", "rp_point")), 2: (("object", "bb_obj"), ("pose", "bb_pose")),
                 3: (("object", "sp_obj"),), 5: (("pose", "rl_a"), ("pose", "rl_b"))}
HUBER = {0: "rp_huber", 2: "bb_huber", 3: "sp_huber", 5: "rl_huber"}


def dense_normal_equations(ba, prob):
    """Robustified J and r of the reduced program, dense, columns = [variable poses, objects, points]."""
    P, L, O = len(prob["poses"]), len(prob["points"]), len(prob["objects"])
    pc = prob["pose_const"].astype(bool)
    pv = -np.ones(P, int); pv[~pc] = np.arange((~pc).sum())
    od = getattr(ba, "od", 7)                                         # parameters of an ellipsoid block (7, or 9: object_block_size)
    nPv = int((~pc).sum()); m = 6 * nPv + od * O; n = m + 3 * L
    col = {"pose": lambda i: None if pv[i] < 0 else (6 * pv[i], 6), "object": lambda i: (6 * nPv + od * i, od), "point": lambda i: (m + 3 * i, 3)}
    rows, rr = [], []
    for t, blocks in FACTOR_BLOCKS.items():
        if ba.num_factors(t) == 0:
            continue
        r, J0, J1 = ba.debug_linearize(t)
        a = prob[HUBER[t]]
        s = (r ** 2).sum(axis=1)
        w = np.sqrt(np.where(s > a * a, a / np.sqrt(np.maximum(s, 1e-300)), 1.0))
        for f in range(len(r)):
            blk = np.zeros((r.shape[1], n)); anyvar = False
            for bi, (kind, key) in enumerate(blocks):
                c = col[kind](prob[key][f])
                if c is None:
                    continue
                anyvar = True
                blk[:, c[0]:c[0] + c[1]] += w[f] * (J0 if bi == 0 else J1)[f]
            if anyvar:
                rows.append(blk); rr.append(w[f] * r[f])
    return np.vstack(rows), np.concatenate(rr), m, pv


def lm_system(J, r, radius):
    H, g = J.T @ J, J.T @ r
    c = np.diag(H).copy(); sc = 1 / (1 + np.sqrt(c))
    lam = np.clip(c * sc * sc, 1e-6, 1e32) / radius / (sc * sc)
    return H + np.diag(lam), g


def small_problem(seed=5, const_poses=2):
    return synth.make_problem(P=12, L=30, O=2, seed=seed, min_obj_obs=4, object_classes=("bench", "chair"), const_poses=const_poses)


def test_gradient_matches_finite_differences():
    prob = small_problem()
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    x0 = [prob["poses"].copy(), prob["points"].copy(), prob["objects"].copy()]

    def cost_at(x):
        ba.set_poses(x[0], prob["pose_const"]); ba.set_points(x[1], prob["point_const"]); ba.set_objects(x[2], prob["object_const"])
        return ba.evaluate(True, False)[0]
    cost_at(x0)
    g = [np.zeros_like(a) for a in x0]
    kind_idx = {"pose": 0, "point": 1, "object": 2}
    for t, blocks in FACTOR_BLOCKS.items():
        r, J0, J1 = ba.debug_linearize(t)
        a = prob[HUBER[t]]; s = (r ** 2).sum(axis=1)
        w = np.where(s > a * a, a / np.sqrt(np.maximum(s, 1e-300)), 1.0)
        for bi, (kind, key) in enumerate(blocks):
            np.add.at(g[kind_idx[kind]], prob[key], np.einsum("n,nmd,nm->nd", w, J0 if bi == 0 else J1, r))
    rng = np.random.default_rng(0)
    eps = 1e-6
    for kind in range(3):
        for _ in range(25):
            i = rng.integers(x0[kind].shape[0]); k = rng.integers(x0[kind].shape[1])
            xp = [a.copy() for a in x0]; xm = [a.copy() for a in x0]
            xp[kind][i, k] += eps; xm[kind][i, k] -= eps
            gn = (cost_at(xp) - cost_at(xm)) / (2 * eps)
            assert abs(gn - g[kind][i, k]) <= 2e-5 * (1 + abs(gn))


def test_schur_complement_and_step_match_dense_solve():
    prob = small_problem()
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    J, r, m, pv = dense_normal_equations(ba, prob)
    radius = 100.0
    A, g = lm_system(J, r, radius)
    S = A[:m, :m] - A[:m, m:] @ np.linalg.solve(A[m:, m:], A[:m, m:].T)
    b = g[:m] - A[:m, m:] @ np.linalg.solve(A[m:, m:], g[m:])
    So, bo = ba.debug_reduced_system(radius)
    assert So.shape == (m, m)
    assert helpers.rel_err(So, S) < 1e-12 and helpers.rel_err(bo, b) < 1e-11
    delta = -np.linalg.solve(A, g)
    before = [ba.get_poses(), ba.get_points(), ba.get_objects()]
    s = ba.solve(helpers.ba_params(max_it=1, ftol=0, ptol=0, gtol=0, radius=radius))
    its = ba.iterations()
    assert s.termination_type == obvi_ba.NO_CONVERGENCE and len(its) == 2 and its[1].step_is_successful
    d = np.concatenate([(ba.get_poses() - before[0])[pv >= 0].ravel(), (ba.get_objects() - before[2]).ravel(), (ba.get_points() - before[1]).ravel()])
    assert helpers.rel_err(d, delta) < 1e-11
    assert abs(its[1].step_norm - np.linalg.norm(delta)) < 1e-10 * np.linalg.norm(delta)


def nine_problem(tilt=0.3, ltm=True):
    prob = small_problem()
    if ltm:
        O = len(prob["objects"]); rng = np.random.default_rng(4); A = rng.normal(size=(O, 7, 7))
        prob.update(lt_obj=np.arange(O, dtype=np.uint32), lt_mean=prob["gt_objects"] + 0.05, lt_cov=(A @ A.transpose(0, 2, 1) + 7 * np.eye(7)).reshape(O, 49) * 0.01, lt_huber=1.0)
    return synth.nine_dof(prob, tilt=tilt, seed=1)


def test_nine_parameter_ellipsoid_block_step_matches_the_dense_solve_and_finite_differences():
    """obvi_ba_options.object_block_size = 9 (vslam_obj_opt_types_refactor.h:15-21, ellipsoid_utils.h:217-229 `#else`): the oracle's linearisation of every
    object factor (bounding box 4 x 9, shape prior 3 x 9, map prior 9 x 9), its reduced system and its LM step against a dense numpy solve of the same
    damped normal equations, and its gradient against finite differences of its own cost."""
    prob = nine_problem()
    ba = helpers.oracle_ba(object_block_size=9); synth.upload(ba, prob)
    assert ba.get_objects().shape == (len(prob["objects"]), 9) and ba.num_residuals() == 2 * ba.num_factors(0) + 4 * ba.num_factors(2) + 3 * ba.num_factors(3) + 9 * ba.num_factors(4) + 6 * ba.num_factors(5)
    FACTOR_BLOCKS[4] = (("object", "lt_obj"),); HUBER[4] = "lt_huber"
    try:
        J, r, m, pv = dense_normal_equations(ba, prob)
    finally:
        del FACTOR_BLOCKS[4], HUBER[4]
    radius = 100.0
    A, g = lm_system(J, r, radius)
    S = A[:m, :m] - A[:m, m:] @ np.linalg.solve(A[m:, m:], A[:m, m:].T)
    b = g[:m] - A[:m, m:] @ np.linalg.solve(A[m:, m:], g[m:])
    So, bo = ba.debug_reduced_system(radius)
    assert So.shape == (m, m) and helpers.rel_err(So, S) < 1e-12 and helpers.rel_err(bo, b) < 1e-11
    x0 = [prob["poses"].copy(), prob["points"].copy(), prob["objects"].copy()]

    def cost_at(x):
        ba.set_poses(x[0], prob["pose_const"]); ba.set_points(x[1], prob["point_const"]); ba.set_objects(x[2], prob["object_const"])
        return ba.evaluate(True, False)[0]
    nPv = int((pv >= 0).sum())
    for o in range(len(prob["objects"])):
        for k in range(9):
            h = 1e-6
            xp = [a.copy() for a in x0]; xm = [a.copy() for a in x0]
            xp[2][o, k] += h; xm[2][o, k] -= h
            fd = (cost_at(xp) - cost_at(xm)) / (2 * h)
            assert abs(fd - g[6 * nPv + 9 * o + k]) < 1e-5 * max(1.0, abs(fd)), (o, k, fd, g[6 * nPv + 9 * o + k])
    cost_at(x0)
    delta = -np.linalg.solve(A, g)
    s = ba.solve(helpers.ba_params(max_it=1, ftol=0, ptol=0, gtol=0, radius=radius))
    its = ba.iterations()
    assert len(its) == 2 and its[1].step_is_successful
    d = np.concatenate([(ba.get_poses() - x0[0])[pv >= 0].ravel(), (ba.get_objects() - x0[2]).ravel(), (ba.get_points() - x0[1]).ravel()])
    assert helpers.rel_err(d, delta) < 1e-11
    # covariance blocks are 9 x 9 blocks of the dense inverse (no damping)
    ba.set_poses(*[x0[0], prob["pose_const"]]); ba.set_points(x0[1], prob["point_const"]); ba.set_objects(x0[2], prob["object_const"])
    cov = ba.object_covariances(np.arange(len(prob["objects"])))
    Hinv = np.linalg.inv(J.T @ J)
    for o in range(len(prob["objects"])):
        blk = Hinv[6 * nPv + 9 * o:6 * nPv + 9 * o + 9, 6 * nPv + 9 * o:6 * nPv + 9 * o + 9]
        assert cov[o].shape == (9, 9) and helpers.rel_err(cov[o], blk) < 1e-8


def test_a_nine_block_with_upright_objects_reproduces_the_seven_block():
    """A 9-parameter problem whose objects are upright -- rotation vector (0, 0, yaw) -- is the 7-parameter problem seen through a larger block: the same
    costs and residuals, and the reduced system with the rows of (ax, ay) struck out IS the 7-block's reduced system (the features' Schur complement does
    not involve the objects; d/d(az) = d/d(yaw) on the z axis).  VERDICT r5 item 5: "the 7 path's numbers must not move"."""
    prob7 = small_problem()
    prob9 = synth.nine_dof(prob7, tilt=0.0)
    b7 = helpers.oracle_ba(); synth.upload(b7, prob7)
    b9 = helpers.oracle_ba(object_block_size=9); synth.upload(b9, prob9)
    c7, r7, _ = b7.evaluate(True, True); c9, r9, _ = b9.evaluate(True, True)
    assert abs(c9 - c7) <= 1e-13 * c7 and np.abs(r9 - r7).max() < 1e-11
    S7, g7 = b7.debug_reduced_system(100.0); S9, g9 = b9.debug_reduced_system(100.0)
    nPv = int((prob7["pose_const"] == 0).sum()); O = len(prob7["objects"])
    keep = list(range(6 * nPv)) + [6 * nPv + 9 * o + k for o in range(O) for k in (0, 1, 2, 5, 6, 7, 8)]
    assert S9.shape[0] == 6 * nPv + 9 * O and S7.shape[0] == 6 * nPv + 7 * O
    assert helpers.rel_err(S9[np.ix_(keep, keep)], S7) < 1e-11 and helpers.rel_err(g9[keep], g7) < 1e-10


def test_fixed_cost_and_constant_blocks():
    prob = small_problem(const_poses=3)
    prob["point_const"][:5] = 1
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=3))
    assert s.fixed_cost > 0.0                       # observations of constant points from constant poses
    assert np.array_equal(ba.get_poses()[:3], prob["poses"][:3])
    assert np.array_equal(ba.get_points()[:5], prob["points"][:5])
    assert s.final_cost < s.initial_cost
    # everything constant: nothing to optimise
    ba.set_const_flags(np.ones(12, np.uint8), np.ones(30, np.uint8), np.ones(2, np.uint8))
    s = ba.solve(helpers.ba_params(max_it=3))
    assert s.termination_type == obvi_ba.CONVERGENCE and s.num_iterations == 1 and s.num_parameters_reduced == 0


def test_noise_free_problem_stays_at_truth():
    """a BA started from ground truth with noiseless measurements must stay at ~0 cost."""
    prob = synth.make_problem(P=10, L=60, O=0, seed=9, outlier_frac=0.0, pixel_noise=0.0, point_noise=0.0)
    prob["poses"] = prob["gt_poses"].copy()
    ba = helpers.oracle_ba(); synth.upload(ba, prob, relpose=False)
    assert ba.evaluate(True, False)[0] < 1e-18
    s = ba.solve(helpers.ba_params(max_it=5))
    assert s.final_cost < 1e-18 and np.abs(ba.get_poses() - prob["gt_poses"]).max() < 1e-9


def test_outlier_selection_semantics():
    prob = small_problem()
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    _, _, sq = ba.evaluate(False)
    n = ba.num_factors(0)
    mask, nex = ba.select_outliers(0, 0.1)
    assert nex == int(len(np.unique(sq[:n])) * 0.1) and (mask == 0).sum() == nex
    assert sq[:n][mask == 0].min() >= np.sort(sq[:n])[-nex]          # the largest residual blocks are the ones dropped
    # phase II: excluded factors leave the problem
    c0 = ba.evaluate(False)[0]
    ba.set_active_mask(0, mask)
    c1, res, sq1 = ba.evaluate(False)
    assert c1 < c0 and np.all(sq1[:n][mask == 0] == 0.0)
    ba.set_active_mask(0, None)
    assert abs(ba.evaluate(False)[0] - c0) < 1e-9 * c0


def test_snapshot_restore():
    prob = small_problem()
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    ba.snapshot()
    ba.solve(helpers.ba_params(max_it=3))
    assert np.abs(ba.get_points() - prob["points"]).max() > 0
    ba.restore()
    assert np.array_equal(ba.get_points(), prob["points"]) and np.array_equal(ba.get_poses(), prob["poses"])


def test_nonmonotonic_returns_minimum_cost_iterate():
    prob = synth.make_problem(P=20, L=150, O=2, seed=2, min_obj_obs=4)
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=25, nonmono=True))
    costs = [it.cost for it in ba.iterations()]
    assert abs(s.final_cost - min(costs)) < 1e-12 * min(costs)
    assert abs(ba.evaluate(True, False)[0] - s.final_cost) < 1e-9 * s.final_cost


def test_iteration_log_bookkeeping():
    """[Ceres-doc trust_region_minimizer.cc] iteration 0 counts as a successful step (successful + unsuccessful == iterations.size());
    a rejected step records the CANDIDATE's cost (which is why it was rejected: it is above the current one, or barely below it);
    final_cost is the minimum over the log."""
    prob = synth.make_problem(P=20, L=150, O=2, seed=2, min_obj_obs=4)
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=40, nonmono=False, radius=1e6, max_radius=1e12, ftol=1e-12))
    its = ba.iterations()
    assert s.num_successful_steps + s.num_unsuccessful_steps == s.num_iterations == len(its)
    assert its[0].step_is_successful and s.num_successful_steps == sum(i.step_is_successful for i in its)
    rejected = [k for k, i in enumerate(its) if i.step_is_valid and not i.step_is_successful]
    assert rejected, "the wide initial radius must produce at least one rejected step"
    for k in rejected:
        cur = [i.cost for i in its[:k] if i.step_is_successful][-1]
        assert abs(its[k].cost - (cur - its[k].cost_change)) <= 1e-12 * abs(cur)      # cost = candidate cost = x_cost - cost_change
        assert its[k].cost != cur
    assert abs(s.final_cost - min(i.cost for i in its)) <= 1e-15 * s.final_cost


def test_parameter_priors_enter_the_covariance_as_jacobian_rows():
    """oracle_ba_object_covariances with ParameterPrior factors == the blocks of the dense inverse of J^T J + diag(1 / std^2)."""
    prob = small_problem()
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    base = ba.object_covariances(np.arange(len(prob["objects"])))
    kind, blk, par, sd = [2, 2, 0, 1], [0, 1, 3, 5], [0, 3, 2, 1], [0.05, 0.02, 0.01, 0.03]
    mean = [prob["objects"][0, 0], prob["objects"][1, 3], prob["poses"][3, 2], prob["points"][5, 1]]
    ba.set_parameter_priors(kind, blk, par, mean, sd)
    with_pr = ba.object_covariances(np.arange(len(prob["objects"])))
    assert with_pr[0][0, 0] < base[0][0, 0] and with_pr[1][3, 3] < base[1][3, 3]        # information was added
    assert with_pr[0][0, 0] <= sd[0] ** 2 and with_pr[1][3, 3] <= sd[1] ** 2
    # dense check: H = J^T J from the reduced system at radius -> infinity, objects last; a prior on an object parameter is a diagonal term
    S0, _ = ba.debug_reduced_system(1e300)
    ba2 = helpers.oracle_ba(); synth.upload(ba2, prob)
    only_obj = ([2, 2], [0, 1], [0, 3], mean[:2], sd[:2])
    ba2.set_parameter_priors(*only_obj)
    got = ba2.object_covariances(np.arange(len(prob["objects"])))
    m = S0.shape[0]; nO = len(prob["objects"]); r0 = m - 7 * nO
    S1 = S0.copy(); S1[r0 + 0, r0 + 0] += 1 / sd[0] ** 2; S1[r0 + 7 + 3, r0 + 7 + 3] += 1 / sd[1] ** 2
    inv = np.linalg.inv(S1)
    for k in range(nO):
        blk_ = inv[r0 + 7 * k:r0 + 7 * k + 7, r0 + 7 * k:r0 + 7 * k + 7]
        assert np.abs(got[k] - blk_).max() <= 1e-8 * np.abs(blk_).max()
    p6, l3, o7 = ba.column_sqnorms()
    assert np.all(p6[prob["pose_const"] == 1] == -1) and np.all(o7 > 0) and o7[0, 0] > 1 / sd[0] ** 2


def test_the_oracle_is_bit_identical_for_any_number_of_host_threads():
    """bench.py's cpu_baseline and the committed end states run the oracle with min(20, hardware) host threads (the reference's num_threads = 20).
    Since round 6 every sum of the oracle adds the same terms in the same order whatever the thread count is (the points' Schur contributions
    row by row by one thread each, the trial cost as one running sum, the arrow split of the skyline factor): the LM trajectory, the blocks it
    ends at and the covariance blocks are the SAME BITS at 1, 3, 4 and 7 threads (a checker whose result depends on its
    thread count cannot back a committed fixture)."""
    import ctypes
    lib = ctypes.CDLL(helpers.ensure_oracle())
    prob = synth.make_problem(P=60, L=2500, O=4, seed=9, min_obj_obs=5, object_classes=("bench",), bbox_noise=5.0)
    out = []
    try:
        for threads in (1, 3, 4, 7):
            lib.oracle_set_threads(ctypes.c_int32(threads))
            ba = helpers.oracle_ba(); synth.upload(ba, prob)
            s = ba.solve(helpers.ba_params(max_it=6, ftol=0, gtol=0, ptol=0))
            its = ba.iterations()
            out.append((s.num_iterations, np.array([i.cost for i in its]), np.array([i.step_norm for i in its]), [i.step_is_successful for i in its],
                        ba.get_poses(), ba.get_points(), ba.get_objects(), ba.object_covariances(np.arange(4)), ba.evaluate(True, True)))
    finally:
        lib.oracle_set_threads(ctypes.c_int32(1))
    ref = out[0]
    assert ref[0] == 7 and sum(ref[3]) >= 3
    for other in out[1:]:
        assert other[0] == ref[0] and other[3] == ref[3]
        for a, b in zip(other, ref):
            if isinstance(a, np.ndarray):
                assert np.array_equal(a, b)
        assert other[8][0] == ref[8][0] and np.array_equal(other[8][1], ref[8][1])


def test_flat_problem_file_for_the_ceres_harness(tmp_path):
    """synth.dump_flat writes what oracle/ceres_harness/ceres_harness.cpp reads (the harness itself only builds where Ceres is)."""
    import struct
    prob = small_problem()
    path = tmp_path / "p.flat"
    synth.dump_flat(prob, str(path), max_it=4)
    raw = path.read_bytes()
    assert raw[:8] == b"OBVIFLT1"
    n_k = struct.unpack_from("<Q", raw, 8)[0]
    assert n_k == prob["K"].size and struct.unpack_from("<%dd" % n_k, raw, 16) == tuple(prob["K"].ravel())
    assert struct.unpack_from("<7d", raw, len(raw) - 56)[0] == 4.0


def test_converged_minimum_matches_scipy_on_the_numpy_restatement():
    """Solver-level pin that does not involve the oracle's own arithmetic: the minimiser of 1/2 sum rho(|r_b|^2) the oracle's LM loop
    converges to must be the one scipy.optimize.least_squares finds for residuals computed by the independent numpy restatement
    of the projection (synth.project_points, scipy Rotation).  Ceres applies Huber per residual BLOCK; scipy's built-in losses are
    per component, so the blocks are handed over pre-robustified: r~ = r sqrt(rho(s)/s) has |r~|^2 = rho(s)."""
    from scipy.optimize import least_squares
    prob = synth.make_problem(P=8, L=60, O=0, seed=7, with_relpose=False, const_poses=2, outlier_frac=0.1, pixel_noise=1.0)
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=120, ftol=1e-15, gtol=1e-14, ptol=1e-14, radius=1e4, max_radius=1e12))
    pv = np.flatnonzero(prob["pose_const"] == 0)
    nP, nL, delta, sigma = len(pv), len(prob["points"]), prob["rp_huber"], prob["rp_sigma"]

    def unpack(x):
        poses = prob["poses"].copy(); poses[pv] = x[:6 * nP].reshape(nP, 6)
        return poses, x[6 * nP:].reshape(nL, 3)

    def robust_residuals(x):
        poses, pts = unpack(x)
        px, _ = synth.project_points(poses[prob["rp_pose"]], pts[prob["rp_point"]], prob["K"][0], prob["ext"][0])
        r = (px - prob["rp_pixel"]) / sigma
        sq = (r * r).sum(axis=1)
        rho = np.where(sq > delta * delta, 2 * delta * np.sqrt(sq) - delta * delta, sq)
        return (r * np.sqrt(rho / np.maximum(sq, 1e-300))[:, None]).ravel()

    x0 = np.concatenate([prob["poses"][pv].ravel(), prob["points"].ravel()])
    assert abs(0.5 * (robust_residuals(x0) ** 2).sum() - s.initial_cost) <= 1e-9 * s.initial_cost   # same objective at the start
    ref = least_squares(robust_residuals, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=2000)
    assert abs(ref.cost - s.final_cost) <= 1e-8 * s.final_cost
    x_oracle = np.concatenate([ba.get_poses()[pv].ravel(), ba.get_points().ravel()])
    assert abs(0.5 * (robust_residuals(x_oracle) ** 2).sum() - s.final_cost) <= 1e-9 * s.final_cost   # and at the oracle's minimiser
    # first-order optimality of the oracle's minimiser under the independent objective (finite-difference Jacobian)
    from scipy.optimize._numdiff import approx_derivative
    grad = lambda x: approx_derivative(robust_residuals, x, method="3-point").T @ robust_residuals(x)   # noqa: E731
    assert np.abs(grad(x_oracle)).max() <= 1e-4 * np.abs(grad(x0)).max()    # 1e-5 is the noise floor of the finite differences (scipy's own minimiser: the same)
    # the minimum is flat along the viewing rays of weakly observed points (one of them drifts off to 'infinity' in both solvers, at
    # no cost): the poses are compared, the points only through the cost above
    poses, _ = unpack(ref.x)
    assert np.abs(ba.get_poses() - poses).max() < 1e-4


def test_all_factor_families_minimum_matches_scipy():
    """The same pin with every factor family of the path (reprojection, bounding box, shape prior, relative pose), each restated in
    numpy / scipy (synth.project_points, synth.project_ellipsoids, scipy Rotation, eigen-decomposition square roots): equal
    objective at the start, equal minimum."""
    from scipy.optimize import least_squares
    prob = synth.make_problem(P=10, L=40, O=2, seed=11, const_poses=1, outlier_frac=0.05, min_obj_obs=4, object_classes=("bench", "chair"), bbox_noise=5.0)
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    s = ba.solve(helpers.ba_params(max_it=150, ftol=1e-15, gtol=1e-14, ptol=1e-14, radius=1e4, max_radius=1e12))

    objective = helpers.numpy_robust_residuals(prob)
    pv = np.flatnonzero(prob["pose_const"] == 0)
    nP, nL, nO = len(pv), len(prob["points"]), len(prob["objects"])

    def unpack(x):
        poses = prob["poses"].copy(); poses[pv] = x[:6 * nP].reshape(nP, 6)
        return poses, x[6 * nP:6 * nP + 3 * nL].reshape(nL, 3), x[6 * nP + 3 * nL:].reshape(nO, 7)

    def residuals(x):
        return objective(*unpack(x))

    x0 = np.concatenate([prob["poses"][pv].ravel(), prob["points"].ravel(), prob["objects"].ravel()])
    assert abs(0.5 * (residuals(x0) ** 2).sum() - s.initial_cost) <= 1e-12 * s.initial_cost
    ref = least_squares(residuals, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=3000)
    # weakly constrained parameters (a point along its viewing ray, an object behind saturated Huber blocks) drift in both solvers
    # -- steps of metres for no change of the cost -- so neither stops on a tolerance; the minima agree to 1e-6
    assert abs(ref.cost - s.final_cost) <= 1e-6 * s.final_cost
    x_oracle = np.concatenate([ba.get_poses()[pv].ravel(), ba.get_points().ravel(), ba.get_objects().ravel()])
    assert abs(0.5 * (residuals(x_oracle) ** 2).sum() - s.final_cost) <= 1e-10 * s.final_cost
    poses, _, _ = unpack(ref.x)
    assert np.abs(ba.get_poses() - poses).max() < 1e-4    # points / objects: through the cost (flat directions)


def test_minimum_of_a_forty_frame_problem_is_a_minimum_of_the_numpy_restatement():
    """The pin above at forty times the size (40 keyframes / 400 features / 3 objects, every factor family): running scipy from the start takes minutes there, but
    whether the point the LM loop converged to IS a minimum of the independent restatement takes seconds -- the restatement's cost at it equals the solver's, its
    gradient (sparse central differences, no solver Jacobian involved) vanishes to the differences' noise, and scipy started at the point finds nothing lower.
    (A wrong fixed point -- a factor mis-weighted, a block left out of the reduced system -- leaves a scaled gradient of 1e-2 ... 1.)"""
    prob = synth.make_problem(P=40, L=400, O=3, seed=11, const_poses=2, outlier_frac=0.05, min_obj_obs=5, object_classes=("bench",), bbox_noise=5.0, min_parallax_deg=3.0)
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    # at the START, where the gradient is large: the oracle's own gradient J^T r (its dual-number Jacobians of every family, its Huber weights) against the restatement's
    J, r, m, pv = dense_normal_equations(ba, prob)
    g_oracle = J.T @ r
    nP, nO = int((pv >= 0).sum()), len(prob["objects"])
    g_oracle = np.concatenate([g_oracle[:6 * nP], g_oracle[6 * nP + 7 * nO:], g_oracle[6 * nP:6 * nP + 7 * nO]])          # -> poses, features, objects
    helpers.first_order_optimality_on_the_numpy_restatement(prob, prob["poses"], prob["points"], prob["objects"])
    g_fd, scale = helpers.first_order_optimality_on_the_numpy_restatement.last_gradient
    assert np.abs(g_fd).max() > 1e3 and (np.abs(g_oracle - g_fd) / scale).max() < 1e-5
    s = ba.solve(helpers.ba_params(max_it=200, ftol=1e-15, gtol=1e-14, ptol=1e-14, radius=1e4, max_radius=1e12))
    cost, scaled_gradient, gain = helpers.first_order_optimality_on_the_numpy_restatement(prob, ba.get_poses(), ba.get_points(), ba.get_objects())
    assert abs(cost - s.final_cost) <= 1e-10 * s.final_cost and scaled_gradient < 1e-5 and gain < 1e-9, (cost, s.final_cost, scaled_gradient, gain)
    # and the check can tell: the same point with one object's centre 2 cm off is not a minimum
    off = ba.get_objects(); off[0, 0] += 0.02
    assert helpers.first_order_optimality_on_the_numpy_restatement(prob, ba.get_poses(), ba.get_points(), off)[1] > 1e-4


def test_object_covariances_are_blocks_of_the_dense_inverse():
    """ceres::Covariance on object blocks (long_term_object_map_extraction.cpp:419-433): the oracle's Schur-complement route
    against numpy's inverse of the full J^T J (poses, objects and points, J robustified, no damping)."""
    prob = small_problem()
    ba = helpers.oracle_ba(); synth.upload(ba, prob)
    J, r, m, pv = dense_normal_equations(ba, prob)
    C = np.linalg.inv(J.T @ J)
    nPv, O = int((pv >= 0).sum()), len(prob["objects"])
    row = lambda o: 6 * nPv + 7 * o
    own = ba.object_covariances(np.arange(O))
    for o in range(O):
        blk = C[row(o):row(o) + 7, row(o):row(o) + 7]
        assert np.abs(own[o] - blk).max() <= 1e-9 * np.abs(blk).max()
        assert np.allclose(own[o], own[o].T, rtol=0, atol=1e-12 * np.abs(blk).max()) and np.all(np.linalg.eigvalsh(own[o]) > 0)
    cross = ba.object_covariances([0, 1], [1, 0])
    blk = C[row(0):row(0) + 7, row(1):row(1) + 7]
    assert np.abs(cross[0] - blk).max() <= 1e-9 * np.abs(own).max() and np.abs(cross[1] - blk.T).max() <= 1e-9 * np.abs(own).max()
    # a constant object has no covariance, and the others no longer share uncertainty with it
    prob2 = dict(prob); prob2["object_const"] = prob["object_const"].copy(); prob2["object_const"][0] = 1
    ba2 = helpers.oracle_ba(); synth.upload(ba2, prob2)
    c2 = ba2.object_covariances(np.arange(O))
    assert np.all(c2[0] == 0.0) and np.all(np.diag(c2[1]) <= np.diag(own[1]) * (1 + 1e-9))


def test_oracle_selection_rule_against_the_map_rule_in_numpy():
    """oracle_ba_debug_select (the rule of offline_problem_runner.h:769-800 as the oracle applies it in oracle_ba_select_outliers) against
    helpers.map_rule: ties, inactive factors, zeros, the fractions' edge cases."""
    o = helpers.oracle_ba()
    rng = np.random.default_rng(12)
    for n in (0, 1, 2, 63, 5000):
        for sq, active in ((np.exp(rng.normal(size=n) * 3.0), rng.random(n) < 0.9), (rng.integers(0, 7, size=n).astype(np.float64), rng.random(n) < 0.8),
                           (np.round(np.exp(rng.normal(size=n)), 2), None)):
            act = np.ones(n, bool) if active is None else active
            for fraction in (0.0, 1e-4, 0.1, 0.37, 1.0):
                want, n_want = helpers.map_rule(sq, act, fraction)
                got, n_got = o.debug_select(sq, active, fraction)
                assert n_got == n_want and np.array_equal(got, want), (n, fraction)


def test_arbiter_build_really_runs_in_extended_precision():
    """oracle/libobvi_oracle_ld.so (`make arbiter`: the oracle's source with `double` = long double in the factors and OBVI_ORACLE_REAL = long
    double in every solver-level sum) is quoted as ground truth in DESIGN.md section 6.  The substitution is a macro: a static_assert in the
    source pins the factor types, and here its numbers are checked to be what extended precision gives -- NOT bit-identical to the fp64
    checker (that would mean the substitution silently did nothing), and not further from it than fp64 round-off."""
    import os
    import synth
    ld = os.path.join(helpers.ROOT, "oracle", "libobvi_oracle_ld.so")
    if not os.path.exists(ld):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(helpers.ROOT, "oracle"), "arbiter"])
    prob = synth.make_problem(P=40, L=600, O=3, seed=7, min_obj_obs=6, object_classes=("bench",), bbox_noise=5.0)
    o, a = helpers.oracle_ba(), obvi_ba.BundleAdjuster(library=ld, prefix="oracle_")
    out = []
    for ba in (o, a):
        synth.upload(ba, prob)
        cost, res = ba.evaluate(True, True)[:2]
        s = ba.solve(helpers.ba_params(max_it=2))
        out.append((cost, np.asarray(res), s.final_cost, [i.step_norm for i in ba.iterations()]))
    (c0, r0, f0, n0), (c1, r1, f1, n1) = out
    assert abs(c0 - c1) <= 1e-12 * c1 and abs(f0 - f1) <= 1e-9 * f1
    # residuals leave the arbiter rounded to fp64: a good share of them sits one ulp away from the checker's (different intermediate rounding)
    differ = np.count_nonzero(r0 != r1)
    assert differ > 0.01 * len(r0), differ
    assert np.abs(r0 - r1).max() <= 1e-11 * max(1.0, np.abs(r1).max())
    assert n0[1] != n1[1] and abs(n0[1] - n1[1]) <= 1e-8 * n1[1]
