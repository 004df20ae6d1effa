"""ctypes binding of the C ABI declared in include/obvi_ba.h.

Plumbing only: tests and bench.py drive libobvi_ba.so (the HIP product) through this class.
The same class can bind any library exporting the ABI under another symbol prefix -- the
tests use that to drive the CPU oracle (oracle/libobvi_oracle.so, prefix "oracle_") with the
very same arrays.  Nothing here computes anything; a missing library raises.
"""
import ctypes as C
import os

import numpy as np

FACTOR_REPROJECTION = 0
FACTOR_BBOX = 2
FACTOR_SHAPE_PRIOR = 3
FACTOR_LTM_PRIOR = 4
FACTOR_REL_POSE = 5
FACTOR_TYPES = (FACTOR_REPROJECTION, FACTOR_BBOX, FACTOR_SHAPE_PRIOR, FACTOR_LTM_PRIOR, FACTOR_REL_POSE)
RESIDUAL_DIM = {0: 2, 2: 4, 3: 3, 4: 7, 5: 6}
BLOCK_DIMS = {0: (6, 3), 2: (7, 6), 3: (7, 0), 4: (7, 0), 5: (6, 6)}

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2


class Options(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("object_block_size", C.c_int32), ("reprojection_variant", C.c_int32), ("deterministic", C.c_int32), ("reserved", C.c_int32 * 4)]


class SolverParams(C.Structure):
    """pose_graph_optimization::OptimizationSolverParams (optimization_solver_params.h:10-30)."""
    _fields_ = [("max_num_iterations", C.c_int32), ("allow_non_monotonic_steps", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double)]

    def __init__(self, max_num_iterations=100, allow_non_monotonic_steps=False, function_tolerance=1e-6,
                 gradient_tolerance=1e-10, parameter_tolerance=1e-8, initial_trust_region_radius=1e4,
                 max_trust_region_radius=1e16):
        super().__init__(max_num_iterations, int(allow_non_monotonic_steps), function_tolerance,
                         gradient_tolerance, parameter_tolerance, initial_trust_region_radius,
                         max_trust_region_radius)


class IterationSummary(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32),
                ("reserved", C.c_int32), ("cost", C.c_double), ("cost_change", C.c_double),
                ("gradient_max_norm", C.c_double), ("gradient_norm", C.c_double), ("step_norm", C.c_double),
                ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
                ("iteration_time_in_seconds", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("is_solution_usable", C.c_int32),
                ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("num_parameters_reduced", C.c_int32),
                ("num_residuals_reduced", C.c_int32), ("reduced_system_size", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
                ("total_time_in_seconds", C.c_double), ("linear_solver_time_in_seconds", C.c_double),
                ("jacobian_evaluation_time_in_seconds", C.c_double),
                ("residual_evaluation_time_in_seconds", C.c_double), ("message", C.c_char * 160)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)


class EpipolarParams(C.Structure):
    """obvi_epipolar_params (include/obvi_frontend.h)."""
    _fields_ = [("inlier_epipolar_err_thresh", C.c_double), ("inlier_majority_percentage", C.c_double), ("early_votes_return", C.c_int32), ("reserved", C.c_int32)]

    def __init__(self, thresh=8.0, majority=0.5, early_return=True):
        super().__init__(thresh, majority, int(early_return), 0)


class ParallaxParams(C.Structure):
    """obvi_parallax_params (include/obvi_frontend.h); defaults: config/base7a_2_fallback.json visual_feature_params."""
    _fields_ = [("min_pixel", C.c_double), ("min_transl", C.c_double), ("min_orient", C.c_double), ("enforce_pixel", C.c_int32), ("enforce_pose", C.c_int32)]

    def __init__(self, min_pixel=5.0, min_transl=0.1, min_orient=0.05, enforce_pixel=True, enforce_pose=False):
        super().__init__(min_pixel, min_transl, min_orient, int(enforce_pixel), int(enforce_pose))


class ObviError(RuntimeError):
    pass


def default_library_path():
    """csrc/libobvi_ba.so; OBVI_BA_LIBRARY names another build of the same product (A/B runs of compile-time variants: scripts/ab_env.sh)."""
    here = os.path.dirname(os.path.abspath(__file__))
    return os.environ.get("OBVI_BA_LIBRARY") or os.path.join(os.path.dirname(here), "csrc", "libobvi_ba.so")


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


class BundleAdjuster:
    """One handle == one GPU == one HIP stream (include/obvi_ba.h)."""

    def __init__(self, device_id=0, library=None, prefix="obvi_", reprojection_variant=0, deterministic=False, object_block_size=7):
        path = library or default_library_path()
        if not os.path.exists(path):
            raise ObviError("%s not found: build it with __graft_entry__.build() -- there is no CPU fallback" % path)
        self._lib = C.CDLL(path)
        self._pre = prefix
        self._h = C.c_void_p()
        self.od = int(object_block_size)                                   # parameters of an ellipsoid block: 7 (yaw only) or 9 (axis-angle)
        opt = Options(device_id, self.od, int(reprojection_variant), 1 if deterministic else 0)
        self._check(self._fn("ba_create")(C.byref(opt), C.byref(self._h)), "create")
        self._keep = []
        self._n = {t: 0 for t in FACTOR_TYPES}
        self.P = self.L = self.O = 0

    def _fn(self, name):
        f = getattr(self._lib, self._pre + name)
        f.restype = C.c_int64 if name in ("ba_num_residuals", "ba_num_factors") else C.c_int
        return f

    def _check(self, rc, what):
        if rc != 0:
            msg = ""
            try:
                fn = getattr(self._lib, self._pre + "ba_last_error")
                fn.restype = C.c_char_p
                msg = (fn(self._h) or b"").decode()
            except AttributeError:
                pass
            raise ObviError("%s%s failed: status %d %s" % (self._pre, what, rc, msg))

    def close(self):
        if self._h:
            f = getattr(self._lib, self._pre + "ba_destroy")
            f.restype = None
            f(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- upload --------------------------------------------------------------------------
    def set_cameras(self, K, ext):
        K, ext = _f64(K, (-1, 4)), _f64(ext, (-1, 7))
        self._check(self._fn("ba_set_cameras")(self._h, C.c_int32(len(K)), _ptr(K, C.c_double), _ptr(ext, C.c_double)), "set_cameras")

    def _set_blocks(self, name, vals, dim, is_const):
        vals = _f64(vals, (-1, dim))
        c = None if is_const is None else np.ascontiguousarray(is_const, dtype=np.uint8)
        self._check(self._fn(name)(self._h, C.c_int64(len(vals)), _ptr(vals, C.c_double), _ptr(c, C.c_uint8)), name)
        return len(vals)

    def set_poses(self, poses, is_const=None):
        self.P = self._set_blocks("ba_set_poses", poses, 6, is_const)

    def set_points(self, pts, is_const=None):
        self.L = self._set_blocks("ba_set_points", pts, 3, is_const)

    def set_objects(self, objs, is_const=None):
        self.O = self._set_blocks("ba_set_objects", objs, self.od, is_const)

    def set_const_flags(self, pose_const=None, point_const=None, object_const=None):
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.uint8) for x in (pose_const, point_const, object_const)]
        self._check(self._fn("ba_set_const_flags")(self._h, _ptr(a[0], C.c_uint8), _ptr(a[1], C.c_uint8), _ptr(a[2], C.c_uint8)), "set_const_flags")

    def set_reproj(self, pose_idx, point_idx, cam_idx, pixel, sigma, huber):
        pi = np.ascontiguousarray(pose_idx, dtype=np.uint32)
        li = np.ascontiguousarray(point_idx, dtype=np.uint32)
        ci = None if cam_idx is None else np.ascontiguousarray(cam_idx, dtype=np.uint16)
        px = _f64(pixel, (-1, 2))
        if np.isscalar(sigma):
            sg, sscalar = None, float(sigma)
        else:
            sg, sscalar = _f64(sigma), 0.0
        self._check(self._fn("ba_set_reproj")(self._h, C.c_int64(len(pi)), _ptr(pi, C.c_uint32), _ptr(li, C.c_uint32),
                                              _ptr(ci, C.c_uint16), _ptr(px, C.c_double), _ptr(sg, C.c_double),
                                              C.c_double(sscalar), C.c_double(huber)), "set_reproj")
        self._n[FACTOR_REPROJECTION] = len(pi)

    def set_bbox(self, obj_idx, pose_idx, cam_idx, corners, cov, huber, invalid_err):
        oi = np.ascontiguousarray(obj_idx, dtype=np.uint32)
        pi = np.ascontiguousarray(pose_idx, dtype=np.uint32)
        ci = None if cam_idx is None else np.ascontiguousarray(cam_idx, dtype=np.uint16)
        co, cv = _f64(corners, (-1, 4)), _f64(cov, (-1, 16))
        self._check(self._fn("ba_set_bbox")(self._h, C.c_int64(len(oi)), _ptr(oi, C.c_uint32), _ptr(pi, C.c_uint32),
                                            _ptr(ci, C.c_uint16), _ptr(co, C.c_double), _ptr(cv, C.c_double),
                                            C.c_double(huber), C.c_double(invalid_err)), "set_bbox")
        self._n[FACTOR_BBOX] = len(oi)

    def set_shape_priors(self, obj_idx, mean3, cov9, huber):
        oi = np.ascontiguousarray(obj_idx, dtype=np.uint32)
        m, cv = _f64(mean3, (-1, 3)), _f64(cov9, (-1, 9))
        self._check(self._fn("ba_set_shape_priors")(self._h, C.c_int64(len(oi)), _ptr(oi, C.c_uint32), _ptr(m, C.c_double),
                                                    _ptr(cv, C.c_double), C.c_double(huber)), "set_shape_priors")
        self._n[FACTOR_SHAPE_PRIOR] = len(oi)

    def set_ltm_priors(self, obj_idx, mean7, cov49, huber):
        oi = np.ascontiguousarray(obj_idx, dtype=np.uint32)
        m, cv = _f64(mean7, (-1, self.od)), _f64(cov49, (-1, self.od * self.od))
        self._check(self._fn("ba_set_ltm_priors")(self._h, C.c_int64(len(oi)), _ptr(oi, C.c_uint32), _ptr(m, C.c_double),
                                                  _ptr(cv, C.c_double), C.c_double(huber)), "set_ltm_priors")
        self._n[FACTOR_LTM_PRIOR] = len(oi)

    def set_relpose(self, idx_a, idx_b, t3, aa3, cov36, huber):
        ia = np.ascontiguousarray(idx_a, dtype=np.uint32)
        ib = np.ascontiguousarray(idx_b, dtype=np.uint32)
        t, a, cv = _f64(t3, (-1, 3)), _f64(aa3, (-1, 3)), _f64(cov36, (-1, 36))
        self._check(self._fn("ba_set_relpose")(self._h, C.c_int64(len(ia)), _ptr(ia, C.c_uint32), _ptr(ib, C.c_uint32),
                                               _ptr(t, C.c_double), _ptr(a, C.c_double), _ptr(cv, C.c_double),
                                               C.c_double(huber)), "set_relpose")
        self._n[FACTOR_REL_POSE] = len(ia)

    def set_active_mask(self, factor_type, mask):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self._fn("ba_set_active_mask")(self._h, C.c_int32(factor_type), _ptr(m, C.c_uint8)), "set_active_mask")

    # ---- evaluate / solve ----------------------------------------------------------------
    def num_factors(self, t):
        return self._n[t]

    def _residual_dim(self, t):
        return self.od if t == 4 else RESIDUAL_DIM[t]        # an LTM prior has one residual per ellipsoid parameter

    def num_residuals(self):
        return sum(self._residual_dim(t) * self._n[t] for t in FACTOR_TYPES)

    def evaluate(self, apply_loss=True, want_residuals=True):
        cost = C.c_double(0.0)
        res = np.zeros(self.num_residuals()) if want_residuals else None
        sq = np.zeros(sum(self._n.values())) if want_residuals else None
        self._check(self._fn("ba_evaluate")(self._h, C.c_int32(int(apply_loss)), C.byref(cost), _ptr(res, C.c_double),
                                            _ptr(sq, C.c_double)), "evaluate")
        return cost.value, res, sq

    def solve(self, params):
        s = Summary()
        self._check(self._fn("ba_solve")(self._h, C.byref(params), C.byref(s)), "solve")
        return s

    def iterations(self, cap=4096):
        buf = (IterationSummary * cap)()
        n = self._fn("ba_get_iterations")(self._h, buf, C.c_int32(cap))
        return [buf[i] for i in range(n)]

    def select_outliers(self, factor_type, fraction):
        mask = np.ones(self._n[factor_type], dtype=np.uint8)
        nex = C.c_int64(0)
        self._check(self._fn("ba_select_outliers")(self._h, C.c_int32(factor_type), C.c_double(fraction),
                                                   _ptr(mask, C.c_uint8), C.byref(nex)), "select_outliers")
        return mask, nex.value

    def debug_select(self, sq, active, fraction):
        """obvi_ba_debug_select: the two-phase selection rule on block norms given here (active None: all)."""
        v = np.ascontiguousarray(sq, dtype=np.float64)
        a = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        mask = np.zeros(len(v), dtype=np.uint8)
        nex = C.c_int64(0)
        self._check(self._fn("ba_debug_select")(self._h, C.c_int64(len(v)), _ptr(v, C.c_double), _ptr(a, C.c_uint8), C.c_double(fraction),
                                                _ptr(mask, C.c_uint8), C.byref(nex)), "debug_select")
        return mask, nex.value

    def object_covariances(self, obj_a, obj_b=None):
        """7x7 covariance blocks of object pairs (obj_b None: the objects' own blocks)."""
        a = np.ascontiguousarray(obj_a, dtype=np.uint32)
        b = a if obj_b is None else np.ascontiguousarray(obj_b, dtype=np.uint32)
        out = np.zeros((len(a), self.od, self.od))
        self._check(self._fn("ba_object_covariances")(self._h, C.c_int64(len(a)), _ptr(a, C.c_uint32), _ptr(b, C.c_uint32),
                                                      _ptr(out, C.c_double)), "object_covariances")
        return out

    def set_parameter_priors(self, block_kind, block_idx, param_idx, mean, std_dev):
        """ParameterPrior factors for the covariance extraction: kind 0 pose / 1 point / 2 object; empty arrays clear them."""
        k = np.ascontiguousarray(block_kind, dtype=np.uint8); b = np.ascontiguousarray(block_idx, dtype=np.uint32)
        q = np.ascontiguousarray(param_idx, dtype=np.uint8); m, sd = _f64(mean), _f64(std_dev)
        self._check(self._fn("ba_set_parameter_priors")(self._h, C.c_int64(len(k)), _ptr(k, C.c_uint8), _ptr(b, C.c_uint32), _ptr(q, C.c_uint8),
                                                        _ptr(m, C.c_double), _ptr(sd, C.c_double)), "set_parameter_priors")

    def column_sqnorms(self):
        """squared column norms of the robustified Jacobian per scalar parameter: (poses [P,6], points [L,3], objects [O,7]); -1 = not a parameter of the problem"""
        p, l, o = np.zeros((self.P, 6)), np.zeros((self.L, 3)), np.zeros((self.O, self.od))
        self._check(self._fn("ba_column_sqnorms")(self._h, _ptr(p, C.c_double), _ptr(l, C.c_double), _ptr(o, C.c_double)), "column_sqnorms")
        return p, l, o

    # ---- visual-feature front-end gating (include/obvi_frontend.h) --------------------------
    def _frontend_fn(self, name):
        f = getattr(self._lib, self._pre + name)      # obvi_frontend_* / oracle_frontend_*
        f.restype = C.c_int
        return f

    def epipolar_votes(self, K, ext, poses, cand_pose, cand_cam, cand_pixel, ref_ptr, ref_pose, ref_cam, ref_pixel, ref_frame, ref_skip, params=None):
        K, ext, poses = _f64(K, (-1, 4)), _f64(ext, (-1, 7)), _f64(poses, (-1, 6))
        cp = np.ascontiguousarray(cand_pose, dtype=np.uint32); cc = np.ascontiguousarray(cand_cam, dtype=np.uint16); cx = _f64(cand_pixel, (-1, 2))
        rp = np.ascontiguousarray(ref_ptr, dtype=np.uint64); ro = np.ascontiguousarray(ref_pose, dtype=np.uint32); rc = np.ascontiguousarray(ref_cam, dtype=np.uint16)
        rx = _f64(ref_pixel, (-1, 2)); rf = np.ascontiguousarray(ref_frame, dtype=np.uint32); rs = np.ascontiguousarray(ref_skip, dtype=np.uint8)
        prm = params or EpipolarParams()
        n = len(cp)
        votes, voters, inl = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self._check(self._frontend_fn("frontend_epipolar_votes")(self._h, C.c_int32(len(K)), _ptr(K, C.c_double), _ptr(ext, C.c_double), C.c_int64(len(poses)), _ptr(poses, C.c_double),
                                                                C.c_int64(n), _ptr(cp, C.c_uint32), _ptr(cc, C.c_uint16), _ptr(cx, C.c_double), _ptr(rp, C.c_uint64), _ptr(ro, C.c_uint32),
                                                                _ptr(rc, C.c_uint16), _ptr(rx, C.c_double), _ptr(rf, C.c_uint32), _ptr(rs, C.c_uint8), C.byref(prm),
                                                                _ptr(votes, C.c_uint32), _ptr(voters, C.c_uint32), _ptr(inl, C.c_uint8)), "frontend_epipolar_votes")
        return votes, voters, inl

    def epipolar_errors(self, K, ext, poses, cand_pose, cand_cam, cand_pixel, ref_ptr, ref_pose, ref_cam, ref_pixel):
        K, ext, poses = _f64(K, (-1, 4)), _f64(ext, (-1, 7)), _f64(poses, (-1, 6))
        cp = np.ascontiguousarray(cand_pose, dtype=np.uint32); cc = np.ascontiguousarray(cand_cam, dtype=np.uint16); cx = _f64(cand_pixel, (-1, 2))
        rp = np.ascontiguousarray(ref_ptr, dtype=np.uint64); ro = np.ascontiguousarray(ref_pose, dtype=np.uint32); rc = np.ascontiguousarray(ref_cam, dtype=np.uint16)
        rx = _f64(ref_pixel, (-1, 2))
        err = np.zeros((len(ro), 2))
        self._check(self._frontend_fn("frontend_epipolar_errors")(self._h, C.c_int32(len(K)), _ptr(K, C.c_double), _ptr(ext, C.c_double), C.c_int64(len(poses)), _ptr(poses, C.c_double),
                                                                 C.c_int64(len(cp)), _ptr(cp, C.c_uint32), _ptr(cc, C.c_uint16), _ptr(cx, C.c_double), _ptr(rp, C.c_uint64), _ptr(ro, C.c_uint32),
                                                                 _ptr(rc, C.c_uint16), _ptr(rx, C.c_double), _ptr(err, C.c_double)), "frontend_epipolar_errors")
        return err

    def parallax(self, frame_ptr, has_pose, poses, obs_ptr, pixels, params=None):
        fp = np.ascontiguousarray(frame_ptr, dtype=np.uint64); hp = np.ascontiguousarray(has_pose, dtype=np.uint8); po = _f64(poses, (-1, 6))
        op = np.ascontiguousarray(obs_ptr, dtype=np.uint64); px = _f64(pixels, (-1, 2))
        prm = params or ParallaxParams()
        out = np.zeros(len(fp) - 1, np.uint8)
        self._check(self._frontend_fn("frontend_parallax")(self._h, C.c_int64(len(fp) - 1), _ptr(fp, C.c_uint64), _ptr(hp, C.c_uint8), _ptr(po, C.c_double), _ptr(op, C.c_uint64),
                                                          _ptr(px, C.c_double), C.byref(prm), _ptr(out, C.c_uint8)), "frontend_parallax")
        return out

    # ---- state ---------------------------------------------------------------------------
    def reset(self):
        """obvi_ba_reset: the handle as create left it (no problem, no hook, nothing shared), allocations kept."""
        self._check(self._fn("ba_reset")(self._h), "reset")
        self._n = {t: 0 for t in FACTOR_TYPES}
        self.P = self.L = self.O = 0
        self._keep.clear()

    def snapshot(self):
        self._check(self._fn("ba_snapshot")(self._h), "snapshot")

    def restore(self):
        self._check(self._fn("ba_restore")(self._h), "restore")

    def _get(self, name, n, dim):
        out = np.zeros((n, dim))
        self._check(self._fn(name)(self._h, _ptr(out, C.c_double)), name)
        return out

    def get_poses(self):
        return self._get("ba_get_poses", self.P, 6)

    def get_points(self):
        return self._get("ba_get_points", self.L, 3)

    def get_objects(self):
        return self._get("ba_get_objects", self.O, self.od)

    def get_state(self):
        """(poses, points, objects) with one wait for the device (obvi_ba_get_state)."""
        po, pt, ob = np.zeros((self.P, 6)), np.zeros((self.L, 3)), np.zeros((self.O, self.od))
        self._check(self._fn("ba_get_state")(self._h, _ptr(po, C.c_double), _ptr(pt, C.c_double), _ptr(ob, C.c_double)), "get_state")
        return po, pt, ob

    def update_points(self, xyz):
        x = _f64(xyz, (-1, 3))
        self._check(self._fn("ba_update_points")(self._h, C.c_int64(len(x)), _ptr(x, C.c_double)), "update_points")

    def update_state(self, poses=None, points=None, objects=None):
        """Values only (obvi_ba_update_state): constness, factors and the symbolic plan stay."""
        po = None if poses is None else _f64(poses, (self.P, 6))
        pt = None if points is None else _f64(points, (self.L, 3))
        ob = None if objects is None else _f64(objects, (self.O, self.od))
        self._check(self._fn("ba_update_state")(self._h, _ptr(po, C.c_double), _ptr(pt, C.c_double), _ptr(ob, C.c_double)), "update_state")

    def prepare(self):
        """The symbolic phase now (obvi_ba_prepare) instead of inside the first solve / evaluate."""
        self._check(self._fn("ba_prepare")(self._h), "prepare")

    # ---- multi-GPU / test hooks ----------------------------------------------------------
    def set_allreduce(self, pyfunc):
        """pyfunc(device_ptr:int, count_f64:int, op:int (0 sum, 1 max), stream:int) -> int (0 = ok)."""
        if pyfunc is None:
            cb = C.cast(None, ALLREDUCE_FN)
        else:
            cb = ALLREDUCE_FN(lambda user, buf, count, op, stream: int(pyfunc(buf or 0, count, op, stream or 0)))
        self._keep.append(cb)
        self._check(self._fn("ba_set_allreduce")(self._h, cb, None), "set_allreduce")

    def set_shared_objects(self, is_shared, rank, world):
        m = None if is_shared is None else np.ascontiguousarray(is_shared, dtype=np.uint8)
        self._check(self._fn("ba_set_shared_objects")(self._h, _ptr(m, C.c_uint8), C.c_int32(rank), C.c_int32(world)), "set_shared_objects")

    def measure_peaks(self):
        """HBM triad / copy / read bandwidth (GB/s) and the fp64 MFMA rates (TFLOP/s) of this device (obvi_ba_measure_peaks)."""
        class Peaks(C.Structure):
            _fields_ = [("hbm_triad_gbs", C.c_double), ("hbm_copy_gbs", C.c_double), ("hbm_read_gbs", C.c_double), ("mfma_f64_issue_tflops", C.c_double),
                        ("mfma_f64_tile_tflops", C.c_double), ("clock_mhz", C.c_double), ("compute_units", C.c_int32), ("reserved", C.c_int32)]
        p = Peaks()
        self._check(self._fn("ba_measure_peaks")(self._h, C.byref(p)), "measure_peaks")
        return {name: getattr(p, name) for name, _ in Peaks._fields_ if name != "reserved"}

    def debug_linearize(self, factor_type):
        n, m = self._n[factor_type], self._residual_dim(factor_type)
        d0, d1 = (self.od if d == 7 else d for d in BLOCK_DIMS[factor_type])
        r, J0 = np.zeros((n, m)), np.zeros((n, m, d0))
        J1 = np.zeros((n, m, d1)) if d1 else None
        self._check(self._fn("ba_debug_linearize")(self._h, C.c_int32(factor_type), _ptr(r, C.c_double),
                                                   _ptr(J0, C.c_double), _ptr(J1, C.c_double)), "debug_linearize")
        return r, J0, J1

    def debug_reduced_system(self, radius, m_cap=4096):
        lhs, rhs = np.zeros((m_cap, m_cap)), np.zeros(m_cap)
        m = C.c_int32(0)
        self._check(self._fn("ba_debug_reduced_system")(self._h, C.c_double(radius), _ptr(lhs, C.c_double),
                                                        _ptr(rhs, C.c_double), C.c_int32(m_cap), C.byref(m)), "debug_reduced_system")
        mm = m.value
        return lhs.ravel()[:mm * mm].reshape(mm, mm).copy(), rhs[:mm].copy()

    def problem_stats(self):
        buf = (C.c_double * 18)()
        n = self._fn("ba_get_problem_stats")(self._h, buf, C.c_int32(18))
        names = ["poses_var", "objects_var", "points_var", "reduced_rows", "tiles_per_dim", "schur_blocks", "schur_pairs",
                 "tiles_nonzero", "trsm_jobs", "update_jobs", "chol_flops", "reproj_active", "bbox_active", "chol_levels", "host_threads", "usable_cpus",
                 "potrf_wait_timeouts", "fused_potrf"]
        return {names[i]: buf[i] for i in range(n)}

    def set_profiling(self, level):
        self._check(self._fn("ba_set_profiling")(self._h, C.c_int32(level)), "set_profiling")

    def kernel_times(self, cap=64):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = self._fn("ba_get_kernel_times")(self._h, names, C.c_int32(4096), ms, cnt, C.c_int32(cap))
        parts = names.raw.split(b"\0")[:n]
        return {parts[i].decode(): (ms[i], cnt[i]) for i in range(n)}
