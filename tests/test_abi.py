"""CPU: the C-ABI library builds for gfx950, loads, exports every declared symbol and refuses to run without a GPU."""
import ctypes as C
import os

import pytest

import helpers
import obvi_ba


def _abi_symbols():
    import __graft_entry__ as ge
    return ge.abi_symbols()


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(helpers.PRODUCT_LIB):
        import __graft_entry__ as ge
        ge.build()
    return C.CDLL(helpers.PRODUCT_LIB)


def test_exports_every_declared_symbol(lib):
    syms = _abi_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_frontend_and_rccl_libraries_export_their_headers(lib):
    import __graft_entry__ as ge
    front = ge.abi_symbols("obvi_frontend.h", "obvi_frontend_")
    assert set(front) == {"obvi_frontend_epipolar_votes", "obvi_frontend_epipolar_errors", "obvi_frontend_parallax"}
    assert not [s for s in front if not hasattr(lib, s)]
    o = C.CDLL(helpers.ensure_oracle())
    assert all(hasattr(o, s.replace("obvi_", "oracle_", 1)) for s in front)
    rccl_path = os.path.join(os.path.dirname(helpers.PRODUCT_LIB), "libobvi_rccl.so")
    syms = ge.abi_symbols("obvi_rccl.h", "obvi_rccl_")
    assert len(syms) >= 10 and "obvi_rccl_allreduce" in syms
    rccl = C.CDLL(rccl_path)                       # links librccl: loads without a GPU; communicators need one
    assert not [s for s in syms if not hasattr(rccl, s)]
    # libobvi_ba.so itself must not depend on RCCL (the collective is a callback)
    import subprocess
    needed = subprocess.run(["readelf", "-d", helpers.PRODUCT_LIB], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()


def test_oracle_mirrors_the_abi():
    o = C.CDLL(helpers.ensure_oracle())
    skip = {"obvi_ba_last_error", "obvi_ba_version", "obvi_ba_get_kernel_times", "obvi_ba_get_problem_stats", "obvi_ba_set_profiling", "obvi_ba_measure_peaks"}   # device-only hooks
    for s in _abi_symbols():
        if s not in skip:
            assert hasattr(o, s.replace("obvi_", "oracle_", 1)), s


def test_version_and_struct_sizes(lib):
    lib.obvi_ba_version.restype = C.c_char_p
    assert b"gfx950" in lib.obvi_ba_version()
    assert C.sizeof(obvi_ba.SolverParams) == 48 and C.sizeof(obvi_ba.IterationSummary) == 80 and C.sizeof(obvi_ba.Summary) == 248


def test_no_cpu_fallback(lib):
    """Without a HIP device the product refuses to create a handle (OBVI_ERR_NO_DEVICE); it never computes on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    opt = obvi_ba.Options(0, 7)
    assert lib.obvi_ba_create(C.byref(opt), C.byref(h)) == -2
    with pytest.raises(obvi_ba.ObviError):
        obvi_ba.BundleAdjuster(device_id=0)
    with pytest.raises(obvi_ba.ObviError):
        obvi_ba.BundleAdjuster(library="/nonexistent/libobvi_ba.so")


def test_invalid_arguments(lib):
    assert lib.obvi_ba_create(None, None) == -1
    bad = obvi_ba.Options(0, 8)     # an ellipsoid block has 7 (yaw only) or 9 (axis-angle) parameters
    h = C.c_void_p()
    assert lib.obvi_ba_create(C.byref(bad), C.byref(h)) == -1
    assert lib.obvi_ba_solve(None, None, None) == -1
    assert lib.obvi_ba_object_covariances(None, C.c_int64(0), None, None, None) == -1


def test_generator_shapes():
    import numpy as np
    import synth
    prob = synth.make_problem(P=40, L=500, O=3, seed=3, min_obj_obs=5)
    st = synth.problem_stats(prob)
    assert st["P"] == 40 and st["L"] == 500 and st["O"] == 3 and st["N_rel"] == 39
    cnt = np.bincount(prob["rp_point"], minlength=500)
    assert cnt.min() >= 5                                  # min_low_level_feature_observations
    assert np.all(np.diff(prob["rp_point"].astype(np.int64)) >= 0)
    assert np.bincount(prob["bb_obj"], minlength=3).min() >= 5
    px, z = synth.project_points(prob["gt_poses"][prob["rp_pose"]], prob["gt_points"][prob["rp_point"]])
    inl = ~prob["rp_is_outlier"]
    assert np.abs(px - prob["rp_pixel"])[inl].std() < 1.2 and z.min() > 0.5


def test_host_pool_hands_every_part_out_exactly_once(tmp_path):
    """csrc/host_util.h HostPool (the worker threads of the symbolic phase): 40 000 back-to-back runs of changing part counts from two
    caller threads -- every part exactly once, nothing after run() has returned (a part index could cross from one run to the next)."""
    import subprocess
    exe = str(tmp_path / "hostpool_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, os.path.join(helpers.ROOT, "tests", "hostpool_stress.cpp")])
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
