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


def oracle_ba():
    return obvi_ba.BundleAdjuster(library=ensure_oracle(), prefix="oracle_")


def product_ba(device=0):
    return obvi_ba.BundleAdjuster(device_id=device, library=PRODUCT_LIB, prefix="obvi_")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def ba_params(max_it=50, nonmono=True, ftol=1e-6, radius=100.0, max_radius=1e4, gtol=1e-10, ptol=1e-8):
    """global_ba / local_ba style parameter block."""
    return obvi_ba.SolverParams(max_num_iterations=max_it, allow_non_monotonic_steps=nonmono, function_tolerance=ftol,
                                gradient_tolerance=gtol, parameter_tolerance=ptol, initial_trust_region_radius=radius,
                                max_trust_region_radius=max_radius)
