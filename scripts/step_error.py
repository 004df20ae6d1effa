#!/usr/bin/env python
"""Where does the round-off of ONE LM step come from?  Window-sized problem, three backends (HIP, fp64 oracle, extended-precision arbiter):
   assembly   the reduced system S, b each backend forms at the same point (debug_reduced_system) against the arbiter's
   solve      each backend's step y (pose / object part, read off an accepted one-iteration solve) against (a) the refined solution of ITS OWN
              system (factorisation + substitutions alone) and (b) the refined solution of the arbiter's system (everything)
Refinement: fp64 solve + residuals in numpy longdouble, four rounds.  Prints a table per radius."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python"))
import numpy as np  # noqa: E402
import obvi_ba  # noqa: E402
import synth  # noqa: E402


def refined_solve(S, b, rounds=4):
    Sl, bl = S.astype(np.longdouble), b.astype(np.longdouble)
    y = np.linalg.solve(S, b).astype(np.longdouble)
    for _ in range(rounds):
        r = bl - Sl @ y
        y = y + np.linalg.solve(S, r.astype(np.float64)).astype(np.longdouble)
    return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=50); ap.add_argument("--L", type=int, default=6000); ap.add_argument("--O", type=int, default=8)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--warm", type=int, default=6, help="LM iterations (oracle) before the measurement: the regime of a converging window")
    args = ap.parse_args()
    prob = synth.make_problem(P=args.P, L=args.L, O=args.O, seed=args.seed, const_poses=1, min_obj_obs=6, bbox_noise=5.0)
    libs = {"oracle": os.path.join(ROOT, "oracle", "libobvi_oracle.so"), "arbiter": os.path.join(ROOT, "oracle", "libobvi_oracle_ld.so")}
    mk = {"hip": lambda: obvi_ba.BundleAdjuster(device_id=0, deterministic=True), "oracle": lambda: obvi_ba.BundleAdjuster(library=libs["oracle"], prefix="oracle_"),
          "arbiter": lambda: obvi_ba.BundleAdjuster(library=libs["arbiter"], prefix="oracle_")}
    bas = {n: f() for n, f in mk.items()}
    # a state part of the way down (the arbiter's): where windows spend their iterations
    synth.upload(bas["arbiter"], prob)
    if args.warm:
        bas["arbiter"].solve(obvi_ba.SolverParams(max_num_iterations=args.warm, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0,
                                                  initial_trust_region_radius=100.0, max_trust_region_radius=1e4))
    state = dict(prob)
    state.update(poses=bas["arbiter"].get_poses(), points=bas["arbiter"].get_points(), objects=bas["arbiter"].get_objects())
    pv = np.flatnonzero(prob["pose_const"] == 0)
    for radius in (1e2, 1e3, 1e4):
        sysm, step = {}, {}
        for n, ba in bas.items():
            synth.upload(ba, state)
            S, b = ba.debug_reduced_system(radius)
            sysm[n] = (S, b)
            p0, o0 = ba.get_poses(), ba.get_objects()
            ba.solve(obvi_ba.SolverParams(max_num_iterations=1, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0,
                                          initial_trust_region_radius=radius, max_trust_region_radius=1e16))
            it = ba.iterations()[1]
            d = np.concatenate([(ba.get_poses() - p0)[pv].ravel(), (ba.get_objects() - o0).ravel()])
            step[n] = (-d if it.step_is_successful else None, it.cost, it.relative_decrease)
        Sa, ba_ = sysm["arbiter"]
        dscale = 1.0 / np.sqrt(np.diag(Sa))
        y_arb = refined_solve(Sa, ba_)
        print("radius %g: m = %d, cond(S) = %.1e, arbiter step accepted: %s" % (radius, len(ba_), np.linalg.cond(Sa), step["arbiter"][0] is not None))
        for n in ("hip", "oracle", "arbiter"):
            S, b = sysm[n]
            eS = np.abs((S - Sa) * dscale[:, None] * dscale[None, :]).max()
            eb = np.abs(b - ba_).max() / np.abs(ba_).max()
            line = "   %-8s assembly: |S - S_arb| (Jacobi-scaled) %.1e, |b - b_arb| %.1e" % (n, eS, eb)
            if step[n][0] is not None and len(step[n][0]) == len(b):
                y = step[n][0].astype(np.longdouble)
                own = refined_solve(S, b)
                line += " | step vs refined solve of its own system %.1e, vs the arbiter's system %.1e" % (float(np.abs(y - own).max() / np.abs(own).max()), float(np.abs(y - y_arb).max() / np.abs(y_arb).max()))
                line += " | cost after %.17g rho %.12f" % (step[n][1], step[n][2])
            print(line)


if __name__ == "__main__":
    main()
