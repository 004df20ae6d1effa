#!/usr/bin/env python
"""Who is right when HIP and the oracle disagree?  (VERDICT r3 "next" 2(ii))

Three runs of the same LM step from the same fp64 state:
    HIP        libobvi_ba.so (default handle: fp64 atomics; and the deterministic handle)
    oracle     oracle/libobvi_oracle.so     -- scalar fp64 (the checker)
    arbiter    oracle/libobvi_oracle_ld.so  -- the same source with every solver-level sum (J^T J, Schur complement, factorisation,
               substitutions, model cost change, cost sums) in x87 extended precision (64 mantissa bits)
and the distances |HIP - arbiter| and |oracle - arbiter| of what an LM decision is taken from: the cost after the step, the step
norm, the relative decrease.  Both fp64 runs sit 1e3 x their own round-off away from the arbiter's, so the arbiter's own error does
not matter for the comparison.

    part A   K states along the ARBITER's trajectory of the bench workload (config 3); from each state every backend takes ONE LM step
             with the arbiter's trust-region radius (a fresh solve: Jacobi scaling of that state) -- the one-step error, four times
    part B   every backend runs K steps freely from the start -- what the amplification of part A's error does to a trajectory
Prints a table and writes gpurun_out/arbiter_<tag>.json.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python"))
import numpy as np  # noqa: E402
import obvi_ba  # noqa: E402
import synth  # noqa: E402


def params(iters, radius):
    return obvi_ba.SolverParams(max_num_iterations=iters, allow_non_monotonic_steps=True, function_tolerance=0.0, gradient_tolerance=0.0,
                                parameter_tolerance=0.0, initial_trust_region_radius=radius, max_trust_region_radius=1e4)


def set_state(ba, prob, st):
    q = dict(prob)
    q.update(poses=st[0], points=st[1], objects=st[2])
    synth.upload(ba, q)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=2000); ap.add_argument("--L", type=int, default=300000); ap.add_argument("--O", type=int, default=200)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--threads", type=int, default=20)
    ap.add_argument("--tag", default="cfg3")
    args = ap.parse_args()
    prob = synth.make_problem(P=args.P, L=args.L, O=args.O, seed=20241008 + 3, const_poses=1, min_obj_obs=10)
    libs = {"oracle": os.path.join(ROOT, "oracle", "libobvi_oracle.so"), "arbiter": os.path.join(ROOT, "oracle", "libobvi_oracle_ld.so")}
    threads = max(1, min(args.threads, os.cpu_count() or 1))
    for lib in libs.values():
        ctypes.CDLL(lib).oracle_set_threads(ctypes.c_int32(threads))
    make = {"hip": lambda: obvi_ba.BundleAdjuster(device_id=0), "hip_det": lambda: obvi_ba.BundleAdjuster(device_id=0, deterministic=True),
            "oracle": lambda: obvi_ba.BundleAdjuster(library=libs["oracle"], prefix="oracle_"), "arbiter": lambda: obvi_ba.BundleAdjuster(library=libs["arbiter"], prefix="oracle_")}
    names = ["hip", "hip_det", "oracle", "arbiter"]
    bas = {n: make[n]() for n in names}
    out = {"problem": {"P": args.P, "L": args.L, "O": args.O, "host_threads": threads}, "one_step": [], "free_running": {}}

    # ---- part A
    state = (prob["poses"].copy(), prob["points"].copy(), prob["objects"].copy())
    radius = 100.0
    for k in range(args.steps):
        rec = {}
        for n in names:
            set_state(bas[n], prob, state)
            t0 = time.time()
            bas[n].solve(params(1, radius))
            it = bas[n].iterations()
            rec[n] = dict(initial_cost=it[0].cost, cost=it[1].cost, step_norm=it[1].step_norm, relative_decrease=it[1].relative_decrease,
                          accepted=int(it[1].step_is_successful), valid=int(it[1].step_is_valid), radius_after=it[1].trust_region_radius, seconds=time.time() - t0)
        a = rec["arbiter"]
        row = {"state": k, "radius": radius, "arbiter": a}
        for n in names[:-1]:
            r = rec[n]
            row[n] = dict(initial_cost_rel=rel(r["initial_cost"], a["initial_cost"]), cost_rel=rel(r["cost"], a["cost"]), step_norm_rel=rel(r["step_norm"], a["step_norm"]),
                          relative_decrease_abs=abs(r["relative_decrease"] - a["relative_decrease"]), same_decision=bool(r["accepted"] == a["accepted"] and r["valid"] == a["valid"]))
        out["one_step"].append(row)
        print("state %d (radius %.4g, arbiter: cost %.9g -> %.9g, rho %.4f, %s)" % (k, radius, a["initial_cost"], a["cost"], a["relative_decrease"], "accepted" if a["accepted"] else "rejected"))
        for n in names[:-1]:
            print("   %-8s |cost - arb| / cost %.2e   |step| %.2e   rho %.2e   decision %s" % (n, row[n]["cost_rel"], row[n]["step_norm_rel"], row[n]["relative_decrease_abs"], "same" if row[n]["same_decision"] else "DIFFERENT"))
        sys.stdout.flush()
        state = (bas["arbiter"].get_poses(), bas["arbiter"].get_points(), bas["arbiter"].get_objects())
        radius = a["radius_after"]

    # ---- part B
    start = (prob["poses"].copy(), prob["points"].copy(), prob["objects"].copy())
    traj = {}
    for n in names:
        set_state(bas[n], prob, start)
        bas[n].solve(params(args.steps, 100.0))
        traj[n] = [dict(cost=i.cost, accepted=int(i.step_is_successful), relative_decrease=i.relative_decrease) for i in bas[n].iterations()]
    for n in names[:-1]:
        out["free_running"][n] = [dict(step=k, cost_rel=rel(traj[n][k]["cost"], traj["arbiter"][k]["cost"]), same_decision=traj[n][k]["accepted"] == traj["arbiter"][k]["accepted"])
                                  for k in range(min(len(traj[n]), len(traj["arbiter"])))]
    out["free_running"]["arbiter_costs"] = [t["cost"] for t in traj["arbiter"]]
    print("free-running, |cost - arbiter| / cost after step k:")
    for n in names[:-1]:
        print("   %-8s " % n + "  ".join("%.2e" % r["cost_rel"] for r in out["free_running"][n]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "arbiter_%s.json" % args.tag), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
