"""Oracle vs HIP LM trajectory on a mid-size problem (dev check):  python scripts/traj_check.py [P L O iters]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "obvi-slam_amd", "python")); sys.path.insert(0, os.path.join(R, "tests"))
import synth, helpers
P, L, O, iters = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (400, 40000, 40, 8)))
prob = synth.make_problem(P=P, L=L, O=O, seed=20241008, const_poses=1, min_obj_obs=10)
o, g = helpers.oracle_ba(), helpers.product_ba()
for ba in (o, g): synth.upload(ba, prob)
prm = helpers.ba_params(max_it=iters, ftol=0, ptol=0, gtol=0)
so, sg = o.solve(prm), g.solve(prm)
for a, b in zip(o.iterations(), g.iterations()):
    print("it %2d  cost %.10e / %.10e  rel %.1e   rho %.6e / %.6e   radius %.4e / %.4e  ok %d/%d" % (a.iteration, a.cost, b.cost, abs(a.cost - b.cost) / a.cost, a.relative_decrease, b.relative_decrease, a.trust_region_radius, b.trust_region_radius, a.step_is_successful, b.step_is_successful))
print("final", so.final_cost, sg.final_cost, "max pose diff", np.abs(g.get_poses() - o.get_poses()).max())
