import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'obvi-slam_amd/python')
import numpy as np
import obvi_ba, synth, helpers
from helpers import rel_err
prob = synth.make_problem(P=30, L=400, O=3, seed=1, object_classes=("bench",), bbox_noise=5.0, min_obj_obs=5)
print(synth.problem_stats(prob))
o = helpers.oracle_ba(); g = helpers.product_ba()
for ba in (o, g): synth.upload(ba, prob)
co = o.evaluate(True); cg = g.evaluate(True)
print("cost", co[0], cg[0], "res err", rel_err(cg[1], co[1]), "sq err", rel_err(cg[2], co[2]))
co = o.evaluate(False); cg = g.evaluate(False)
print("raw cost", co[0], cg[0], "res err", rel_err(cg[1], co[1]))
for t in (0, 2, 3, 5):
    ro, J0o, J1o = o.debug_linearize(t); rg, J0g, J1g = g.debug_linearize(t)
    print("type", t, "r", rel_err(rg, ro), "J0", rel_err(J0g, J0o), "J1", None if J1o is None else rel_err(J1g, J1o))
So, bo = o.debug_reduced_system(100.0); Sg, bg = g.debug_reduced_system(100.0)
print("S", So.shape, Sg.shape, "err", rel_err(Sg, So), "rhs err", rel_err(bg, bo))
prm = helpers.ba_params(max_it=1, ftol=0, ptol=0, gtol=0)
so = o.solve(prm); sg = g.solve(prm)
print("1 step: cost", so.final_cost, sg.final_cost, "pose err", rel_err(g.get_poses(), o.get_poses()), "pt err", rel_err(g.get_points(), o.get_points()), "obj err", rel_err(g.get_objects(), o.get_objects()))
for ba in (o, g): synth.upload(ba, prob)
prm = helpers.ba_params(max_it=40)
t = time.time(); so = o.solve(prm); t1 = time.time() - t
t = time.time(); sg = g.solve(prm); t2 = time.time() - t
print("oracle", so.termination_type, so.message, so.num_iterations, so.initial_cost, so.final_cost, "%.3fs" % t1)
print("gpu   ", sg.termination_type, sg.message, sg.num_iterations, sg.initial_cost, sg.final_cost, "%.3fs" % t2)
io, ig = o.iterations(), g.iterations()
for a, b in zip(io, ig):
    print(a.iteration, "%.10e %.10e" % (a.cost, b.cost), "%.3e %.3e" % (a.gradient_max_norm, b.gradient_max_norm), "%.3e %.3e" % (a.step_norm, b.step_norm), "%.4f %.4f" % (a.relative_decrease, b.relative_decrease), a.step_is_successful, b.step_is_successful)
print("final pose err", rel_err(g.get_poses(), o.get_poses()), "points", rel_err(g.get_points(), o.get_points()), "objects", rel_err(g.get_objects(), o.get_objects()))
print(g.kernel_times())
