import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'obvi-slam_amd/python')
import numpy as np
import obvi_ba, synth, helpers
from helpers import rel_err
P, L, O = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
prob = synth.make_problem(P=P, L=L, O=O, seed=3, bbox_noise=5.0)
print(synth.problem_stats(prob))
o = helpers.oracle_ba(); g = helpers.product_ba()
for ba in (o, g): synth.upload(ba, prob)
if 6*P+7*O < 3000:
    So, bo = o.debug_reduced_system(100.0); Sg, bg = g.debug_reduced_system(100.0)
    print("S", So.shape, Sg.shape, "err", rel_err(Sg, So), "rhs err", rel_err(bg, bo))
prm = helpers.ba_params(max_it=iters, ftol=0, ptol=0, gtol=0)
t = time.time(); so = o.solve(prm); t1 = time.time() - t
g.set_profiling(1)
t = time.time(); sg = g.solve(prm); t2 = time.time() - t
print("oracle", so.termination_type, so.message, so.num_iterations, so.initial_cost, so.final_cost, "%.3fs" % t1)
print("gpu   ", sg.termination_type, sg.message, sg.num_iterations, sg.initial_cost, sg.final_cost, "%.3fs" % t2)
for a, b in zip(o.iterations(), g.iterations()):
    print(a.iteration, "%.12e %.12e" % (a.cost, b.cost), "%.3e %.3e" % (a.gradient_max_norm, b.gradient_max_norm), "%.6e %.6e" % (a.step_norm, b.step_norm), "%.5f %.5f" % (a.relative_decrease, b.relative_decrease), a.step_is_successful, b.step_is_successful)
print("final pose err", np.abs(g.get_poses()-o.get_poses()).max(), "points", np.abs(g.get_points()-o.get_points()).max(), "objects", np.abs(g.get_objects()-o.get_objects()).max() if O else 0)
print(g.problem_stats())
print({k: (round(v[0]/max(v[1],1),4), v[1]) for k, v in g.kernel_times().items()})
