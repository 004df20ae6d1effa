"""Which stage of a deterministic solve is not bit-reproducible?  (GPU; development probe.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "obvi-slam_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, synth
from test_gpu_deterministic import all_families

P, L, O, seed = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (40, 500, 4, 3)))
prob = all_families(P, L, O, seed)
def one(max_it):
    ba = helpers.product_ba(deterministic=True)
    synth.upload(ba, prob)
    S, b = ba.debug_reduced_system(100.0)
    r = [ba.debug_linearize(t) for t in (0, 2, 3, 4, 5)]
    s = ba.solve(helpers.ba_params(max_it=max_it, ftol=0, gtol=0, ptol=0))
    it = ba.iterations()
    return dict(S=S, b=b, poses=ba.get_poses(), points=ba.get_points(), objects=ba.get_objects(),
                its=[(i.cost, i.gradient_max_norm, i.gradient_norm, i.step_norm, i.relative_decrease) for i in it])
for max_it in (1, 2):
    a, c = one(max_it), one(max_it)
    for k in ("S", "b", "poses", "points", "objects"):
        d = np.abs(a[k] - c[k])
        print("max_it", max_it, k, "equal" if np.array_equal(a[k], c[k]) else "DIFFER max %.3e at %s of %d" % (d.max(), np.argwhere(d > 0)[:4].tolist(), (d > 0).sum()))
    for x, y in zip(a["its"], c["its"]):
        print("   ", ["=" if p == q else "%.17g|%.17g" % (p, q) for p, q in zip(x, y)])
