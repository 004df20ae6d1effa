"""Writes a synthetic problem (synth.make_problem) as the text scene the C++ host driver reads
(obvi-slam_amd/host/run_offline_ba.cpp: loadScene).  Objects are renumbered in first-observation order,
the order in which the frame data adder creates them."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot

import synth


def write_scene(prob, path):
    K, ext = prob["K"], prob["ext"]
    n_obj = len(prob["objects"])
    first_seen = np.full(n_obj, np.iinfo(np.int64).max)
    np.minimum.at(first_seen, prob["bb_obj"].astype(np.int64), prob["bb_pose"].astype(np.int64))
    order = np.lexsort((np.arange(n_obj), first_seen))      # old ids sorted by first observation
    seen = order[first_seen[order] < np.iinfo(np.int64).max]
    new_id = {int(o): i for i, o in enumerate(seen)}
    with open(path, "w") as f:
        f.write("obvi_scene 1\ncameras %d\n" % len(K))
        for c in range(len(K)):
            aa = Rot.from_quat(ext[c, 0:4] / np.linalg.norm(ext[c, 0:4])).as_rotvec()
            f.write("%d %s\n" % (c, " ".join(repr(float(v)) for v in list(K[c]) + list(ext[c, 4:7]) + list(aa))))
        f.write("frames %d\n" % len(prob["poses"]))
        for p in prob["poses"]:
            f.write(" ".join(repr(float(v)) for v in p) + "\n")
        f.write("features %d\n" % len(prob["points"]))
        for i, p in enumerate(prob["points"]):
            f.write("%d %s\n" % (i, " ".join(repr(float(v)) for v in p)))
        f.write("visual_obs %d\n" % len(prob["rp_pose"]))
        for k in range(len(prob["rp_pose"])):
            f.write("%d %d %d %r %r\n" % (prob["rp_pose"][k], prob["rp_point"][k], prob["rp_cam"][k], float(prob["rp_pixel"][k, 0]), float(prob["rp_pixel"][k, 1])))
        f.write("objects %d\n" % len(seen))
        for o in seen:
            f.write("%d %s %s\n" % (new_id[int(o)], prob["obj_class"][o], " ".join(repr(float(v)) for v in prob["objects"][o])))
        keep = [k for k in range(len(prob["bb_obj"])) if int(prob["bb_obj"][k]) in new_id]
        f.write("box_obs %d\n" % len(keep))
        for k in keep:
            f.write("%d %d %d %s %r\n" % (prob["bb_pose"][k], new_id[int(prob["bb_obj"][k])], prob["bb_cam"][k],
                                           " ".join(repr(float(v)) for v in prob["bb_corners"][k]), float(prob["bb_cov"][k, 0])))
        f.write("classes %d\n" % len(synth.SHAPE_CLASSES))
        for name, (mean, sd) in synth.SHAPE_CLASSES.items():
            f.write("%s %s\n" % (name, " ".join(repr(float(v)) for v in list(mean) + list(sd))))
    return new_id
