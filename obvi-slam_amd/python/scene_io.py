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


def write_scene_binary(prob, path):
    """The same scene as raw arrays (run_offline_ba.cpp: loadSceneBinary): a 3 M-sighting scene in a fraction of a second."""
    K, ext = prob["K"], prob["ext"]
    n_obj = len(prob["objects"])
    first_seen = np.full(n_obj, np.iinfo(np.int64).max)
    np.minimum.at(first_seen, prob["bb_obj"].astype(np.int64), prob["bb_pose"].astype(np.int64))
    order = np.lexsort((np.arange(n_obj), first_seen))
    seen = order[first_seen[order] < np.iinfo(np.int64).max]
    new_id = -np.ones(n_obj, dtype=np.int64)
    new_id[seen] = np.arange(len(seen))
    names = list(synth.SHAPE_CLASSES)
    with open(path, "wb") as f:
        def count(n):
            f.write(np.uint64(n).tobytes())
        f.write(b"OBVISCN1")
        count(len(K))
        for c in range(len(K)):
            aa = Rot.from_quat(ext[c, 0:4] / np.linalg.norm(ext[c, 0:4])).as_rotvec()
            f.write(np.concatenate([[float(c)], K[c], ext[c, 4:7], aa]).astype("<f8").tobytes())
        count(len(prob["poses"])); f.write(np.ascontiguousarray(prob["poses"], dtype="<f8").tobytes())
        count(len(prob["points"])); f.write(np.ascontiguousarray(prob["points"], dtype="<f8").tobytes())
        n = len(prob["rp_pose"])
        count(n)
        f.write(np.stack([prob["rp_pose"], prob["rp_point"], prob["rp_cam"]], axis=1).astype("<i8").tobytes())
        f.write(np.ascontiguousarray(prob["rp_pixel"], dtype="<f8").tobytes())
        count(len(seen))
        for o in seen:
            f.write(np.concatenate([[float(new_id[o]), float(names.index(prob["obj_class"][o]))], prob["objects"][o]]).astype("<f8").tobytes())
        keep = np.flatnonzero(new_id[prob["bb_obj"].astype(np.int64)] >= 0)
        count(len(keep))
        rows = np.concatenate([prob["bb_pose"][keep, None].astype(np.float64), new_id[prob["bb_obj"][keep].astype(np.int64), None].astype(np.float64), prob["bb_cam"][keep, None].astype(np.float64),
                               prob["bb_corners"][keep], prob["bb_cov"][keep, 0:1]], axis=1)
        f.write(np.ascontiguousarray(rows, dtype="<f8").tobytes())
        count(len(names))
        for name in names:
            mean, sd = synth.SHAPE_CLASSES[name]
            f.write(name.encode()[:31].ljust(32, b"\0"))
            f.write(np.concatenate([mean, sd]).astype("<f8").tobytes())
    return {int(o): int(new_id[o]) for o in seen}
