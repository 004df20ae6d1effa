#!/usr/bin/env python3
"""Writes tests/golden/pose_graph_state_reference_roundtrip.json: the pose-graph state the reference's own round-trip test builds
(test/file_io/cv_file_storage/object_and_reprojection_feature_pose_graph_file_storage_io_tests.cc:9-250), laid out the way
cv::FileStorage writes it as JSON (maps as sequences of k/v, ids as decimal strings, Eigen matrices as Rows/Cols/Data, doubles in
OpenCV's spelling: integral values as "4.", others with 16 significant digits and an exponent).  Data only: the VALUES of that test
-- what a checkpoint of the reference holds -- so that our reader (obvi-slam_amd/host/obvi_checkpoint_io.h) is checked on input it
did not write itself.  Also writes pose_graph_state_reference_roundtrip.expected.json: the same values in plain JSON for the test.
usage: python tests/golden/gen_checkpoint_fixture.py"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REPROJ, OBJOBS, SHAPE, LTM, RELPOSE, PAIRWISE = 0, 2, 3, 4, 5, 1     # FactorType ids (low_level_feature_pose_graph.h:18-23, object_pose_graph.h:18-20); pairwise error = 1


def num(v):
    if float(v) == int(v) and abs(v) < 1e15:
        return "%d." % int(v)
    return "%.16e" % v


def mat(rows, cols, data):
    return '{ "Rows":%d, "Cols":%d, "Data":[ %s ] }' % (rows, cols, ", ".join(num(x) for x in data))


def sid(v):
    return '"%d"' % v


def kv(k, v):
    return '{ "k":%s, "v":%s }' % (k, v)


def seq(items):
    return "[ " + ",\n".join(items) + " ]"


def pair(ft, fid):
    return '{ "f":%d, "s":%s }' % (ft, sid(fid))


def pose3d(t, angle, axis):
    return '{ "transl":%s, "rot":{ "angle":%s, "axis":%s } }' % (mat(3, 1, t), num(angle), mat(3, 1, axis))


E = dict(
    classes={"chair": ([1.2, 94.3, 92.3], [1.0, 2.1, 3.2, 4.3, 5.4, 6.5, 7.6, 8.7, 9.8]), "trashcan": ([3.2, -3.2, 18.3], [1.9, 2.0, 3.1, 4.2, 5.3, 6.4, 7.5, 8.6, 9.7])},
    min_object_id=93, max_object_id=19038,
    ellipsoids={14: [84.3, 913.3, 8.4, 19.3, 9.4, 58.2, 3.1], 94: [9.4, -184.4, 4.2, 18.3, -10.3, 4.2, 0.3]},
    semantic_class_for_object={324: "abc", 183: "def"},
    last_observed_frame_by_object={493: 139, 129: 492}, first_observed_frame_by_object={1848: 10, 19348: 193},
    min_object_observation_factor=13, max_object_observation_factor=93, min_obj_specific_factor=31, max_obj_specific_factor=193,
    long_term_map_object_ids=[13, 493, 472, 846],
    object_observation_factors={32: dict(frame=94, cam=23, obj=43, corners=[1.2, 2.3, 3.4, 1.4], cov=[1, 2, 3, 4, 11, 12, 13, 14, 21, 22, 23, 24, 31, 32, 33, 34], conf=13.4),
                                94: dict(frame=92, cam=91, obj=42, corners=[94.2, 42.4, 0.1, 92.1], cov=[0.1, 0.2, 0.3, 0.4, 1.1, 1.2, 1.3, 1.4, 2.1, 2.2, 2.3, 2.4, 3.1, 3.2, 3.3, 3.4], conf=94.1)},
    shape_dim_prior_factors={90: dict(obj=42, mean=[4.2, 0.3, 13.3], cov=[3.2, 45.2, 0.1, 34.1, 3.1, 0.4, 9.3, 2.5, 13.4]),
                             13: dict(obj=135, mean=[9.4, 13.4, 9.3], cov=[0.32, 4.52, 0.01, 3.41, 0.31, 0.04, 0.93, 0.25, 1.34])},
    observation_factors_by_frame={42: [(REPROJ, 23)], 91: [(SHAPE, 40), (PAIRWISE, 13)], 194: [(LTM, 99), (OBJOBS, 138), (RELPOSE, 924)]},
    observation_factors_by_object={84: [(REPROJ, 3), (SHAPE, 45), (OBJOBS, 914)], 76: [(LTM, 342)], 95: [(PAIRWISE, 94842), (RELPOSE, 1345)]},
    object_only_factors_by_object={24: [(RELPOSE, 84), (SHAPE, 4567), (REPROJ, 678), (LTM, 34)], 62: [(RELPOSE, 7892)]},
    extrinsics={1: ([-0.3, 4.2, 2.3], 4.3, [-.3, 12.3, -9]), 2: ([-1.3, 7.2, -2.3], 413, [-.13, 142.3, -9.1])},
    intrinsics={1: [3.2, 89.3, 0.2, 1.4, 3.4, 9.3, 0.5, 0.2, 1.3], 2: [13.2, 19.3, 1.2, 2.4, 6.4, 8.3, 1.5, 9.2, 1.5]},
    visual_factor_type=REPROJ, min_frame_id=0, max_frame_id=500, max_feature_factor_id=9825256, max_pose_factor_id=135,
    robot_poses={1: [1.2, 2.3, 3.4, 4.5, 5.6, 6.7], 2: [1.3, 2.4, 3.5, 4.6, 5.7, 6.8], 5: [1.4, 2.5, 3.6, 4.7, 5.8, 6.9]},
    pose_factors_by_frame={10: [(REPROJ, 12), (PAIRWISE, 72)], 510: [(RELPOSE, 973)], 190: [(REPROJ, 10384), (PAIRWISE, 384), (OBJOBS, 104)]},
    visual_feature_factors_by_frame={284: [(REPROJ, 13), (OBJOBS, 420)], 953: [(LTM, 134)], 344: [(LTM, 42), (OBJOBS, 3), (RELPOSE, 948)]},
    visual_factors_by_feature={24: [(RELPOSE, 21)], 94: [(OBJOBS, 124), (LTM, 13)], 301: [(PAIRWISE, 139), (SHAPE, 938), (REPROJ, 492)]},
    pose_factors={123: dict(f1=1, f2=2, t=[4.2, 0.4, -0.3], angle=-math.pi, axis=[0.4, -19.3, 48.2],
                            cov=[1.2, 4, 3.5, 10.4, -0.3, -20.3, 1.25, 4.5, 3.0, 11.4, -0.8, -21.3, 1.24, 4.4, 3.4, 12.4, -0.7, -22.3, 1.23, 4.3, 3.3, 13.4, -0.6, -23.3, 1.22, 4.2, 3.2, 14.4,
                                 -0.5, -24.3, 1.21, 4.1, 3.1, 15.4, -0.4, -25.3]),
                  94: dict(f1=3, f2=4, t=[4.6, 0.2, -9.4], angle=-math.pi / 3, axis=[-9.3, 34.2, -0.2],
                           cov=[1.2, 2.3, 3.4, 4.5, 5.6, 6.7, 11.2, 12.3, 13.4, 14.5, 15.6, 16.7, 1.21, 2.31, 3.41, 4.51, 5.61, 6.71, 21.2, 22.3, 23.4, 24.5, 25.6, 26.7, 1.22, 2.32, 3.42, 4.52,
                                5.62, 6.72, 31.2, 32.3, 33.4, 34.5, 35.6, 36.7])},
    factors={32: dict(frame=1, feat=2, cam=3, px=[1.2, 3.4], sd=4.2), 832: dict(frame=4, feat=3, cam=49, px=[-38.4, 39.4], sd=1.3)},
    last_observed_frame_by_feature={4: 1, 38: 183, 188: 973}, first_observed_frame_by_feature={5: 2, 39: 184, 189: 974},
    min_feature_id=10, max_feature_id=50, feature_positions={5: [1.2, 3.4, 5.6], 6: [2.3, 4.5, 6.7], 7: [-0.35, -483.3, 9.2]},
)


def idmap(d, val):
    return seq([kv(sid(k), val(v)) for k, v in d.items()])


def setof(pairs):
    return seq([pair(a, b) for a, b in pairs])


def build():
    low = ",\n".join([
        '"camera_extrinsics_by_camera":' + idmap(E["extrinsics"], lambda e: pose3d(*e)),
        '"camera_intrinsics_by_camera":' + idmap(E["intrinsics"], lambda k: mat(3, 3, k)),
        '"visual_factor_type":%d' % E["visual_factor_type"],
        '"min_frame_id":' + sid(E["min_frame_id"]), '"max_frame_id":' + sid(E["max_frame_id"]),
        '"max_feature_factor_id":' + sid(E["max_feature_factor_id"]), '"max_pose_factor_id":' + sid(E["max_pose_factor_id"]),
        '"robot_poses":' + idmap(E["robot_poses"], lambda p: mat(6, 1, p)),
        '"pose_factors_by_frame":' + idmap(E["pose_factors_by_frame"], setof),
        '"visual_feature_factors_by_frame":' + idmap(E["visual_feature_factors_by_frame"], lambda v: seq(['{ "i":%d, "v":%s }' % (i, pair(*p)) for i, p in enumerate(v)])),
        '"visual_factors_by_feature":' + idmap(E["visual_factors_by_feature"], setof),
        '"pose_factors":' + idmap(E["pose_factors"], lambda f: '{ "frame_id_1":%s, "frame_id_2":%s, "measured_pose_deviation":%s, "pose_deviation_cov":%s }' % (
            sid(f["f1"]), sid(f["f2"]), pose3d(f["t"], f["angle"], f["axis"]), mat(6, 6, f["cov"]))),
        '"factors":' + idmap(E["factors"], lambda f: '{ "frame_id":%s, "feature_id":%s, "camera_id":%s, "feature_pos":%s, "reprojection_error_std_dev":%s }' % (
            sid(f["frame"]), sid(f["feat"]), sid(f["cam"]), mat(2, 1, f["px"]), num(f["sd"]))),
        '"last_observed_frame_by_feature":' + idmap(E["last_observed_frame_by_feature"], sid),
        '"first_observed_frame_by_feature":' + idmap(E["first_observed_frame_by_feature"], sid),
    ])
    reproj = '{ "low_level_pg_state":{ %s },\n"min_feature_id":%s, "max_feature_id":%s,\n"feature_positions":%s }' % (
        low, sid(E["min_feature_id"]), sid(E["max_feature_id"]), idmap(E["feature_positions"], lambda p: mat(3, 1, p)))
    obj = ",\n".join([
        '"mean_and_cov_by_semantic_class":' + seq([kv('"%s"' % k, '{ "f":%s, "s":%s }' % (mat(3, 1, v[0]), mat(3, 3, v[1]))) for k, v in E["classes"].items()]),
        '"min_object_id":' + sid(E["min_object_id"]), '"max_object_id":' + sid(E["max_object_id"]),
        '"ellipsoid_estimates":' + idmap(E["ellipsoids"], lambda e: mat(7, 1, e)),
        '"semantic_class_for_object":' + idmap(E["semantic_class_for_object"], lambda c: '"%s"' % c),
        '"last_observed_frame_by_object":' + idmap(E["last_observed_frame_by_object"], sid),
        '"first_observed_frame_by_object":' + idmap(E["first_observed_frame_by_object"], sid),
        '"min_object_observation_factor":' + sid(E["min_object_observation_factor"]), '"max_object_observation_factor":' + sid(E["max_object_observation_factor"]),
        '"min_obj_specific_factor":' + sid(E["min_obj_specific_factor"]), '"max_obj_specific_factor":' + sid(E["max_obj_specific_factor"]),
        '"long_term_map_object_ids":' + seq([sid(i) for i in E["long_term_map_object_ids"]]),
        '"object_observation_factors":' + idmap(E["object_observation_factors"], lambda f: '{ "frame_id":%s, "camera_id":%s, "object_id":%s, "bounding_box_corners":%s, '
                                                '"bounding_box_corners_covariance":%s, "detection_confidence":%s }' % (sid(f["frame"]), sid(f["cam"]), sid(f["obj"]), mat(4, 1, f["corners"]),
                                                                                                                     mat(4, 4, f["cov"]), num(f["conf"]))),
        '"shape_dim_prior_factors":' + idmap(E["shape_dim_prior_factors"], lambda f: '{ "object_id":%s, "mean_shape_dim":%s, "shape_dim_cov":%s }' % (sid(f["obj"]), mat(3, 1, f["mean"]), mat(3, 3, f["cov"]))),
        '"observation_factors_by_frame":' + idmap(E["observation_factors_by_frame"], setof),
        '"observation_factors_by_object":' + idmap(E["observation_factors_by_object"], setof),
        '"object_only_factors_by_object":' + idmap(E["object_only_factors_by_object"], setof),
    ])
    return '{\n"pose_graph":{ "reprojection_low_level_feature_pose_graph_state":%s,\n"obj_only_pose_graph_state_":{ %s } }\n}\n' % (reproj, obj)


if __name__ == "__main__":
    open(os.path.join(HERE, "pose_graph_state_reference_roundtrip.json"), "w").write(build())
    exp = json.loads(json.dumps(E, default=str))
    json.dump(exp, open(os.path.join(HERE, "pose_graph_state_reference_roundtrip.expected.json"), "w"), indent=1, sort_keys=True)
    print("written")
