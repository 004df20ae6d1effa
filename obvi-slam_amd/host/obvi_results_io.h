// obvi_results_io.h -- the result files the reference's offline executable leaves behind (offline_object_visual_slam_main.cpp:1046-1105), in its layout
// (include/file_io/cv_file_storage/output_problem_data_file_storage_io.h): what its evaluation scripts and the next tools in its chain read.
//   --robot_poses_results_file     "robot_poses":  {"robot_pose_results_map": [ {"frame_id": id, "pose": Pose3D}, ... ]}            :225-296
//   --ellipsoids_results_file      "ellipsoids":   {"ellipsoid_results_map": [ {"object_id", "class", "state"}, ... ]}              :19-69, :312-320
//   --visual_feature_results_file  "visual_feats": {"visual_feature_results_map": [ {"k": id, "v": 3x1}, ... ]}                     :170-207, main :1046-1054
// Pose3D = {"transl": 3x1, "rot": {"angle", "axis": 3x1}}; ids are decimal strings; entries in ascending id order (the reference walks unordered maps).
#ifndef OBVI_HOST_RESULTS_IO_H_
#define OBVI_HOST_RESULTS_IO_H_

#include <fstream>
#include <map>
#include <string>
#include <unordered_map>

#include "obvi_ltm_io.h"

namespace vslam_types_refactor {

inline std::string writeRobotPoseResultsToString(const std::unordered_map<FrameId, RawPose3d>& robot_pose_results) {
  using namespace checkpoint_detail;   // NOLINT
  Writer w;
  const std::map<FrameId, RawPose3d> ordered(robot_pose_results.begin(), robot_pose_results.end());
  w.os << "{\"robot_poses\": {\"robot_pose_results_map\": [";
  bool first = true;
  for (const auto& e : ordered) {
    w.os << (first ? "\n" : ",\n") << "  {\"frame_id\": "; w.id(e.first);
    w.os << ", \"pose\": "; w.pose3d(convertToPose3D(e.second));
    w.os << "}";
    first = false;
  }
  w.os << "]}}\n";
  return w.os.str();
}
inline bool readRobotPoseResultsFromString(const std::string& text, std::unordered_map<FrameId, RawPose3d>& robot_pose_results, std::string* error = nullptr) {
  using namespace checkpoint_detail;   // NOLINT
  obvi::json::Value root;
  std::string err;
  if (!obvi::json::Parser(text).parse(&root, &err)) { if (error) *error = "not JSON: " + err; return false; }
  try {
    const Value& list = member(member(root, "robot_poses"), "robot_pose_results_map");
    if (list.kind != Value::Array) throw ReadError("robot_pose_results_map: sequence expected");
    std::unordered_map<FrameId, RawPose3d> out;
    for (const Value& e : list.array) out[read_id(member(e, "frame_id"))] = convertPoseToArray(read_pose3d(member(e, "pose")));
    robot_pose_results = out;
    return true;
  } catch (const std::runtime_error& e) { if (error) *error = e.what(); return false; }
}
inline std::string writeEllipsoidResultsToString(const std::map<ObjectId, std::pair<std::string, RawEllipsoid>>& ellipsoid_results) {
  checkpoint_detail::Writer w;
  w.os << "{\"ellipsoids\": ";
  ltm_detail::write_results(w, ellipsoid_results);
  w.os << "}\n";
  return w.os.str();
}
inline bool readEllipsoidResultsFromString(const std::string& text, std::map<ObjectId, std::pair<std::string, RawEllipsoid>>& ellipsoid_results, std::string* error = nullptr) {
  obvi::json::Value root;
  std::string err;
  if (!obvi::json::Parser(text).parse(&root, &err)) { if (error) *error = "not JSON: " + err; return false; }
  try {
    std::map<ObjectId, std::pair<std::string, RawEllipsoid>> out;
    ltm_detail::read_results(checkpoint_detail::member(root, "ellipsoids"), out);
    ellipsoid_results = out;
    return true;
  } catch (const std::runtime_error& e) { if (error) *error = e.what(); return false; }
}
inline std::string writeVisualFeatureResultsToString(const std::unordered_map<FeatureId, Position3d>& visual_feature_results) {
  checkpoint_detail::Writer w;
  w.os << "{\"visual_feats\": {\"visual_feature_results_map\": ";
  w.map(visual_feature_results, [&](FeatureId k) { w.id(k); }, [&](const Position3d& p) { w.mat(p, 3, 1); });
  w.os << "}}\n";
  return w.os.str();
}
inline bool writeTextFile(const std::string& file, const std::string& text) {
  std::ofstream out(file, std::ios::binary);
  if (!out) return false;
  out << text;
  return (bool)out;
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_RESULTS_IO_H_
