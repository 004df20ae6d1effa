// obvi_ltm_io.h -- the reference's long-term object map file, read and written.
//
// A session ends by writing its map (offline_object_visual_slam_main.cpp:1070-1076: cv::FileStorage, top-level entry "long_term_map",
// SerializableIndependentEllipsoidsLongTermObjectMap: include/file_io/cv_file_storage/long_term_object_map_file_storage_io.h:29-115) and the
// next session starts from it (--long_term_map_input, :789-805): every mapped ellipsoid enters the pose graph with its estimate and a prior
// made of the estimate and its 7x7 marginal covariance (IndependentObjectMapFactor, long_term_map_factor_creator.h:265-322).  That chain is
// BASELINE config #5 in the reference's own form (src/evaluation/ltm_trajectory_sequence_executor.py:45-92); with this header it runs through
// the HIP backend on the reference's files.
//
// Layout (yaw-only ellipsoids, the only parameterisation that compiles: SURVEY.md fact 6):
//   "long_term_map": {
//     "ellipsoid_parameterization": "yaw_only",                                   :31, :101-107 (another value: the reference exits)
//     "ellipsoid_results":               {"ellipsoid_results_map": [ entry ... ]}, the map's estimates            :32-36
//     "prev_traj_est_ellipsoid_results": {"ellipsoid_results_map": [ entry ... ]}, the last trajectory's estimates :37-39
//     "obj_id_covariance_map": [ {"k": id, "v": 7x7 matrix}, ... ],                                                :40-48
//     "front_end_map_data":    [ {"k": id, "v": {}}, ... ]     (util::EmptyStruct in the offline executable, main :791-799)
//   }
//   entry = {"object_id": id, "class": name, "state": {"pose": {"transl": 3x1, "yaw": number}, "dim": 3x1}}
//           (output_problem_data_file_storage_io.h:27-69, vslam_obj_types_file_storage_io.h:31-62, vslam_basic_types_file_storage_io.h:196-226)
//   ids are decimal strings, matrices {"Rows", "Cols", "Data"} row-major (obvi_checkpoint_io.h).
#ifndef OBVI_HOST_LTM_IO_H_
#define OBVI_HOST_LTM_IO_H_

#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "obvi_checkpoint_io.h"

namespace vslam_types_refactor {

// IndependentEllipsoidsLongTermObjectMap (long_term_object_map.h) as far as the optimisation path reads it
struct LongTermObjectMapFile {
  std::map<ObjectId, std::pair<std::string, RawEllipsoid>> ellipsoid_results_;             // the map: class + [x y z yaw dx dy dz]
  std::map<ObjectId, std::pair<std::string, RawEllipsoid>> prev_traj_est_ellipsoid_results_;
  std::map<ObjectId, Covariance<7>> ellipsoid_covariances_;                                // marginal covariance of the block above, row-major
};

namespace ltm_detail {
using namespace checkpoint_detail;   // NOLINT
inline void read_results(const Value& v, std::map<ObjectId, std::pair<std::string, RawEllipsoid>>& out) {
  const Value& list = member(v, "ellipsoid_results_map");
  if (list.kind != Value::Array) throw ReadError("ellipsoid_results_map: sequence expected");
  for (const Value& e : list.array) {
    const Value& cls = member(e, "class");
    if (cls.kind != Value::String) throw ReadError("class: string expected");
    const Value& state = member(e, "state");
    const Value& pose = member(state, "pose");
    const std::array<double, 3> t = read_mat<3>(member(pose, "transl"), 3, 1), d = read_mat<3>(member(state, "dim"), 3, 1);
    out[read_id(member(e, "object_id"))] = {cls.string, RawEllipsoid{{t[0], t[1], t[2], read_num(member(pose, "yaw")), d[0], d[1], d[2]}}};
  }
}
inline void write_results(Writer& w, const std::map<ObjectId, std::pair<std::string, RawEllipsoid>>& results) {
  w.os << "{\"ellipsoid_results_map\": [";
  bool first = true;
  for (const auto& e : results) {
    const RawEllipsoid& s = e.second.second;
    w.os << (first ? "\n" : ",\n") << "  {\"object_id\": "; w.id(e.first);
    w.os << ", \"class\": "; w.str(e.second.first);
    w.os << ", \"state\": {\"pose\": {\"transl\": "; w.mat(std::array<double, 3>{{s[0], s[1], s[2]}}, 3, 1);
    w.os << ", \"yaw\": "; w.num(s[3]);
    w.os << "}, \"dim\": "; w.mat(std::array<double, 3>{{s[4], s[5], s[6]}}, 3, 1);
    w.os << "}}";
    first = false;
  }
  w.os << "]}";
}
}  // namespace ltm_detail

inline bool readLongTermObjectMapFromString(const std::string& text, LongTermObjectMapFile& map, std::string* error = nullptr) {
  using namespace ltm_detail;   // NOLINT
  obvi::json::Value root;
  std::string err;
  if (!obvi::json::Parser(text).parse(&root, &err)) { if (error) *error = "not JSON: " + err; return false; }
  try {
    const Value& m = member(root, "long_term_map");
    const Value& par = member(m, "ellipsoid_parameterization");
    if (par.kind != Value::String || par.string != "yaw_only") throw ReadError("ellipsoid_parameterization is not \"yaw_only\" (the reference exits here: long_term_object_map_file_storage_io.h:61-67)");
    LongTermObjectMapFile out;
    read_results(member(m, "ellipsoid_results"), out.ellipsoid_results_);
    read_results(member(m, "prev_traj_est_ellipsoid_results"), out.prev_traj_est_ellipsoid_results_);
    for_each_map_entry(member(m, "obj_id_covariance_map"), [&](const Value& k, const Value& v) { out.ellipsoid_covariances_[read_id(k)] = read_mat<49>(v, 7, 7); });
    // front_end_map_data: the front end's own record per object (EmptyStruct in the offline executable): not on this path
    for (const auto& e : out.ellipsoid_results_) if (!out.ellipsoid_covariances_.count(e.first)) throw ReadError("object " + std::to_string(e.first) + " has an estimate and no covariance");
    map = out;
    return true;
  } catch (const std::runtime_error& e) {
    if (error) *error = e.what();
    return false;
  }
}
inline bool readLongTermObjectMapFromFile(const std::string& file, LongTermObjectMapFile& map) {
  std::ifstream in(file, std::ios::binary);
  if (!in) { std::cerr << "could not open long-term map " << file << std::endl; return false; }
  std::ostringstream text;
  text << in.rdbuf();
  std::string error;
  if (!readLongTermObjectMapFromString(text.str(), map, &error)) { std::cerr << "long-term map " << file << ": " << error << std::endl; return false; }
  return true;
}
inline std::string writeLongTermObjectMapToString(const LongTermObjectMapFile& map) {
  using namespace ltm_detail;   // NOLINT
  Writer w;
  w.os << "{\"long_term_map\": {\n\"ellipsoid_parameterization\": \"yaw_only\",\n\"ellipsoid_results\": ";
  write_results(w, map.ellipsoid_results_);
  w.os << ",\n\"prev_traj_est_ellipsoid_results\": ";
  write_results(w, map.prev_traj_est_ellipsoid_results_);
  w.os << ",\n\"obj_id_covariance_map\": ";
  w.map(map.ellipsoid_covariances_, [&](ObjectId k) { w.id(k); }, [&](const Covariance<7>& c) { w.mat(c, 7, 7); });
  w.os << ",\n\"front_end_map_data\": ";
  w.map(map.ellipsoid_results_, [&](ObjectId k) { w.id(k); }, [&](const std::pair<std::string, RawEllipsoid>&) { w.os << "{}"; });
  w.os << "\n}}\n";
  return w.os.str();
}
inline bool writeLongTermObjectMapToFile(const std::string& file, const LongTermObjectMapFile& map) {
  std::ofstream out(file, std::ios::binary);
  if (!out) return false;
  out << writeLongTermObjectMapToString(map);
  return (bool)out;
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_LTM_IO_H_
