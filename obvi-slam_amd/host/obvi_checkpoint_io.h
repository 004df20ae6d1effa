// obvi_checkpoint_io.h -- pose-graph checkpoints of ObVi-SLAM, read and written without OpenCV.
//
// The reference stores a pose-graph state with cv::FileStorage
// (include/file_io/cv_file_storage/object_and_reprojection_feature_pose_graph_file_storage_io.h: outputPoseGraphStateToFile /
// readPoseGraphStateFromFile :1021-1046, top-level key "pose_graph" :25) and replays it through run_opt_from_pg_state
// (src/refactoring/run_opt_from_pg_state.cpp:67-312: final global BA + long-term-map extraction from a checkpoint).  This header
// reads / writes the same JSON layout so that such checkpoints go through the HIP backend:
//   map            [ {"k": key, "v": value}, ... ]                      file_storage_io_utils.h:44-51, 69-70
//   pair           {"f": first, "s": second}                             :123-126, 143-144
//   vector         [ {"i": index, "v": value}, ... ]                     :193-201, 217-218
//   (hash) set     [ entry, ... ]                                        :251-255, 302-306
//   ids            decimal strings (SerializableUint64 :441-460); FactorType an int (pose-graph io :29-45)
//   Eigen matrix   {"Rows": r, "Cols": c, "Data": [row-major]}           vslam_basic_types_file_storage_io.h:19-69
//   Pose3D         {"transl": mat3x1, "rot": {"angle": a, "axis": mat3x1}}   :94-121, 144-173
// and the state labels of pose-graph io :101-105, 159-161, 228-236, 475-497, 568-573, 648-651, 906-937, 995-999.
// The JSON reader below is own code: objects, arrays, strings, numbers (also the "1." / ".5" / ".Inf" / ".Nan" spellings OpenCV's
// emitter may produce), true / false / null, and OpenCV's optional "%YAML"-less JSON prologue is plain JSON.
#ifndef OBVI_HOST_CHECKPOINT_IO_H_
#define OBVI_HOST_CHECKPOINT_IO_H_

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "obvi_pose_graph.h"

namespace obvi {
namespace json {

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  double number = 0.0;
  bool boolean = false;
  std::string string;
  std::vector<Value> array;
  std::vector<std::pair<std::string, Value>> object;
  const Value* find(const std::string& key) const {
    for (const auto& kv : object) if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}
  bool parse(Value* out, std::string* err) {
    try { skip(); *out = value(); skip(); if (p_ != s_.size()) fail("trailing characters"); return true; }
    catch (const std::runtime_error& e) { if (err) *err = e.what(); return false; }
  }

 private:
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string(what) + " at byte " + std::to_string(p_)); }
  void skip() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_; }
  bool lit(const char* w) { const size_t n = std::strlen(w); if (s_.compare(p_, n, w) == 0) { p_ += n; return true; } return false; }
  struct Depth { int& d; explicit Depth(int& x) : d(x) { ++d; } ~Depth() { --d; } };
  Value value() {
    if (p_ >= s_.size()) fail("unexpected end");
    const Depth depth(depth_);
    if (depth_ > kMaxDepth) fail("nesting deeper than 64 levels");   // a pose-graph state nests 8 deep; a hostile file must not exhaust the stack
    const char c = s_[p_];
    Value v;
    if (c == '{') {
      v.kind = Value::Object; ++p_; skip();
      if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
      for (;;) {
        skip();
        if (p_ >= s_.size() || s_[p_] != '"') fail("object key expected");
        std::string key = str();
        skip();
        if (p_ >= s_.size() || s_[p_] != ':') fail("':' expected");
        ++p_; skip();
        v.object.emplace_back(std::move(key), value());
        skip();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
        fail("',' or '}' expected");
      }
    }
    if (c == '[') {
      v.kind = Value::Array; ++p_; skip();
      if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
      for (;;) {
        skip();
        v.array.push_back(value());
        skip();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
        fail("',' or ']' expected");
      }
    }
    if (c == '"') { v.kind = Value::String; v.string = str(); return v; }
    if (lit("true")) { v.kind = Value::Bool; v.boolean = true; return v; }
    if (lit("false")) { v.kind = Value::Bool; return v; }
    if (lit("null")) return v;
    v.kind = Value::Number;
    // OpenCV spellings of the non-finite values
    if (lit(".Inf") || lit("+.Inf") || lit(".inf")) { v.number = std::numeric_limits<double>::infinity(); return v; }
    if (lit("-.Inf") || lit("-.inf")) { v.number = -std::numeric_limits<double>::infinity(); return v; }
    if (lit(".Nan") || lit(".NaN") || lit(".nan")) { v.number = std::numeric_limits<double>::quiet_NaN(); return v; }
    // std::from_chars: locale-independent (std::strtod follows the process locale: a comma-decimal locale would stop at the '.');
    // accepts "1.", ".5", exponents
    size_t e = p_;
    while (e < s_.size() && (std::isdigit((unsigned char)s_[e]) || s_[e] == '+' || s_[e] == '-' || s_[e] == '.' || s_[e] == 'e' || s_[e] == 'E')) ++e;
    if (e == p_) fail("value expected");
    const char* first = s_.data() + p_ + (s_[p_] == '+' ? 1 : 0);
    const std::from_chars_result r = std::from_chars(first, s_.data() + e, v.number);
    if (r.ec != std::errc() || r.ptr != s_.data() + e) fail("malformed number");
    p_ = e;
    return v;
  }
  std::string str() {
    std::string out;
    ++p_;
    while (p_ < s_.size() && s_[p_] != '"') {
      char c = s_[p_++];
      if (c == '\\') {
        if (p_ >= s_.size()) fail("bad escape");
        const char e = s_[p_++];
        switch (e) {
          case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break; case 'b': c = '\b'; break; case 'f': c = '\f'; break;
          case 'u': {   // \uXXXX -> UTF-8 (surrogate pairs joined)
            auto hex4 = [&]() -> unsigned {
              if (p_ + 4 > s_.size()) fail("bad \\u escape");
              unsigned code = 0;
              for (int k = 0; k < 4; ++k) {
                const char h = s_[p_++];
                code = code * 16 + (h >= '0' && h <= '9' ? (unsigned)(h - '0') : h >= 'a' && h <= 'f' ? (unsigned)(h - 'a' + 10) : h >= 'A' && h <= 'F' ? (unsigned)(h - 'A' + 10) : (fail("bad \\u escape"), 0u));
              }
              return code;
            };
            unsigned code = hex4();
            if (code >= 0xD800 && code < 0xDC00 && p_ + 6 <= s_.size() && s_[p_] == '\\' && s_[p_ + 1] == 'u') {
              p_ += 2;
              const unsigned lo = hex4();
              if (lo < 0xDC00 || lo > 0xDFFF) fail("bad surrogate pair");
              code = 0x10000 + ((code - 0xD800) << 10) + (lo - 0xDC00);
            }
            if (code < 0x80) out.push_back((char)code);
            else if (code < 0x800) { out.push_back((char)(0xC0 | (code >> 6))); out.push_back((char)(0x80 | (code & 0x3F))); }
            else if (code < 0x10000) { out.push_back((char)(0xE0 | (code >> 12))); out.push_back((char)(0x80 | ((code >> 6) & 0x3F))); out.push_back((char)(0x80 | (code & 0x3F))); }
            else { out.push_back((char)(0xF0 | (code >> 18))); out.push_back((char)(0x80 | ((code >> 12) & 0x3F))); out.push_back((char)(0x80 | ((code >> 6) & 0x3F))); out.push_back((char)(0x80 | (code & 0x3F))); }
            continue;
          }
          default: c = e;
        }
      }
      out.push_back(c);
    }
    if (p_ >= s_.size()) fail("unterminated string");
    ++p_;
    return out;
  }
  static constexpr int kMaxDepth = 64;
  const std::string& s_;
  size_t p_ = 0;
  int depth_ = 0;
};

}  // namespace json
}  // namespace obvi

namespace vslam_types_refactor {

namespace checkpoint_detail {
using obvi::json::Value;

struct ReadError : std::runtime_error { using std::runtime_error::runtime_error; };
inline const Value& member(const Value& v, const char* key) {
  if (v.kind != Value::Object) throw ReadError(std::string("object expected for member ") + key);
  const Value* m = v.find(key);
  if (!m) throw ReadError(std::string("missing member ") + key);
  return *m;
}
inline uint64_t read_id(const Value& v) {   // SerializableUint64: a decimal string (a bare number is accepted too, while a double holds it exactly)
  if (v.kind == Value::String) {
    const std::string& t = v.string;
    if (t.empty() || t.size() > 20) throw ReadError("id: a decimal string of 1 to 20 digits expected, got \"" + t + "\"");
    uint64_t id = 0;
    for (char c : t) {
      if (c < '0' || c > '9') throw ReadError("id: not a decimal number: \"" + t + "\"");
      const uint64_t d = (uint64_t)(c - '0');
      if (id > (std::numeric_limits<uint64_t>::max() - d) / 10) throw ReadError("id: exceeds 64 bits: \"" + t + "\"");
      id = id * 10 + d;
    }
    return id;
  }
  if (v.kind == Value::Number) {
    if (!(v.number >= 0.0) || v.number > 9007199254740992.0 || v.number != std::floor(v.number)) throw ReadError("id: a bare number must be a non-negative integer below 2^53");
    return (uint64_t)v.number;
  }
  throw ReadError("id expected");
}
inline double read_num(const Value& v) { if (v.kind != Value::Number) throw ReadError("number expected"); return v.number; }
template <size_t N> std::array<double, N> read_mat(const Value& v, int rows, int cols) {
  if ((int)read_num(member(v, "Rows")) != rows || (int)read_num(member(v, "Cols")) != cols) throw ReadError("matrix of another shape");
  const Value& d = member(v, "Data");
  if (d.kind != Value::Array || d.array.size() != N) throw ReadError("matrix data of another size");
  std::array<double, N> out;
  for (size_t i = 0; i < N; ++i) out[i] = read_num(d.array[i]);
  return out;
}
inline Pose3D read_pose3d(const Value& v) {
  Pose3D p;
  p.transl_ = read_mat<3>(member(v, "transl"), 3, 1);
  const Value& rot = member(v, "rot");
  const double angle = read_num(member(rot, "angle"));
  const std::array<double, 3> axis = read_mat<3>(member(rot, "axis"), 3, 1);
  for (int k = 0; k < 3; ++k) p.orientation_[k] = angle * axis[k];
  return p;
}
inline CameraIntrinsicsMat read_intrinsics(const Value& v) {
  const std::array<double, 9> k = read_mat<9>(v, 3, 3);
  return CameraIntrinsicsMat{k[0], k[4], k[2], k[5]};
}
inline FactorInfo read_factor_info(const Value& v) { return {(FactorType)(int)read_num(member(v, "f")), read_id(member(v, "s"))}; }
template <class F> void for_each_map_entry(const Value& v, F&& fn) {
  if (v.kind != Value::Array) throw ReadError("map (sequence of k/v) expected");
  for (const Value& e : v.array) fn(member(e, "k"), member(e, "v"));
}
inline FactorInfoSet read_factor_info_set(const Value& v) {
  if (v.kind != Value::Array) throw ReadError("set (sequence) expected");
  FactorInfoSet s;
  for (const Value& e : v.array) s.insert(read_factor_info(e));
  return s;
}

// ---- writer -----------------------------------------------------------------------------------------
struct Writer {
  std::ostringstream os;
  Writer() { os.imbue(std::locale::classic()); os.precision(17); }   // '.' as the decimal point whatever the process locale
  void num(double v) {
    if (std::isnan(v)) os << ".Nan"; else if (std::isinf(v)) os << (v > 0 ? ".Inf" : "-.Inf");
    else { os << v; }
  }
  void id(uint64_t v) { os << '"' << v << '"'; }
  void str(const std::string& s) { os << '"'; for (char c : s) { if (c == '"' || c == '\\') os << '\\'; os << c; } os << '"'; }
  template <size_t N> void mat(const std::array<double, N>& m, int rows, int cols) {
    os << "{\"Rows\": " << rows << ", \"Cols\": " << cols << ", \"Data\": [";
    for (size_t i = 0; i < N; ++i) { if (i) os << ", "; num(m[i]); }
    os << "]}";
  }
  void pose3d(const Pose3D& p) {
    const double a = std::sqrt(p.orientation_[0] * p.orientation_[0] + p.orientation_[1] * p.orientation_[1] + p.orientation_[2] * p.orientation_[2]);
    const std::array<double, 3> axis = a > 0.0 ? std::array<double, 3>{{p.orientation_[0] / a, p.orientation_[1] / a, p.orientation_[2] / a}} : std::array<double, 3>{{1.0, 0.0, 0.0}};
    os << "{\"transl\": "; mat(p.transl_, 3, 1);
    os << ", \"rot\": {\"angle\": "; num(a); os << ", \"axis\": "; mat(axis, 3, 1); os << "}}";
  }
  void factor_info(const FactorInfo& f) { os << "{\"f\": " << (int)f.first << ", \"s\": "; id(f.second); os << "}"; }
  template <class Set> void factor_info_set(const Set& s) {   // written in ascending order so that files are reproducible
    std::vector<FactorInfo> v(s.begin(), s.end());
    std::sort(v.begin(), v.end());
    os << "[";
    for (size_t i = 0; i < v.size(); ++i) { if (i) os << ", "; factor_info(v[i]); }
    os << "]";
  }
  // map with integer keys, ascending; `value(key, mapped)` writes the value
  template <class Map, class KeyFn, class ValFn> void map(const Map& m, KeyFn&& key, ValFn&& value) {
    std::vector<typename Map::key_type> keys;
    for (const auto& kv : m) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    os << "[";
    for (size_t i = 0; i < keys.size(); ++i) {
      if (i) os << ",\n";
      os << "{\"k\": "; key(keys[i]); os << ", \"v\": "; value(m.at(keys[i])); os << "}";
    }
    os << "]";
  }
};
}  // namespace checkpoint_detail

// readPoseGraphStateFromFile / outputPoseGraphStateToFile: the reference's entry points (pose-graph io :1021-1046).  Return false
// (and say why on stderr) on a missing or malformed file -- the reference logs an error and leaves the state untouched.
inline bool readPoseGraphStateFromString(const std::string& text, ObjectAndReprojectionFeaturePoseGraphState& pose_graph_state, std::string* error = nullptr) {
  using namespace checkpoint_detail;   // NOLINT
  obvi::json::Value root;
  std::string err;
  if (!obvi::json::Parser(text).parse(&root, &err)) { if (error) *error = "not JSON: " + err; return false; }
  try {
    const Value& pg = member(root, "pose_graph");
    const Value& rl = member(pg, "reprojection_low_level_feature_pose_graph_state");
    const Value& ll = member(rl, "low_level_pg_state");
    ObjectAndReprojectionFeaturePoseGraphState st;
    LowLevelFeaturePoseGraphState& L = st.reprojection_low_level_feature_pose_graph_state_.low_level_pg_state_;
    for_each_map_entry(member(ll, "camera_extrinsics_by_camera"), [&](const Value& k, const Value& v) { L.camera_extrinsics_by_camera_[read_id(k)] = read_pose3d(v); });
    for_each_map_entry(member(ll, "camera_intrinsics_by_camera"), [&](const Value& k, const Value& v) { L.camera_intrinsics_by_camera_[read_id(k)] = read_intrinsics(v); });
    L.visual_factor_type_ = (FactorType)(int)read_num(member(ll, "visual_factor_type"));
    L.min_frame_id_ = read_id(member(ll, "min_frame_id")); L.max_frame_id_ = read_id(member(ll, "max_frame_id"));
    L.max_feature_factor_id_ = read_id(member(ll, "max_feature_factor_id")); L.max_pose_factor_id_ = read_id(member(ll, "max_pose_factor_id"));
    for_each_map_entry(member(ll, "robot_poses"), [&](const Value& k, const Value& v) { L.robot_poses_[read_id(k)] = read_mat<6>(v, 6, 1); });
    for_each_map_entry(member(ll, "pose_factors_by_frame"), [&](const Value& k, const Value& v) { L.pose_factors_by_frame_[read_id(k)] = read_factor_info_set(v); });
    for_each_map_entry(member(ll, "visual_feature_factors_by_frame"), [&](const Value& k, const Value& v) {
      if (v.kind != Value::Array) throw ReadError("vector expected");
      std::vector<FactorInfo>& out = L.visual_feature_factors_by_frame_[read_id(k)];
      out.resize(v.array.size());
      for (const Value& e : v.array) {
        const size_t i = (size_t)read_num(member(e, "i"));
        if (i >= out.size()) throw ReadError("vector index out of range");
        out[i] = read_factor_info(member(e, "v"));
      }
    });
    for_each_map_entry(member(ll, "visual_factors_by_feature"), [&](const Value& k, const Value& v) { L.visual_factors_by_feature_[read_id(k)] = read_factor_info_set(v); });
    for_each_map_entry(member(ll, "pose_factors"), [&](const Value& k, const Value& v) {
      RelPoseFactor f;
      f.frame_id_1_ = read_id(member(v, "frame_id_1")); f.frame_id_2_ = read_id(member(v, "frame_id_2"));
      f.measured_pose_deviation_ = read_pose3d(member(v, "measured_pose_deviation")); f.pose_deviation_cov_ = read_mat<36>(member(v, "pose_deviation_cov"), 6, 6);
      L.pose_factors_[read_id(k)] = f;
    });
    for_each_map_entry(member(ll, "factors"), [&](const Value& k, const Value& v) {
      ReprojectionErrorFactor f;
      f.frame_id_ = read_id(member(v, "frame_id")); f.feature_id_ = read_id(member(v, "feature_id")); f.camera_id_ = read_id(member(v, "camera_id"));
      f.feature_pos_ = read_mat<2>(member(v, "feature_pos"), 2, 1); f.reprojection_error_std_dev_ = read_num(member(v, "reprojection_error_std_dev"));
      L.factors_[read_id(k)] = f;
    });
    for_each_map_entry(member(ll, "last_observed_frame_by_feature"), [&](const Value& k, const Value& v) { L.last_observed_frame_by_feature_[read_id(k)] = read_id(v); });
    for_each_map_entry(member(ll, "first_observed_frame_by_feature"), [&](const Value& k, const Value& v) { L.first_observed_frame_by_feature_[read_id(k)] = read_id(v); });
    ReprojectionLowLevelFeaturePoseGraphState& R = st.reprojection_low_level_feature_pose_graph_state_;
    R.min_feature_id_ = read_id(member(rl, "min_feature_id")); R.max_feature_id_ = read_id(member(rl, "max_feature_id"));
    for_each_map_entry(member(rl, "feature_positions"), [&](const Value& k, const Value& v) { R.feature_positions_[read_id(k)] = read_mat<3>(v, 3, 1); });

    const Value& ob = member(pg, "obj_only_pose_graph_state_");
    ObjOnlyPoseGraphState& O = st.obj_only_pose_graph_state_;
    for_each_map_entry(member(ob, "mean_and_cov_by_semantic_class"), [&](const Value& k, const Value& v) {
      if (k.kind != Value::String) throw ReadError("semantic class name expected");
      O.mean_and_cov_by_semantic_class_[k.string] = {read_mat<3>(member(v, "f"), 3, 1), read_mat<9>(member(v, "s"), 3, 3)};
    });
    O.min_object_id_ = read_id(member(ob, "min_object_id")); O.max_object_id_ = read_id(member(ob, "max_object_id"));
    for_each_map_entry(member(ob, "ellipsoid_estimates"), [&](const Value& k, const Value& v) { O.ellipsoid_estimates_[read_id(k)] = read_mat<7>(v, 7, 1); });
    for_each_map_entry(member(ob, "semantic_class_for_object"), [&](const Value& k, const Value& v) { if (v.kind != Value::String) throw ReadError("class name expected"); O.semantic_class_for_object_[read_id(k)] = v.string; });
    for_each_map_entry(member(ob, "last_observed_frame_by_object"), [&](const Value& k, const Value& v) { O.last_observed_frame_by_object_[read_id(k)] = read_id(v); });
    for_each_map_entry(member(ob, "first_observed_frame_by_object"), [&](const Value& k, const Value& v) { O.first_observed_frame_by_object_[read_id(k)] = read_id(v); });
    O.min_object_observation_factor_ = read_id(member(ob, "min_object_observation_factor")); O.max_object_observation_factor_ = read_id(member(ob, "max_object_observation_factor"));
    O.min_obj_specific_factor_ = read_id(member(ob, "min_obj_specific_factor")); O.max_obj_specific_factor_ = read_id(member(ob, "max_obj_specific_factor"));
    {
      const Value& ids = member(ob, "long_term_map_object_ids");
      if (ids.kind != Value::Array) throw ReadError("set expected");
      for (const Value& e : ids.array) O.long_term_map_object_ids_.insert(read_id(e));
    }
    for_each_map_entry(member(ob, "object_observation_factors"), [&](const Value& k, const Value& v) {
      ObjectObservationFactor f;
      f.frame_id_ = read_id(member(v, "frame_id")); f.camera_id_ = read_id(member(v, "camera_id")); f.object_id_ = read_id(member(v, "object_id"));
      f.bounding_box_corners_ = read_mat<4>(member(v, "bounding_box_corners"), 4, 1);
      f.bounding_box_corners_covariance_ = read_mat<16>(member(v, "bounding_box_corners_covariance"), 4, 4);
      f.detection_confidence_ = read_num(member(v, "detection_confidence"));
      O.object_observation_factors_[read_id(k)] = f;
    });
    for_each_map_entry(member(ob, "shape_dim_prior_factors"), [&](const Value& k, const Value& v) {
      ShapeDimPriorFactor f;
      f.object_id_ = read_id(member(v, "object_id")); f.mean_shape_dim_ = read_mat<3>(member(v, "mean_shape_dim"), 3, 1); f.shape_dim_cov_ = read_mat<9>(member(v, "shape_dim_cov"), 3, 3);
      O.shape_dim_prior_factors_[read_id(k)] = f;
    });
    for_each_map_entry(member(ob, "observation_factors_by_frame"), [&](const Value& k, const Value& v) { O.observation_factors_by_frame_[read_id(k)] = read_factor_info_set(v); });
    for_each_map_entry(member(ob, "observation_factors_by_object"), [&](const Value& k, const Value& v) { O.observation_factors_by_object_[read_id(k)] = read_factor_info_set(v); });
    for_each_map_entry(member(ob, "object_only_factors_by_object"), [&](const Value& k, const Value& v) { O.object_only_factors_by_object_[read_id(k)] = read_factor_info_set(v); });
    pose_graph_state = std::move(st);
    return true;
  } catch (const std::runtime_error& e) {
    if (error) *error = e.what();
    return false;
  }
}

inline bool readPoseGraphStateFromFile(const std::string& in_file, ObjectAndReprojectionFeaturePoseGraphState& pose_graph_state) {
  std::ifstream in(in_file, std::ios::binary);
  if (!in) { std::cerr << "Trying to read file " << in_file << " that does not exist" << std::endl; return false; }
  std::stringstream buf; buf << in.rdbuf();
  std::string err;
  if (!readPoseGraphStateFromString(buf.str(), pose_graph_state, &err)) { std::cerr << "Could not read pose graph state from " << in_file << ": " << err << std::endl; return false; }
  return true;
}

inline std::string poseGraphStateToString(const ObjectAndReprojectionFeaturePoseGraphState& st) {
  using namespace checkpoint_detail;   // NOLINT
  Writer w;
  auto& os = w.os;
  const LowLevelFeaturePoseGraphState& L = st.reprojection_low_level_feature_pose_graph_state_.low_level_pg_state_;
  const ReprojectionLowLevelFeaturePoseGraphState& R = st.reprojection_low_level_feature_pose_graph_state_;
  const ObjOnlyPoseGraphState& O = st.obj_only_pose_graph_state_;
  auto id_key = [&](uint64_t k) { w.id(k); };
  auto id_val = [&](uint64_t v) { w.id(v); };
  auto set_val = [&](const FactorInfoSet& s) { w.factor_info_set(s); };
  os << "{\n\"pose_graph\": {\n\"reprojection_low_level_feature_pose_graph_state\": {\n\"low_level_pg_state\": {\n";
  os << "\"camera_extrinsics_by_camera\": "; w.map(L.camera_extrinsics_by_camera_, id_key, [&](const CameraExtrinsics& e) { w.pose3d(e); });
  os << ",\n\"camera_intrinsics_by_camera\": "; w.map(L.camera_intrinsics_by_camera_, id_key, [&](const CameraIntrinsicsMat& k) { w.mat(std::array<double, 9>{{k.fx, 0, k.cx, 0, k.fy, k.cy, 0, 0, 1}}, 3, 3); });
  os << ",\n\"visual_factor_type\": " << (int)L.visual_factor_type_;
  os << ",\n\"min_frame_id\": "; w.id(L.min_frame_id_); os << ",\n\"max_frame_id\": "; w.id(L.max_frame_id_);
  os << ",\n\"max_feature_factor_id\": "; w.id(L.max_feature_factor_id_); os << ",\n\"max_pose_factor_id\": "; w.id(L.max_pose_factor_id_);
  os << ",\n\"robot_poses\": "; w.map(L.robot_poses_, id_key, [&](const RawPose3d& p) { w.mat(p, 6, 1); });
  os << ",\n\"pose_factors_by_frame\": "; w.map(L.pose_factors_by_frame_, id_key, set_val);
  os << ",\n\"visual_feature_factors_by_frame\": "; w.map(L.visual_feature_factors_by_frame_, id_key, [&](const std::vector<FactorInfo>& v) {
    os << "[";
    for (size_t i = 0; i < v.size(); ++i) { if (i) os << ", "; os << "{\"i\": " << i << ", \"v\": "; w.factor_info(v[i]); os << "}"; }
    os << "]";
  });
  os << ",\n\"visual_factors_by_feature\": "; w.map(L.visual_factors_by_feature_, id_key, set_val);
  os << ",\n\"pose_factors\": "; w.map(L.pose_factors_, id_key, [&](const RelPoseFactor& f) {
    os << "{\"frame_id_1\": "; w.id(f.frame_id_1_); os << ", \"frame_id_2\": "; w.id(f.frame_id_2_);
    os << ", \"measured_pose_deviation\": "; w.pose3d(f.measured_pose_deviation_); os << ", \"pose_deviation_cov\": "; w.mat(f.pose_deviation_cov_, 6, 6); os << "}";
  });
  os << ",\n\"factors\": "; w.map(L.factors_, id_key, [&](const ReprojectionErrorFactor& f) {
    os << "{\"frame_id\": "; w.id(f.frame_id_); os << ", \"feature_id\": "; w.id(f.feature_id_); os << ", \"camera_id\": "; w.id(f.camera_id_);
    os << ", \"feature_pos\": "; w.mat(f.feature_pos_, 2, 1); os << ", \"reprojection_error_std_dev\": "; w.num(f.reprojection_error_std_dev_); os << "}";
  });
  os << ",\n\"last_observed_frame_by_feature\": "; w.map(L.last_observed_frame_by_feature_, id_key, id_val);
  os << ",\n\"first_observed_frame_by_feature\": "; w.map(L.first_observed_frame_by_feature_, id_key, id_val);
  os << "\n},\n\"min_feature_id\": "; w.id(R.min_feature_id_); os << ",\n\"max_feature_id\": "; w.id(R.max_feature_id_);
  os << ",\n\"feature_positions\": "; w.map(R.feature_positions_, id_key, [&](const Position3d& p) { w.mat(p, 3, 1); });
  os << "\n},\n\"obj_only_pose_graph_state_\": {\n";
  os << "\"mean_and_cov_by_semantic_class\": "; w.map(O.mean_and_cov_by_semantic_class_, [&](const std::string& k) { w.str(k); }, [&](const std::pair<ObjectDim, Covariance<3>>& v) {
    os << "{\"f\": "; w.mat(v.first, 3, 1); os << ", \"s\": "; w.mat(v.second, 3, 3); os << "}";
  });
  os << ",\n\"min_object_id\": "; w.id(O.min_object_id_); os << ",\n\"max_object_id\": "; w.id(O.max_object_id_);
  os << ",\n\"ellipsoid_estimates\": "; w.map(O.ellipsoid_estimates_, id_key, [&](const RawEllipsoid& e) { w.mat(e, 7, 1); });
  os << ",\n\"semantic_class_for_object\": "; w.map(O.semantic_class_for_object_, id_key, [&](const std::string& s) { w.str(s); });
  os << ",\n\"last_observed_frame_by_object\": "; w.map(O.last_observed_frame_by_object_, id_key, id_val);
  os << ",\n\"first_observed_frame_by_object\": "; w.map(O.first_observed_frame_by_object_, id_key, id_val);
  os << ",\n\"min_object_observation_factor\": "; w.id(O.min_object_observation_factor_); os << ",\n\"max_object_observation_factor\": "; w.id(O.max_object_observation_factor_);
  os << ",\n\"min_obj_specific_factor\": "; w.id(O.min_obj_specific_factor_); os << ",\n\"max_obj_specific_factor\": "; w.id(O.max_obj_specific_factor_);
  {
    std::vector<ObjectId> ids(O.long_term_map_object_ids_.begin(), O.long_term_map_object_ids_.end());
    std::sort(ids.begin(), ids.end());
    os << ",\n\"long_term_map_object_ids\": [";
    for (size_t i = 0; i < ids.size(); ++i) { if (i) os << ", "; w.id(ids[i]); }
    os << "]";
  }
  os << ",\n\"object_observation_factors\": "; w.map(O.object_observation_factors_, id_key, [&](const ObjectObservationFactor& f) {
    os << "{\"frame_id\": "; w.id(f.frame_id_); os << ", \"camera_id\": "; w.id(f.camera_id_); os << ", \"object_id\": "; w.id(f.object_id_);
    os << ", \"bounding_box_corners\": "; w.mat(f.bounding_box_corners_, 4, 1); os << ", \"bounding_box_corners_covariance\": "; w.mat(f.bounding_box_corners_covariance_, 4, 4);
    os << ", \"detection_confidence\": "; w.num(f.detection_confidence_); os << "}";
  });
  os << ",\n\"shape_dim_prior_factors\": "; w.map(O.shape_dim_prior_factors_, id_key, [&](const ShapeDimPriorFactor& f) {
    os << "{\"object_id\": "; w.id(f.object_id_); os << ", \"mean_shape_dim\": "; w.mat(f.mean_shape_dim_, 3, 1); os << ", \"shape_dim_cov\": "; w.mat(f.shape_dim_cov_, 3, 3); os << "}";
  });
  os << ",\n\"observation_factors_by_frame\": "; w.map(O.observation_factors_by_frame_, id_key, set_val);
  os << ",\n\"observation_factors_by_object\": "; w.map(O.observation_factors_by_object_, id_key, set_val);
  os << ",\n\"object_only_factors_by_object\": "; w.map(O.object_only_factors_by_object_, id_key, set_val);
  os << "\n}\n}\n}\n";
  return os.str();
}
inline bool outputPoseGraphStateToFile(const ObjectAndReprojectionFeaturePoseGraphState& pose_graph_state, const std::string& out_file) {
  std::ofstream out(out_file, std::ios::binary);
  if (!out) return false;
  out << poseGraphStateToString(pose_graph_state);
  return (bool)out;
}

inline bool outputPoseGraphToFile(const std::shared_ptr<ObjectAndReprojectionFeaturePoseGraph>& pose_graph, const std::string& out_file) {   // pose-graph io :1048-1054
  ObjectAndReprojectionFeaturePoseGraphState pose_graph_state;
  pose_graph->getState(pose_graph_state);
  return outputPoseGraphStateToFile(pose_graph_state, out_file);
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_CHECKPOINT_IO_H_
