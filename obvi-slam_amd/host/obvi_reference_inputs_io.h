// obvi_reference_inputs_io.h -- the input files of the reference's offline executable that feed the optimisation path without a front end in between:
//   --intrinsics_file         CSV with a header line: camera_id, img_width, img_height, mat_00 ... mat_22 (row-major 3x3)    include/file_io/camera_intrinsics_with_id_io.h:14-75
//   --extrinsics_file         CSV with a header line: camera_id, transl_x, transl_y, transl_z, quat_x, quat_y, quat_z, quat_w include/file_io/camera_extrinsics_with_id_io.h:17-60
//                             (robot <- camera, taken as written: camera_info_io_utils.h:41-84)
//   --poses_by_node_id_file   CSV with a header line: node_id, x, y, z, qx, qy, qz, qw (the initial trajectory)               include/file_io/pose_3d_with_node_id_io.h:14-45
//   --low_level_feats_dir     one *.txt per frame -- line 1 the frame id, line 2 ignored, then `feature_id camera_id x y [camera_id x y ...]` with the pixels read
//                             as float -- and features/features.txt, CSV with a header line: feat_id, x, y, z (initial positions)
//                             src/refactoring/visual_feature_processing/orb_output_low_level_feature_reader.cpp:25-236, include/file_io/features_ests_with_id_io.h:14-62
// Rules of that reader kept here: a later sighting of the same (feature, frame, camera) replaces the earlier one (:184-186); frames behind the limit of
// LimitTrajectoryEvaluationParams are skipped (:48-53); a feature seen in ONE frame only is dropped (:66-70); a feature without an initial position is dropped (:73-79).
// Bounding boxes (--bounding_boxes_by_node_id_file) need the data-association front end, which is out of scope: sessions from these files are visual-feature
// sessions (BASELINE config #1's shape), optionally on top of a long-term map.
#ifndef OBVI_HOST_REFERENCE_INPUTS_IO_H_
#define OBVI_HOST_REFERENCE_INPUTS_IO_H_

#include <algorithm>
#include <cmath>
#include <filesystem>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "obvi_runner.h"

namespace vslam_types_refactor {

struct ReferenceInputFiles { std::string intrinsics_file, extrinsics_file, poses_by_node_id_file, low_level_feats_dir; };

namespace reference_inputs_detail {
// file_io_utils.h:42-65: the first line is a header; an empty file is an error (the reference exits)
inline bool read_csv_with_header(const std::string& file, size_t min_columns, std::vector<std::vector<std::string>>* rows, std::string* error) {
  std::ifstream in(file);
  std::string line;
  if (!in || !std::getline(in, line)) { *error = "the file was completely empty (and likely doesn't exist): " + file; return false; }
  while (std::getline(in, line)) {
    if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    std::vector<std::string> cells;
    std::stringstream ss(line);
    std::string cell;
    while (std::getline(ss, cell, ',')) {
      const size_t a = cell.find_first_not_of(" \t\r"), b = cell.find_last_not_of(" \t\r");
      cells.push_back(a == std::string::npos ? std::string() : cell.substr(a, b - a + 1));
    }
    if (cells.size() < min_columns) { *error = file + ": a line with " + std::to_string(cells.size()) + " entries, " + std::to_string(min_columns) + " expected"; return false; }
    rows->push_back(cells);
  }
  return true;
}
inline bool to_double(const std::string& s, double* v) { try { size_t n = 0; *v = std::stod(s, &n); return n > 0; } catch (const std::exception&) { return false; } }
inline bool to_id(const std::string& s, uint64_t* v) { try { size_t n = 0; *v = std::stoull(s, &n); return n > 0; } catch (const std::exception&) { return false; } }
// unit quaternion (x, y, z, w) -> rotation vector
inline std::array<double, 3> quat_to_axis_angle(double x, double y, double z, double w) {
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
  const double s = std::sqrt(x * x + y * y + z * z);
  if (s < 1e-15) return {{2.0 * x, 2.0 * y, 2.0 * z}};
  const double angle = 2.0 * std::atan2(s, w);
  return {{angle * x / s, angle * y / s, angle * z / s}};
}
}  // namespace reference_inputs_detail

inline bool loadReferenceInputs(const ReferenceInputFiles& files, const LimitTrajectoryEvaluationParams& limit, OfflineProblemData* d, std::string* error) {
  using namespace reference_inputs_detail;   // NOLINT
  std::string err;
  auto fail = [&](const std::string& what) { if (error) *error = what; return false; };
  {   // cameras
    std::vector<std::vector<std::string>> rows;
    if (!read_csv_with_header(files.intrinsics_file, 12, &rows, &err)) return fail(err);
    for (const auto& r : rows) {
      uint64_t cam = 0; double m[9];
      if (!to_id(r[0], &cam)) return fail(files.intrinsics_file + ": camera id expected, got " + r[0]);
      for (int k = 0; k < 9; ++k) if (!to_double(r[3 + k], &m[k])) return fail(files.intrinsics_file + ": number expected, got " + r[3 + k]);
      d->camera_intrinsics_by_camera_[(CameraId)cam] = CameraIntrinsicsMat{m[0], m[4], m[2], m[5]};
    }
    rows.clear();
    if (!read_csv_with_header(files.extrinsics_file, 8, &rows, &err)) return fail(err);
    for (const auto& r : rows) {
      uint64_t cam = 0; double v[7];
      if (!to_id(r[0], &cam)) return fail(files.extrinsics_file + ": camera id expected, got " + r[0]);
      for (int k = 0; k < 7; ++k) if (!to_double(r[1 + k], &v[k])) return fail(files.extrinsics_file + ": number expected, got " + r[1 + k]);
      CameraExtrinsics e;
      e.transl_ = {{v[0], v[1], v[2]}};
      e.orientation_ = quat_to_axis_angle(v[3], v[4], v[5], v[6]);
      d->camera_extrinsics_by_camera_[(CameraId)cam] = e;
    }
  }
  {   // the initial trajectory: one pose per node id, ids 0 ... max without a gap (the runner walks the frames in order)
    std::vector<std::vector<std::string>> rows;
    if (!read_csv_with_header(files.poses_by_node_id_file, 8, &rows, &err)) return fail(err);
    std::map<uint64_t, Pose3D> by_id;
    for (const auto& r : rows) {
      uint64_t id = 0; double v[7];
      if (!to_id(r[0], &id)) return fail(files.poses_by_node_id_file + ": node id expected, got " + r[0]);
      for (int k = 0; k < 7; ++k) if (!to_double(r[1 + k], &v[k])) return fail(files.poses_by_node_id_file + ": number expected, got " + r[1 + k]);
      Pose3D p;
      p.transl_ = {{v[0], v[1], v[2]}};
      p.orientation_ = quat_to_axis_angle(v[3], v[4], v[5], v[6]);
      by_id[id] = p;
    }
    if (by_id.empty() || by_id.begin()->first != 0 || by_id.rbegin()->first + 1 != by_id.size()) return fail(files.poses_by_node_id_file + ": node ids 0 ... n-1 expected, each once");
    d->robot_poses_.clear();
    for (const auto& e : by_id) d->robot_poses_.push_back(e.second);
  }
  const size_t n_frames = d->robot_poses_.size();
  // initial feature positions
  std::string dir = files.low_level_feats_dir;
  if (!dir.empty() && dir.back() != '/') dir += "/";
  std::map<FeatureId, Position3d> initial;
  {
    std::vector<std::vector<std::string>> rows;
    if (!read_csv_with_header(dir + "features/features.txt", 4, &rows, &err)) return fail(err);
    for (const auto& r : rows) {
      uint64_t id = 0; double v[3];
      if (!to_id(r[0], &id)) return fail(dir + "features/features.txt: feature id expected, got " + r[0]);
      for (int k = 0; k < 3; ++k) if (!to_double(r[1 + k], &v[k])) return fail(dir + "features/features.txt: number expected, got " + r[1 + k]);
      initial[(FeatureId)id] = Position3d{{v[0], v[1], v[2]}};
    }
  }
  // sightings: (frame, feature, camera) -> pixel, a later line replaces an earlier one
  std::map<FrameId, std::map<FeatureId, std::map<CameraId, PixelCoord>>> sightings;
  {
    std::error_code ec;
    std::vector<std::filesystem::path> frame_files;
    for (const auto& entry : std::filesystem::directory_iterator(std::filesystem::path(files.low_level_feats_dir), ec))
      if (entry.is_regular_file() && entry.path().extension() == ".txt") frame_files.push_back(entry.path());
    if (ec) return fail("could not list " + files.low_level_feats_dir + ": " + ec.message());
    std::sort(frame_files.begin(), frame_files.end());
    for (const auto& path : frame_files) {
      std::ifstream in(path);
      if (!in) return fail("failed to load " + path.string());
      std::string line;
      if (!std::getline(in, line)) continue;
      uint64_t frame = 0;
      { std::stringstream ss(line); if (!(ss >> frame)) return fail(path.string() + ": frame id expected on the first line"); }
      std::getline(in, line);   // the frame's pose as the feature extractor saw it: not used (:160-162)
      if (limit.should_limit_trajectory_evaluation_ && frame > limit.max_frame_id_) continue;
      if (frame >= n_frames) return fail(path.string() + ": frame " + std::to_string(frame) + " has no pose in " + files.poses_by_node_id_file);
      while (std::getline(in, line)) {
        std::stringstream ss(line);
        uint64_t feature = 0, cam = 0;
        if (!(ss >> feature)) continue;
        float x = 0, y = 0;   // (the reference parses the pixels as float: :168)
        while (ss >> cam >> x >> y) sightings[(FrameId)frame][(FeatureId)feature][(CameraId)cam] = PixelCoord{{(double)x, (double)y}};
      }
    }
  }
  std::map<FeatureId, size_t> frames_of;
  for (const auto& fr : sightings) for (const auto& ft : fr.second) ++frames_of[ft.first];
  d->visual_obs_by_frame_.assign(n_frames, {});
  d->box_obs_by_frame_.assign(n_frames, {});
  d->initial_feature_positions_.clear();
  for (const auto& fr : sightings)
    for (const auto& ft : fr.second) {
      const auto init = initial.find(ft.first);
      if (frames_of[ft.first] < 2 || init == initial.end()) continue;
      d->initial_feature_positions_[ft.first] = init->second;
      for (const auto& px : ft.second) d->visual_obs_by_frame_[fr.first].push_back(OfflineProblemData::VisualObs{ft.first, px.first, px.second});
    }
  return true;
}

}  // namespace vslam_types_refactor
#endif  // OBVI_HOST_REFERENCE_INPUTS_IO_H_
