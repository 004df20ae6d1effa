#!/usr/bin/env python3
"""Packs the reference's own data sets into compact fixtures (data only: frame poses, pixel observations,
ground-truth features, calibration).  Run in the build container, where /root/reference exists; the
fixtures travel, the reference does not.

  data/vslam_set2, vslam_set4, vslam_set5, vslam_set6, vslam_set7   simulated sequences with ground truth (data/vslam_set2/README.md):
        pixels are the exact projections of features/features.txt from the frame poses, so the reprojection
        residual of the restated model at ground truth must vanish -- a known answer for the whole convention
        chain (quaternion -> axis-angle pose block, robot <- camera extrinsics, rectification, pixel axes).
  data/TUM_fr2_pioneer_360_consecutive_frame_matching   ORB-SLAM2 tracks of a real sequence (BASELINE config #1):
        *_curr_* files only (line 1 frame id, line 2 pose, then `id x y`); no ground-truth features.

Format read here: one text file per frame (orb_output_low_level_feature_reader.cpp:137-193 reads the newer
`id cam x y` flavour of the same layout; these directories hold the older `id x y` one), features/features.txt
(`id x y z`), calibration/camera_matrix.txt (`fx fy cx cy`).
"""
import glob
import os
import sys

import numpy as np

REF = "/root/reference/data"
OUT = os.path.dirname(os.path.abspath(__file__))


def read_frame_file(path):
    with open(path) as f:
        frame_id = int(f.readline().split()[0])
        pose = [float(v) for v in f.readline().split()]
        rows = [ln.split() for ln in f if ln.strip()]
    ids = np.array([int(r[0]) for r in rows], dtype=np.int64)
    px = np.array([[float(r[1]), float(r[2])] for r in rows], dtype=np.float64).reshape(-1, 2)
    return frame_id, np.array(pose), ids, px


def pack(directory, pattern, out_name, with_features=True, pixel_dtype=np.float64):
    files = sorted(glob.glob(os.path.join(directory, pattern)))
    frames = sorted((read_frame_file(p) for p in files), key=lambda fr: fr[0])
    frame_ids = np.array([fr[0] for fr in frames], dtype=np.int64)
    assert len(set(frame_ids.tolist())) == len(frame_ids)
    poses = np.stack([fr[1] for fr in frames])                      # [n][7]  tx ty tz qx qy qz qw
    obs_frame = np.concatenate([np.full(len(fr[2]), i, dtype=np.int32) for i, fr in enumerate(frames)])
    obs_feat = np.concatenate([fr[2] for fr in frames])
    obs_px = np.concatenate([fr[3] for fr in frames]).astype(pixel_dtype)
    K = np.loadtxt(os.path.join(directory, "calibration", "camera_matrix.txt")).reshape(-1)[-4:]
    out = dict(frame_ids=frame_ids, poses_tq=poses, obs_frame=obs_frame, obs_feature=obs_feat.astype(np.int32),
               obs_pixel=obs_px, K=K)
    if with_features:
        ft = np.loadtxt(os.path.join(directory, "features", "features.txt")).reshape(-1, 4)
        out["feature_ids"] = ft[:, 0].astype(np.int32)
        out["feature_xyz"] = ft[:, 1:4]
    np.savez_compressed(os.path.join(OUT, out_name), **out)
    print(out_name, "frames", len(frames), "obs", len(obs_feat), "bytes", os.path.getsize(os.path.join(OUT, out_name)))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    for s in ("vslam_set2", "vslam_set4", "vslam_set5", "vslam_set6", "vslam_set7"):
        pack(os.path.join(REF, s), "[0-9]*.txt", s + ".npz")
    # pixels of the ORB tracks are written with 6 decimals of a float32 detector output: float32 holds them exactly enough
    pack(os.path.join(REF, "TUM_fr2_pioneer_360_consecutive_frame_matching"), "*_curr_*.txt", "tum_fr2_360_tracks.npz",
         with_features=False, pixel_dtype=np.float32)
