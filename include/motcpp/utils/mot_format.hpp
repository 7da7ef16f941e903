// MOT-challenge result formatting (reference: include/motcpp/utils/mot_format.hpp:23-79): same function names,
// argument meaning and byte-for-byte the same file format. Host-only (no GPU involved).
#pragma once
#include <filesystem>
#include <fstream>
#include <iomanip>

#include "../compat/eigen.hpp"

namespace motcpp::utils {

// tracks (N x 8) [x1,y1,x2,y2,id,conf,cls,det_ind] -> (N x 10) [frame,id,x1,y1,w,h,conf,-1,-1,-1]
inline Eigen::MatrixXf convert_to_mot_format(const Eigen::MatrixXf& tracks, int frame_id) {
  if (tracks.rows() == 0) return Eigen::MatrixXf(0, 10);
  Eigen::MatrixXf out(tracks.rows(), 10);
  for (int i = 0; i < tracks.rows(); ++i) {
    const float x1 = tracks(i, 0), y1 = tracks(i, 1), x2 = tracks(i, 2), y2 = tracks(i, 3);
    out(i, 0) = static_cast<float>(frame_id);
    out(i, 1) = tracks(i, 4);
    out(i, 2) = x1;
    out(i, 3) = y1;
    out(i, 4) = x2 - x1;
    out(i, 5) = y2 - y1;
    out(i, 6) = tracks(i, 5);
    out(i, 7) = -1.0f;
    out(i, 8) = -1.0f;
    out(i, 9) = -1.0f;
  }
  return out;
}

// appends "frame,id,x1,y1,w,h,conf,x,y,z" lines: integers by truncation, conf with 6 decimals
inline void write_mot_results(const std::filesystem::path& output_path, const Eigen::MatrixXf& mot_results) {
  std::filesystem::create_directories(output_path.parent_path());
  std::ofstream file(output_path, std::ios::app);
  file << std::fixed << std::setprecision(6);
  for (int i = 0; i < mot_results.rows(); ++i) {
    file << static_cast<int>(mot_results(i, 0)) << "," << static_cast<int>(mot_results(i, 1)) << ","
         << static_cast<int>(mot_results(i, 2)) << "," << static_cast<int>(mot_results(i, 3)) << ","
         << static_cast<int>(mot_results(i, 4)) << "," << static_cast<int>(mot_results(i, 5)) << "," << mot_results(i, 6) << ","
         << static_cast<int>(mot_results(i, 7)) << "," << static_cast<int>(mot_results(i, 8)) << ","
         << static_cast<int>(mot_results(i, 9)) << "\n";
  }
}

}  // namespace motcpp::utils
