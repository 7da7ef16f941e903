// The reference's internal primitive seam (include/motcpp/utils/matching.hpp:32-55,107-108) on the GPU:
// same names, argument meaning and result layout; each call is one synchronous round trip through the
// C ABI (mot_*_host). Useful for the five trackers outside this build's scope, which all funnel into
// utils::linear_assignment.
#pragma once
#include <array>
#include <string>
#include <vector>

#include "../compat/eigen.hpp"

namespace motcpp::utils {

struct LinearAssignmentResult {
  std::vector<std::array<int, 2>> matches;
  std::vector<int> unmatched_a, unmatched_b;
};
LinearAssignmentResult linear_assignment(const Eigen::MatrixXf& cost_matrix, float thresh, int device_index = 0);
Eigen::MatrixXf iou_batch(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2, int device_index = 0);
Eigen::MatrixXf iou_distance(const Eigen::MatrixXf& atracks, const Eigen::MatrixXf& btracks, int device_index = 0);
Eigen::MatrixXf embedding_distance(const Eigen::MatrixXf& track_features, const Eigen::MatrixXf& det_features,
                                   const std::string& metric = "cosine", int device_index = 0);

}  // namespace motcpp::utils
