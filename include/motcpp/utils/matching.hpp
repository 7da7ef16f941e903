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

// fuse_iou(reid_cost, tracks_xyxy, detections_xyxy, det_confs) — matching.hpp:99-103 / matching.cpp:109-128 (confidences unused there)
Eigen::MatrixXf fuse_iou(const Eigen::MatrixXf& reid_cost_matrix, const Eigen::MatrixXf& tracks_xyxy, const Eigen::MatrixXf& detections_xyxy,
                         const Eigen::MatrixXf& det_confs = Eigen::MatrixXf(), int device_index = 0);

// The motion gate. The reference spells it kf.gating_distance(mean, covariance, measurements, only_position, metric)
// (kalman_filter.hpp:54, xywh_kf.hpp:140) and fuse_motion(kf, cost, tracks, measurements, only_position, lambda)
// (matching.hpp:60-94); here the filter is named ("xyah" = BaseKalmanFilter/KalmanFilterXYAH, "xywh" = KalmanFilterXYWH) and the
// tracks' states come as matrices: means n x 8, covariances n x 64 (row-major 8 x 8 per row), measurements m x 4.
// gating_distance returns n x m (row i = the reference's vector for track i); metric "maha" | "gaussian" (xyah only).
Eigen::MatrixXf gating_distance(const std::string& filter, const Eigen::MatrixXf& means, const Eigen::MatrixXf& covariances,
                                const Eigen::MatrixXf& measurements, bool only_position = false, const std::string& metric = "maha",
                                int device_index = 0);
Eigen::MatrixXf fuse_motion(const std::string& filter, const Eigen::MatrixXf& cost_matrix, const Eigen::MatrixXf& means,
                            const Eigen::MatrixXf& covariances, const Eigen::MatrixXf& measurements, bool only_position = false,
                            float lambda = 0.98f, int device_index = 0);
// StrongSORT's linear_assignment::gate_cost_matrix (strongsort.cpp:449-492) over the same arguments
Eigen::MatrixXf gate_cost_matrix(const std::string& filter, const Eigen::MatrixXf& cost_matrix, const Eigen::MatrixXf& means,
                                 const Eigen::MatrixXf& covariances, const Eigen::MatrixXf& measurements, float mc_lambda,
                                 float gated_cost, bool only_position = false, int device_index = 0);

}  // namespace motcpp::utils
