// Association measures of the reference (include/motcpp/utils/iou.hpp:63-414) on the GPU: same names, argument
// meaning, result layout (N x M) and error behaviour; each call is one synchronous round trip through the C ABI
// (mot_assoc_cost_host). The oriented-box modes ("iou_obb", "centroid_obb") are outside this build's scope.
#pragma once
#include <functional>
#include <string>

#include "../compat/eigen.hpp"
#include "matching.hpp"

namespace motcpp::utils {

Eigen::MatrixXf hmiou_batch(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2, int device_index = 0);
Eigen::MatrixXf giou_batch(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2, int device_index = 0);
Eigen::MatrixXf ciou_batch(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2, int device_index = 0);
Eigen::MatrixXf diou_batch(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2, int device_index = 0);
Eigen::MatrixXf centroid_batch(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2, int frame_width, int frame_height,
                               int device_index = 0);

// Association function selector (iou.hpp:371-414): throws std::invalid_argument("Invalid association mode: ...")
class AssociationFunction {
 public:
  AssociationFunction(int w, int h, const std::string& asso_mode = "iou", int device_index = 0);
  Eigen::MatrixXf operator()(const Eigen::MatrixXf& bboxes1, const Eigen::MatrixXf& bboxes2) const;

 private:
  int frame_width_, frame_height_, kind_, device_;
};

}  // namespace motcpp::utils
