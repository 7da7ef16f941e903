// motcpp::BaseTracker — the reference's plugin surface (include/motcpp/tracker.hpp:33-139,
// src/tracker.cpp:17-56,108-125,166-183) kept verbatim in shape: same constructor arguments and
// defaults, same update(dets, img, embs) contract (dets N x 6 [x1,y1,x2,y2,conf,cls] column-major,
// returns M x 8 [x1,y1,x2,y2,id,conf,cls,det_ind]), same std::invalid_argument rules in check_inputs.
// The association hot path underneath runs on the MI355X through the C ABI in motcpp_amd.h.
#pragma once
#include <string>

#include "compat/eigen.hpp"
#include "compat/opencv.hpp"

namespace motcpp {

enum class TrackState { New = 0, Tracked = 1, Lost = 2, Removed = 3 };

class BaseTracker {
 public:
  BaseTracker(float det_thresh = 0.3f, int max_age = 30, int max_obs = 50, int min_hits = 3,
              float iou_threshold = 0.3f, bool per_class = false, int nr_classes = 80,
              const std::string& asso_func = "iou", bool is_obb = false);
  virtual ~BaseTracker() = default;

  virtual Eigen::MatrixXf update(const Eigen::MatrixXf& dets, const cv::Mat& img,
                                 const Eigen::MatrixXf& embs = Eigen::MatrixXf()) = 0;
  virtual void reset();
  void check_inputs(const Eigen::MatrixXf& dets, const cv::Mat& img,
                    const Eigen::MatrixXf& embs = Eigen::MatrixXf()) const;

 protected:
  void setup_association_function(const cv::Mat& img);
  void setup_detection_format(const Eigen::MatrixXf& dets);

  float det_thresh_;
  int max_age_, max_obs_, min_hits_;
  float iou_threshold_;
  bool per_class_;
  int nr_classes_;
  std::string asso_func_name_;
  bool is_obb_;
  int frame_count_ = 0;
  bool first_frame_processed_ = false, first_dets_processed_ = false;
  int frame_width_ = 0, frame_height_ = 0;
};

}  // namespace motcpp
