// motcpp::trackers::BotSort — constructor signature and defaults of include/motcpp/trackers/botsort.hpp:108-144.
// ReID inference (reid_weights) and ECC camera-motion compensation are outside the hot path: pass the
// embeddings to update(); cmc_method is accepted and no image registration runs here — a caller that has a warp for the
// frame hands it over with set_camera_motion() and the track states are compensated on the GPU.
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class BotSort : public DeviceTracker {
 public:
  BotSort(const std::string& reid_weights = "", bool use_half = false, bool use_gpu = false, float det_thresh = 0.3f,
          int max_age = 30, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f, bool per_class = false,
          int nr_classes = 80, const std::string& asso_func = "iou", bool is_obb = false,
          float track_high_thresh = 0.5f, float track_low_thresh = 0.1f, float new_track_thresh = 0.6f,
          int track_buffer = 30, float match_thresh = 0.8f, float proximity_thresh = 0.5f,
          float appearance_thresh = 0.25f, const std::string& cmc_method = "ecc", int frame_rate = 30,
          bool fuse_first_associate = false, bool with_reid = true, int device_index = 0);
  // The 2x3 warp cmc_->apply(img, dets) would return for the NEXT update() (botsort.cpp:317-324): applied to the predicted
  // pool and to the unconfirmed tracks by BotSTrack::multi_gmc's rule (:60-91) on the GPU, then forgotten. Estimating the
  // warp from pixels (ECC/ORB/SOF) is the caller's business.
  void set_camera_motion(const Eigen::MatrixXf& warp_2x3);  // std::invalid_argument unless 2 x 3
};
}  // namespace motcpp::trackers
