// motcpp::trackers::BoostTrackTracker — constructor signature and defaults of include/motcpp/trackers/boosttrack.hpp:95-125 (reference).
// Constant-noise Kalman filter, detection-confidence boost (DLO; soft-BIoU and visual-tracking variants), association on 1 - IoU minus the
// weighted Mahalanobis similarity and — with_reid = true and embeddings passed to update() (N x D, one row per detection) — minus the
// weighted embedding similarity, on the GPU (csrc/host/boosttrack.cpp, csrc/boost_kernels.hip). Outside the path, as for the other trackers:
// ReID inference (reid_weights is refused) and the ECC image registration (use_ecc is accepted and no camera-motion step runs).
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class BoostTrackTracker : public DeviceTracker {
 public:
  BoostTrackTracker(const std::string& reid_weights = "", bool use_half = false, bool use_gpu = false, float det_thresh = 0.6f, int max_age = 60,
                    int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f, bool per_class = false, int nr_classes = 80,
                    const std::string& asso_func = "iou", bool is_obb = false, bool use_ecc = true, int min_box_area = 10,
                    float aspect_ratio_thresh = 1.6f, const std::string& cmc_method = "ecc", float lambda_iou = 0.5f, float lambda_mhd = 0.25f,
                    float lambda_shape = 0.25f, bool use_dlo_boost = true, bool use_duo_boost = true, float dlo_boost_coef = 0.65f,
                    bool s_sim_corr = false, bool use_rich_s = false, bool use_sb = false, bool use_vt = false, bool with_reid = false,
                    int device_index = 0);
};
}  // namespace motcpp::trackers
