// motcpp::trackers::StrongSORT — constructor signature and defaults of include/motcpp/trackers/strongsort.hpp:294-312.
// ReID inference (reid_weights / use_half / use_gpu) and the ECC image registration of its camera-motion step are outside the
// hot path: pass the embeddings (N x D, one row per detection of `dets`) to update(); without them the appearance stage matches
// nothing and the IoU stage decides (strongsort.cpp:688-690). update() runs without the camera-motion step (:900-906).
// The matching is the reference's, quirks included (csrc/host/strongsort.cpp lists them); a new track is Tentative unless the
// process runs under GITHUB_ACTIONS=true, as in the reference (:61-76).
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class StrongSORT : public DeviceTracker {
 public:
  StrongSORT(const std::string& reid_weights = "", bool use_half = false, bool use_gpu = false, float det_thresh = 0.3f, int max_age = 30,
             int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f, bool per_class = false, int nr_classes = 80,
             const std::string& asso_func = "iou", bool is_obb = false, float min_conf = 0.1f, float max_cos_dist = 0.2f,
             float max_iou_dist = 0.7f, int n_init = 3, int nn_budget = 100, float mc_lambda = 0.98f, float ema_alpha = 0.9f,
             int device_index = 0);
};
}  // namespace motcpp::trackers
