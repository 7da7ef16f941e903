// motcpp::trackers::DeepOCSort — constructor signature and defaults of include/motcpp/trackers/deepocsort.hpp:97-119.
// ReID inference (reid_weights / use_half / use_gpu) and the image registration behind the camera-motion compensation are
// outside the hot path: pass the embeddings (N x D, one row per detection) to update() — without them update() throws
// unless embedding_off — and hand the 2 x 3 warp of the next frame over with set_camera_motion(); track states,
// last observations and the observations inside the delta_t window are then compensated (deepocsort.cpp:189-236, 633-643).
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class DeepOCSort : public DeviceTracker {
 public:
  DeepOCSort(const std::string& reid_weights = "", bool use_half = false, bool use_gpu = false, float det_thresh = 0.3f,
             int max_age = 30, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f, bool per_class = false,
             int nr_classes = 80, const std::string& asso_func = "iou", bool is_obb = false, int delta_t = 3,
             float inertia = 0.2f, float w_association_emb = 0.5f, float alpha_fixed_emb = 0.95f, float aw_param = 0.5f,
             bool embedding_off = false, bool cmc_off = false, bool aw_off = false, float Q_xy_scaling = 0.01f,
             float Q_s_scaling = 0.0001f, int device_index = 0);
  void set_camera_motion(const Eigen::MatrixXf& warp_2x3);  // std::invalid_argument unless 2 x 3
};
}  // namespace motcpp::trackers
