// motcpp::trackers::HybridSort — constructor signature and defaults of include/motcpp/trackers/hybridsort.hpp:126-164 (reference).
// Built: the tracker as the reference runs it without embeddings (csrc/host/hybridsort.cpp lists what that is: its association
// functions are the "simplified" ones) — nine-state Kalman filter, three assignments per frame on IoU / HMIoU costs, zero-measurement
// updates of unmatched tracks — on the GPU. asso_func: "hmiou", or "iou" / "giou" / "ciou" / "diou" (all plain IoU there, :579-592).
// with_reid = true behaves like the reference without embeddings and without a ReID model (all-zero features, :868-871); passing
// embeddings to update() is refused (the ReID branch is not built), as are ReID weights. The ECC step is outside the path.
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class HybridSort : public DeviceTracker {
 public:
  HybridSort(const std::string& reid_weights = "", bool use_half = false, bool use_gpu = false, float det_thresh = 0.7f, int max_age = 30,
             int max_obs = 50, int min_hits = 3, float iou_threshold = 0.15f, bool per_class = false, int nr_classes = 80,
             const std::string& asso_func = "hmiou", bool is_obb = false, float low_thresh = 0.1f, int delta_t = 3, float inertia = 0.05f,
             bool use_byte = true, bool use_custom_kf = true, int longterm_bank_length = 30, float alpha = 0.9f, bool adapfs = false,
             float track_thresh = 0.5f, float EG_weight_high_score = 4.6f, float EG_weight_low_score = 1.3f, bool TCM_first_step = true,
             bool TCM_byte_step = true, float TCM_byte_step_weight = 1.0f, float high_score_matching_thresh = 0.7f,
             bool with_longterm_reid = true, float longterm_reid_weight = 0.0f, bool with_longterm_reid_correction = true,
             float longterm_reid_correction_thresh = 0.4f, float longterm_reid_correction_thresh_low = 0.4f,
             const std::string& cmc_method = "ecc", bool with_reid = true, int device_index = 0);
};
}  // namespace motcpp::trackers
