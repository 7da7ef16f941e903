// motcpp::trackers::OCSort — constructor signature and defaults of include/motcpp/trackers/ocsort.hpp:88-108.
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class OCSort : public DeviceTracker {
 public:
  OCSort(float det_thresh = 0.2f, int max_age = 30, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f,
         bool per_class = false, int nr_classes = 80, const std::string& asso_func = "iou", bool is_obb = false,
         float min_conf = 0.1f, int delta_t = 3, float inertia = 0.2f, bool use_byte = false,
         float Q_xy_scaling = 0.01f, float Q_s_scaling = 0.0001f, int device_index = 0);
};
}  // namespace motcpp::trackers
