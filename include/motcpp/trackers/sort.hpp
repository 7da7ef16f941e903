// motcpp::trackers::Sort — constructor signature and defaults of include/motcpp/trackers/sort.hpp:69-83.
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class Sort : public DeviceTracker {
 public:
  Sort(float det_thresh = 0.3f, int max_age = 1, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f,
       bool per_class = false, int nr_classes = 80, const std::string& asso_func = "iou", bool is_obb = false,
       int device_index = 0);
};
}  // namespace motcpp::trackers
