// motcpp::trackers::ByteTrack — constructor signature and defaults of include/motcpp/trackers/bytetrack.hpp:97-116.
#pragma once
#include "../device_tracker.hpp"
namespace motcpp::trackers {
class ByteTrack : public DeviceTracker {
 public:
  ByteTrack(float det_thresh = 0.3f, int max_age = 30, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f,
            bool per_class = false, int nr_classes = 80, const std::string& asso_func = "iou", bool is_obb = false,
            float min_conf = 0.1f, float track_thresh = 0.45f, float match_thresh = 0.8f, int track_buffer = 25,
            int frame_rate = 30, int device_index = 0);
};
}  // namespace motcpp::trackers
