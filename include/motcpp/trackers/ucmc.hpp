// motcpp::trackers::UCMCTrack — constructor signature and defaults of include/motcpp/trackers/ucmc.hpp:140-159 (reference).
// The ground-plane Kalman filter ([x, vx, y, vy], double precision), the Mahalanobis + log-determinant costs and the three assignments
// of a frame run on the GPU (csrc/host/ucmc.cpp, csrc/ucmc_kernels.hip); Ki (3 x 4) / Ko (4 x 4) are the camera matrices as the
// reference takes them (12 / 16 values, row-major), empty: its image-space fallback. update() returns the matched detections' own
// boxes for the confirmed tracks (ucmc.cpp:303-342); img and embs are not used (as in the reference).
#pragma once
#include <vector>

#include "../device_tracker.hpp"
namespace motcpp::trackers {
class UCMCTrack : public DeviceTracker {
 public:
  UCMCTrack(float det_thresh = 0.3f, int max_age = 30, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f, bool per_class = false,
            int nr_classes = 80, const std::string& asso_func = "iou", bool is_obb = false, double a1 = 100.0, double a2 = 100.0,
            double wx = 5.0, double wy = 5.0, double vmax = 10.0, double dt = 1.0 / 30.0, float high_score = 0.5f,
            const std::vector<double>& Ki = {}, const std::vector<double>& Ko = {}, int device_index = 0);
};
}  // namespace motcpp::trackers
