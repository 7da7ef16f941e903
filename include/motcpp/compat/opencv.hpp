// Minimal stand-in for cv::Mat as the four in-scope trackers use it: only empty(), rows and cols are
// ever read (src/tracker.cpp:114,168-169; src/trackers/ocsort.cpp:295-296 of the reference). Pixels are
// only touched by ECC/ReID, which are outside the hot path. A real OpenCV wins when present.
#pragma once
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#else
namespace cv {
constexpr int CV_8UC3_ = 16;
#ifndef CV_8UC3
#define CV_8UC3 16
#endif
class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() = default;
  Mat(int r, int c, int /*type*/ = CV_8UC3) : rows(r), cols(c) {}
  bool empty() const { return rows == 0 || cols == 0; }
  static Mat zeros(int r, int c, int type = CV_8UC3) { return Mat(r, c, type); }
};
}  // namespace cv
#endif
