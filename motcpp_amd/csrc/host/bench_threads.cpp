// motcpp_bench_threads: the reference's own usage — one tracker OBJECT per camera, each on its own host thread calling
// BaseTracker::update(dets, img) (include/motcpp/tracker.hpp:67-69, docs/guides/architecture.md:242-255) — timed end to end.
// The objects are the public classes (motcpp::trackers::*), the detections are host Eigen matrices: this is the number a user of
// the drop-in surface sees, PCIe and the combiner's batching window included.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "motcpp/motcpp.hpp"
#include "motcpp_c.h"

namespace {
std::unique_ptr<motcpp::BaseTracker> make_tracker(int kind, const float* p, int np, int device) {
  auto P = [&](int i, float d) { return (p && i < np) ? p[i] : d; };
  using namespace motcpp::trackers;
  switch (kind) {
    case 0: return std::make_unique<Sort>(P(0, 0.3f), static_cast<int>(P(1, 1)), static_cast<int>(P(2, 50)), static_cast<int>(P(3, 3)), P(4, 0.3f), false, 80, "iou", false, device);
    case 1: return std::make_unique<ByteTrack>(0.3f, static_cast<int>(P(5, 30)), static_cast<int>(P(6, 50)), 3, 0.3f, false, 80, "iou", false, P(0, 0.1f), P(1, 0.45f), P(2, 0.8f),
                                               static_cast<int>(P(3, 25)), static_cast<int>(P(4, 30)), device);
    case 2: {
      static const char* const names[] = {"iou", "hmiou", "giou", "ciou", "diou", "centroid"};
      const int a = static_cast<int>(P(11, 0.f));
      return std::make_unique<OCSort>(P(0, 0.2f), static_cast<int>(P(1, 30)), static_cast<int>(P(2, 50)), static_cast<int>(P(3, 3)), P(4, 0.3f), false, 80,
                                      names[(a >= 0 && a < 6) ? a : 0], false, P(5, 0.1f), static_cast<int>(P(6, 3)), P(7, 0.2f), P(8, 0.f) != 0.f, P(9, 0.01f),
                                      P(10, 0.0001f), device);
    }
    case 3: return std::make_unique<BotSort>("", false, false, 0.3f, static_cast<int>(P(10, 30)), static_cast<int>(P(11, 50)), 3, 0.3f, false, 80, "iou", false, P(0, 0.5f),
                                             P(1, 0.1f), P(2, 0.6f), static_cast<int>(P(3, 30)), P(4, 0.8f), P(5, 0.5f), P(6, 0.25f), "none", static_cast<int>(P(7, 30)),
                                             P(8, 0.f) != 0.f, P(9, 1.f) != 0.f, device);
    // SURVEY 8 f3 (host stage machines; round 5: concurrent update() calls are merged by run_frame_combined): reference defaults, the given device
    case 4: return std::make_unique<DeepOCSort>("", false, false, 0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 3, 0.2f, 0.5f, 0.95f, 0.5f, /*embedding_off (this harness carries no embeddings)*/ true, false, false, 0.01f, 0.0001f, device);
    case 5: return std::make_unique<StrongSORT>("", false, false, 0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.1f, 0.2f, 0.7f, 3, 100, 0.98f, 0.9f, device);
    case 6: return std::make_unique<UCMCTrack>(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 100.0, 100.0, 5.0, 5.0, 10.0, 1.0 / 30.0, 0.5f, std::vector<double>{}, std::vector<double>{}, device);
    case 7: return std::make_unique<BoostTrackTracker>("", false, false, 0.6f, 60, 50, 3, 0.3f, false, 80, "iou", false, false, 10, 1.6f, "ecc", 0.5f, 0.25f, 0.25f, true, true, 0.65f,
                                                       false, false, false, false, false, device);
    case 8: return std::make_unique<HybridSort>("", false, false, 0.7f, 30, 50, 3, 0.15f, false, 80, "hmiou", false, 0.1f, 3, 0.05f, true, true, 30, 0.9f, false, 0.5f, 4.6f, 1.3f,
                                                true, true, 1.0f, 0.7f, true, 0.0f, true, 0.4f, 0.4f, "ecc", false, device);
  }
  return nullptr;
}
}  // namespace

// out5: wall seconds, timed updates, rows, mean latency (ms), largest latency (ms); lat_pct (optional, 3 doubles): p50 / p99 / p99.9 of the timed
// update() calls' latencies over all objects (ms). `frames` may exceed the number of distinct frames given (`distinct` > 0): object t then plays
// its frames back and forth (0 .. distinct-1 .. 0 ..), so that a long timed region needs no more input than a short one.
extern "C" int motcpp_bench_threads_ex(int kind, const float* params, int nparams, int T, int frames, int warm, const float* dets, const int* counts,
                                       int max_n, int device, double* out5, double* checksum, int distinct, double* lat_pct);
extern "C" int motcpp_bench_threads(int kind, const float* params, int nparams, int T, int frames, int warm, const float* dets, const int* counts,
                                    int max_n, int device, double* out5, double* checksum) {
  return motcpp_bench_threads_ex(kind, params, nparams, T, frames, warm, dets, counts, max_n, device, out5, checksum, 0, nullptr);
}
extern "C" int motcpp_bench_threads_ex(int kind, const float* params, int nparams, int T, int frames, int warm, const float* dets, const int* counts,
                                       int max_n, int device, double* out5, double* checksum, int distinct, double* lat_pct) {
  using clk = std::chrono::steady_clock;
  if (T <= 0 || frames <= 0 || warm < 0 || warm >= frames || !dets || !counts || !out5) return -1;
  std::vector<std::unique_ptr<motcpp::BaseTracker>> trk(T);
  for (int t = 0; t < T; ++t) { trk[t] = make_tracker(kind, params, nparams, device); if (!trk[t]) return -1; }
  // the frames as the caller of the reference would hold them: one column-major N x 6 matrix per frame
  const int nd = (distinct > 0 && distinct < frames) ? distinct : frames;  // distinct frames per object
  auto frame_of = [nd](int f) { if (nd <= 1) return 0; const int period = 2 * (nd - 1); const int q = f % period; return q < nd ? q : period - q; };
  std::vector<std::vector<Eigen::MatrixXf>> in(T, std::vector<Eigen::MatrixXf>(nd));
  for (int t = 0; t < T; ++t)
    for (int f = 0; f < nd; ++f) {
      const int n = counts[static_cast<size_t>(t) * nd + f];
      Eigen::MatrixXf m(n, 6);
      const float* src = dets + (static_cast<size_t>(t) * nd + f) * max_n * 6;
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < 6; ++k) m(i, k) = src[static_cast<size_t>(i) * 6 + k];
      in[t][f] = std::move(m);
    }
  cv::Mat img = cv::Mat::zeros(1080, 1920, CV_8UC3);
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  std::string err;
  std::vector<double> lat_sum(T, 0.0), lat_max(T, 0.0), csum(T, 0.0);
  std::vector<std::vector<float>> lats(T);
  if (lat_pct) for (auto& v : lats) v.reserve(static_cast<size_t>(frames - warm));
  std::vector<long> rows(T, 0);
  std::vector<clk::time_point> t0(T), t1(T);
  auto body = [&](int t) {
    try {
      for (int f = 0; f < warm; ++f) (void)trk[t]->update(in[t][frame_of(f)], img);
      {
        std::unique_lock<std::mutex> lk(mu);
        if (++arrived == T) cv.notify_all();
        else cv.wait(lk, [&] { return arrived == T; });
      }
      t0[t] = clk::now();
      for (int f = warm; f < frames; ++f) {
        const auto a = clk::now();
        const Eigen::MatrixXf out = trk[t]->update(in[t][frame_of(f)], img);
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - a).count();
        lat_sum[t] += ms;
        if (lat_pct) lats[t].push_back(static_cast<float>(ms));
        if (ms > lat_max[t]) lat_max[t] = ms;
        rows[t] += static_cast<long>(out.rows());
        for (Eigen::Index i = 0; i < out.rows(); ++i) csum[t] += static_cast<double>(out(i, 4)) * (1.0 + static_cast<double>(out(i, 7)));
      }
      t1[t] = clk::now();
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(mu);
      if (err.empty()) err = e.what();
      if (arrived < T) { arrived = T; cv.notify_all(); }  // (do not leave the others at the barrier)
      t0[t] = t1[t] = clk::now();
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(body, t);
  body(0);
  for (auto& x : th) x.join();
  if (!err.empty()) { std::fprintf(stderr, "motcpp_bench_threads: %s\n", err.c_str()); return -1; }
  clk::time_point first = t0[0], last = t1[0];
  double ls = 0.0, lm = 0.0;
  long r = 0;
  for (int t = 0; t < T; ++t) {
    if (t0[t] < first) first = t0[t];
    if (t1[t] > last) last = t1[t];
    ls += lat_sum[t]; if (lat_max[t] > lm) lm = lat_max[t];
    r += rows[t];
    if (checksum) checksum[t] = csum[t];
  }
  const double n_timed = static_cast<double>(T) * (frames - warm);
  out5[0] = std::chrono::duration<double>(last - first).count();
  out5[1] = n_timed; out5[2] = static_cast<double>(r); out5[3] = ls / n_timed; out5[4] = lm;
  if (lat_pct && std::getenv("MOTCPP_BENCH_SPIKES")) {
    // diagnostic: the calls slower than 2 ms — whole rounds (every object at the same frame) or single callers?
    std::map<int, int> per_frame;
    int spikes = 0;
    float worst = 0.f;
    const float thr = std::atof(std::getenv("MOTCPP_BENCH_SPIKES")) > 0.0 ? static_cast<float>(std::atof(std::getenv("MOTCPP_BENCH_SPIKES"))) : 2.0f;
    for (int t = 0; t < T; ++t)
      for (size_t k = 0; k < lats[t].size(); ++k)
        if (lats[t][k] > thr) { per_frame[static_cast<int>(k)] += 1; spikes += 1; if (lats[t][k] > worst) worst = lats[t][k]; }
    std::fprintf(stderr, "[bench_threads] T %d: %d calls of %zu over %.2f ms (worst %.1f ms) in %zu distinct frames:", T, spikes, static_cast<size_t>(T) * lats[0].size(), thr, worst, per_frame.size());
    int shown = 0;
    for (const auto& kv : per_frame) { if (shown++ < 24) std::fprintf(stderr, " f%d x%d", kv.first, kv.second); }
    std::fprintf(stderr, "\n");
  }
  if (lat_pct) {
    std::vector<float> all;
    for (auto& v : lats) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    auto pct = [&](double q) { return all.empty() ? 0.0 : static_cast<double>(all[static_cast<size_t>(q * static_cast<double>(all.size() - 1))]); };
    lat_pct[0] = pct(0.5); lat_pct[1] = pct(0.99); lat_pct[2] = pct(0.999);
  }
  return 0;
}
