// BaseTracker / DeviceTracker / StreamBatch and the four public tracker classes (constructor signatures of
// the reference's include/motcpp/trackers/*.hpp), plus the frame driver that steps stage machines in lockstep.
#include <chrono>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <functional>
#include <map>
#include <unordered_map>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "motcpp/motcpp.hpp"
#include "pool.hpp"
#include "staged.hpp"
#include "team.hpp"

namespace motcpp {

// ---- BaseTracker (src/tracker.cpp:17-56,108-125,166-183) -------------------------------------
BaseTracker::BaseTracker(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class,
                         int nr_classes, const std::string& asso_func, bool is_obb)
    : det_thresh_(det_thresh), max_age_(max_age), max_obs_(max_obs), min_hits_(min_hits), iou_threshold_(iou_threshold),
      per_class_(per_class), nr_classes_(nr_classes), asso_func_name_(asso_func), is_obb_(is_obb) {
  if (max_age_ >= max_obs_) max_obs_ = max_age_ + 5;
}
void BaseTracker::reset() {
  frame_count_ = 0;
  first_frame_processed_ = false;
  first_dets_processed_ = false;
}
void BaseTracker::check_inputs(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) const {
  if (dets.rows() > 0 && dets.cols() != 6 && dets.cols() != 7)
    throw std::invalid_argument("Detections must have 6 (AABB) or 7 (OBB) columns");
  if (img.empty()) throw std::invalid_argument("Image cannot be empty");
  if (embs.rows() > 0 && dets.rows() != embs.rows())
    throw std::invalid_argument("Detections and embeddings must have same number of rows");
  if (is_obb_ && dets.rows() > 0 && dets.cols() != 7)
    throw std::invalid_argument("OBB mode requires 7 columns in detections");
}
void BaseTracker::setup_association_function(const cv::Mat& img) {
  if (!first_frame_processed_ && !img.empty()) {
    frame_height_ = img.rows; frame_width_ = img.cols;
    first_frame_processed_ = true;
  }
}
void BaseTracker::setup_detection_format(const Eigen::MatrixXf& dets) {
  if (!first_dets_processed_ && dets.rows() > 0) {
    if (dets.cols() == 6) is_obb_ = false;
    else if (dets.cols() == 7) is_obb_ = true;
    first_dets_processed_ = true;
  }
}

namespace rt {
int asso_kind(const std::string& name) {
  static const char* const names[] = {"iou", "hmiou", "giou", "ciou", "diou", "centroid"};
  for (int k = 0; k < 6; ++k)
    if (name == names[k]) return k;
  return -1;
}
void run_frame(Device& dev, Staged* const* trackers, const FrameIn* inputs, int count, Team* team, std::string* errors) {
  using clk = std::chrono::steady_clock;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  std::lock_guard<std::mutex> frame_lock(dev.frame_mu);  // (trackers sharing a Device updated from several threads)
  dev.begin_frame();
  //  whatever ends this frame early — a stream's exception in the all-or-nothing mode, a failing flush — the tasks queued so far must
  // not survive into the next frame: begin_frame() resets the arenas, and a stale task would be flushed against spans that are gone
  struct ClearOnAbort {
    Device& dev; bool armed = true;
    ~ClearOnAbort() { if (armed) for (auto& l : dev.lists) l.clear(); }
  } clear_on_abort{dev};
  std::vector<char> done(count, 0);
  std::string err;
  std::mutex err_mu;
  auto fail = [&](int i, const std::exception& e) {
    if (errors) { errors[i] = e.what(); if (errors[i].empty()) errors[i] = "motcpp_amd: update() failed"; }  // (slot i is written by the one worker stepping stream i)
    else { std::lock_guard<std::mutex> g(err_mu); if (err.empty()) err = e.what(); }
  };
  // Host lifecycle of different streams is independent: the team's workers step the stage machines of their own
  // contiguous slice of streams (task lists and arena leases are per worker; kernels are launched once per stage).
  auto for_streams = [&](const std::function<void(int)>& fn) {
    if (team && count > 1) team->parallel_for(count, fn);
    else for (int i = 0; i < count; ++i) fn(i);
  };
  auto t0 = clk::now();
  for_streams([&](int i) {
    // (a tracker validates its frame at the head of begin(), before it queues anything: a stream that fails here has left no task behind)
    try { trackers[i]->begin(inputs[i]); }
    catch (const std::exception& e) { done[i] = 1; fail(i, e); }
  });
  if (!err.empty()) throw Error(err);
  dev.counters.ms_begin += ms(t0, clk::now());
  std::vector<int> any_w(team ? team->size() : 1);
  while (true) {
    auto t1 = clk::now();
    if (dev.pending()) dev.flush();
    auto t2 = clk::now();
    dev.counters.ms_flush += ms(t1, t2);
    std::fill(any_w.begin(), any_w.end(), 0);
    for_streams([&](int i) {
      if (done[i]) return;
      try {
        if (trackers[i]->advance()) any_w[Team::worker_id() % any_w.size()] = 1;
        else done[i] = 1;
      } catch (const std::exception& e) {
        done[i] = 1;
        fail(i, e);
      }
    });
    dev.counters.ms_advance += ms(t2, clk::now());
    if (!err.empty()) throw Error(err);
    int any = 0;
    for (int a : any_w) any |= a;
    if (!any) break;
  }
  clear_on_abort.armed = false;
}

// ---- rounds of host stage machines (see staged.hpp) ---------------------------------------------------------------------
namespace {
class HostRounds {
 public:
  explicit HostRounds(Device* d) : dev_(d) {
    const char* e = std::getenv("MOTCPP_BATCH_WINDOW_US");
    if (e && *e) window_us_ = std::atol(e);
  }
  void update(Staged* s, const FrameIn& in) {
    Req me{s, &in, {}};
    uint64_t r;
    bool first;
    {
      std::lock_guard<std::mutex> lk(mu_);
      r = open_;
      first = reqs_[r & 1].empty();
      reqs_[r & 1].push_back(&me);
      joined_[r & 1].store(static_cast<int>(reqs_[r & 1].size()), std::memory_order_release);
    }
    if (first) {
      const bool idle = completed_.load(std::memory_order_acquire) >= r;
      wait_completed(r, static_cast<int>((r + 1) & 1));
      lead(r, idle);
    } else {
      wait_completed(r + 1, static_cast<int>(r & 1));
    }
    if (!me.error.empty()) throw Error(me.error);
  }

 private:
  struct Req { Staged* s; const FrameIn* in; std::string error; };
  static void futex_wait(std::atomic<uint32_t>* w, uint32_t seen) { syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); }
  static void futex_wake_all(std::atomic<uint32_t>* w) { syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, 0x7fffffff, nullptr, nullptr, 0); }
  void wait_completed(uint64_t want, int p) {
    for (;;) {
      const uint32_t seen = word_[p].load(std::memory_order_acquire);
      if (completed_.load(std::memory_order_acquire) >= want) return;
      futex_wait(&word_[p], seen);
    }
  }
  void lead(uint64_t r, bool idle) {
    using clk = std::chrono::steady_clock;
    const int p = static_cast<int>(r & 1);
    // callers to expect: the trackers that took part in one of the last two rounds (as the pooled segments do, host/pool.cpp: "the larger of the
    // last two batches" let a population that once split into two alternating groups stay split — every camera then waited for two frames' worth
    // of flushes per update(): DeepOC-SORT on 64 threads had a median of 5.3 ms against 2.5 ms for the same 64 streams in one batch)
    int expected = 0;
    for (const auto& kv : seen_) expected += (kv.second + 2 > r) ? 1 : 0;
    if (idle && window_us_ > 0 && joined_[p].load(std::memory_order_acquire) < expected) {
      const auto t_a = clk::now();
      // The window scales with what a round costs here: these trackers' frames are three to five flushes, 1-3 ms with 64 cameras — a caller that
      // misses the round waits that long for the next one, so waiting a tenth of it for the cameras of the last two rounds is cheap (the fixed
      // 100 us / 20 us of the pooled segments, whose rounds are a fraction of a millisecond, closed on the first scheduling hiccup of 64 threads on 16 CPUs)
      const long scaled = last_round_us_ / 10;
      const long hard_us = window_us_ + (expected < 20 ? 2 * expected : 40) + (scaled > 0 ? scaled : 0);
      const long gap_us = 20 + (scaled > 0 ? scaled / 4 : 0);
      const auto hard = t_a + std::chrono::microseconds(hard_us);
      int seen = joined_[p].load(std::memory_order_acquire);
      auto last_arrival = t_a;
      for (;;) {
        std::this_thread::yield();
        const auto now = clk::now();
        const int j = joined_[p].load(std::memory_order_acquire);
        if (j >= expected || now >= hard) break;
        if (j != seen) { seen = j; last_arrival = now; }
        else if (now - last_arrival >= std::chrono::microseconds(gap_us)) break;
      }
    }
    //  from here on the round completes on every way out of this function — a vector that cannot grow, a worker team that cannot
    // start its threads (a pids cgroup at its limit), anything out of the frame — or the callers asleep on the futex word, and every later
    // caller of this Device, would wait for good
    std::vector<Req*> reqs;
    struct Publish {
      HostRounds* self; uint64_t r; int p; std::vector<Req*>* reqs; std::string err; bool failed = false;
      ~Publish() {
        if (failed)
          for (Req* q : *reqs)
            if (q->error.empty()) { try { q->error = err.empty() ? std::string("motcpp_amd: the round's leader failed") : err; } catch (...) {} }
        self->completed_.store(r + 1, std::memory_order_release);
        self->word_[p].fetch_add(1, std::memory_order_release);
        futex_wake_all(&self->word_[p]);
      }
    } publish{this, r, p, &reqs, {}};
    {
      std::lock_guard<std::mutex> lk(mu_);
      open_ = r + 1;
      reqs.swap(reqs_[p]);  // (no allocation: the vectors trade their buffers)
      joined_[p].store(0, std::memory_order_relaxed);
    }
    try {
      std::vector<Staged*> st(reqs.size());
      std::vector<FrameIn> in(reqs.size());
      std::vector<std::string> errs(reqs.size());
      for (size_t i = 0; i < reqs.size(); ++i) { st[i] = reqs[i]->s; in[i] = *reqs[i]->in; }
      // many cameras in one round: their stage machines are stepped by a small worker team (the host share of a frame is ~15-50 us per
      // stream and stage; one leader thread stepping 64 of them was the round's longest part); a team that cannot be created is done without
      if (st.size() >= 16 && !team_ && !team_failed_) {
        long w = 8;
        if (const char* e = std::getenv("MOTCPP_ROUND_WORKERS")) w = std::atol(e);
        if (w > 1) {
          try { team_ = std::make_unique<Team>(static_cast<int>(w)); }
          catch (...) { team_failed_ = true; }
        }
      }
      // every camera's failure is its own: the others finish their frame and get their rows (run_frame's `errors` mode)
      const auto t_run = clk::now();
      run_frame(*dev_, st.data(), in.data(), static_cast<int>(st.size()), (st.size() >= 16) ? team_.get() : nullptr, errs.data());
      for (size_t i = 0; i < reqs.size(); ++i)
        if (!errs[i].empty()) reqs[i]->error = std::move(errs[i]);
      last_round_us_ = static_cast<long>(std::chrono::duration<double, std::micro>(clk::now() - t_run).count());
      for (Req* q : reqs) seen_[q->s] = r + 1;
      if ((r & 1023) == 1023)  // trackers that went away long ago
        for (auto it = seen_.begin(); it != seen_.end();) it = (it->second + 64 < r) ? seen_.erase(it) : std::next(it);
    } catch (const std::exception& e) {
      publish.failed = true;
      try { publish.err = e.what(); } catch (...) {}
    } catch (...) {
      publish.failed = true;
    }
  }
  Device* dev_;
  std::mutex mu_;
  std::vector<Req*> reqs_[2];
  uint64_t open_ = 0;
  std::atomic<uint64_t> completed_{0};
  std::atomic<uint32_t> word_[2] = {{0}, {0}};
  std::atomic<int> joined_[2] = {{0}, {0}};
  long last_round_us_ = 0;
  std::unordered_map<Staged*, uint64_t> seen_;  // 1 + the last round a tracker took part in (leader only)
  long window_us_ = 60;
  std::unique_ptr<Team> team_;
  bool team_failed_ = false;
};
}  // namespace

void run_frame_combined(const std::shared_ptr<Device>& dev, Staged* tracker, const FrameIn& input) {
  // one combiner per Device, kept alive with it (the map holds weak owners: a Device that went away takes its rounds along)
  static std::mutex m;
  static std::map<Device*, std::pair<std::weak_ptr<Device>, std::shared_ptr<HostRounds>>> all;
  std::shared_ptr<HostRounds> hr;
  {
    std::lock_guard<std::mutex> g(m);
    auto& e = all[dev.get()];
    if (!e.second || e.first.expired()) { e.first = dev; e.second = std::make_shared<HostRounds>(dev.get()); }
    hr = e.second;
  }
  hr->update(tracker, input);
}
}  // namespace rt

// ---- DeviceTracker ---------------------------------------------------------------------------
DeviceTracker::DeviceTracker(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class,
                             int nr_classes, const std::string& asso_func, bool is_obb, int device_index)
    : BaseTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb),
      dev_(rt::Device::shared(device_index)) {}  // asso_func is only read by OC-SORT, at update time (ocsort.cpp:413)
DeviceTracker::~DeviceTracker() = default;
void DeviceTracker::adopt(rt::Staged* impl) { impl_.reset(impl); }
void DeviceTracker::adopt_pooled(int c_kind, const std::vector<float>& c_params) {
  std::vector<float> dp;
  const int pk = rt::pooled_params(c_kind, c_params.data(), static_cast<int>(c_params.size()), &dp);
  pooled_ = std::make_unique<rt::PooledStream>(dev_->index, pk, dp.data(), static_cast<int>(dp.size()));
}
void DeviceTracker::reset() {
  BaseTracker::reset();
  if (pooled_) pooled_->reset();
  else impl_->reset();
}
bool DeviceTracker::camera_motion(const float* warp2x3) {
  if (pooled_) { pooled_->set_camera_motion(warp2x3); return true; }
  return impl_->set_camera_motion(warp2x3);
}
int DeviceTracker::dump_states(std::vector<int>* ids, std::vector<float>* mean, std::vector<float>* cov) {
  if (!pooled_) throw std::logic_error("dump_states: this tracker keeps its lifecycle on the host (use the C handle's hooks)");
  return pooled_->dump(ids, mean, cov, nullptr, nullptr);
}

namespace {
rt::FrameIn make_input(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) {
  rt::FrameIn in;
  if (dets.rows() > 0 && dets.cols() == 7) throw std::invalid_argument("motcpp_amd: oriented boxes (7 columns) are out of scope");
  in.dets = dets.data(); in.n = static_cast<int>(dets.rows()); in.ld = static_cast<int>(dets.rows());
  if (embs.rows() > 0 && embs.cols() > 0) {
    in.embs = embs.data(); in.emb_ld = static_cast<int>(embs.rows()); in.emb_dim = static_cast<int>(embs.cols());
  }
  in.img_w = img.cols; in.img_h = img.rows;
  return in;
}
Eigen::MatrixXf to_matrix(const float* rows, int m) {
  Eigen::MatrixXf out(m, 8);
  for (int i = 0; i < m; ++i)
    for (int k = 0; k < 8; ++k) out(i, k) = rows[static_cast<size_t>(i) * 8 + k];
  return out;
}
Eigen::MatrixXf to_matrix(const std::vector<float>& rows) { return to_matrix(rows.data(), static_cast<int>(rows.size() / 8)); }
rt::PooledFrame make_pooled_input(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) {
  rt::PooledFrame f;
  f.dets = dets.data(); f.n = static_cast<int>(dets.rows()); f.ld = static_cast<int>(dets.rows());
  if (embs.rows() > 0 && embs.cols() > 0) { f.embs = embs.data(); f.emb_ld = static_cast<int>(embs.rows()); f.emb_dim = static_cast<int>(embs.cols()); }
  f.img_w = img.cols; f.img_h = img.rows;
  return f;
}
}  // namespace

// What the reference's update() does before it touches a track (checks, format and association setup, frame counter):
// shared by the single-tracker call and by StreamBatch, so that a batched tracker rejects exactly what it rejects alone.
// false: the frame is skipped (BoT-SORT's empty-frame early return, botsort.cpp:267-269).
void DeviceTracker::validate_update(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) const {
  if (!asso_error_.empty()) throw std::invalid_argument(asso_error_);
  if (validate_inputs_) check_inputs(dets, img, skip_empty_ ? Eigen::MatrixXf() : embs);
  // botsort.cpp reads embs.row(first_indices[i]) under an index guard; here the rows are consumed wholesale on the device
  if (embs.rows() > 0 && embs.cols() > 0 && embs.rows() != dets.rows())
    throw std::invalid_argument("motcpp_amd: embs must have one row per detection (" + std::to_string(embs.rows()) + " vs " +
                                std::to_string(dets.rows()) + ")");
  if (dets.rows() > 0 && dets.cols() == 7) throw std::invalid_argument("motcpp_amd: oriented boxes (7 columns) are out of scope");
}
bool DeviceTracker::commit_update(const Eigen::MatrixXf& dets, const cv::Mat& img) {
  if (skip_empty_ && dets.rows() == 0) {
    camera_motion(nullptr);  // the reference returns before its CMC step: this frame's warp is dropped
    return false;
  }
  setup_detection_format(dets);
  setup_association_function(img);
  ++frame_count_;
  return true;
}
bool DeviceTracker::prepare_update(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) {
  validate_update(dets, img, embs);
  return commit_update(dets, img);
}

Eigen::MatrixXf DeviceTracker::update(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) {
  if (!prepare_update(dets, img, embs)) return Eigen::MatrixXf(0, 8);
  if (pooled_) {  // one stream of a shared device batch: calls that arrive together run as ONE launch sequence (host/pool.hpp)
    const rt::PooledFrame f = make_pooled_input(dets, img, embs);
    const float* rows = nullptr;
    const int m = pooled_->update(f, &rows);
    return to_matrix(rows, m);
  }
  rt::FrameIn in = make_input(dets, img, embs);
  rt::run_frame_combined(dev_, impl_.get(), in);  // merged with the calls other threads make at the same time (staged.hpp)
  return to_matrix(impl_->rows());
}

StreamBatch::StreamBatch(std::vector<DeviceTracker*> trackers) : trackers_(std::move(trackers)) {
  for (DeviceTracker* t : trackers_)
    if (t->device().get() != trackers_[0]->device().get()) throw std::invalid_argument("StreamBatch: trackers must share one device");
}
std::vector<Eigen::MatrixXf> StreamBatch::update(const std::vector<Eigen::MatrixXf>& dets, const cv::Mat& img,
                                                 const std::vector<Eigen::MatrixXf>& embs) {
  if (dets.size() != trackers_.size()) throw std::invalid_argument("StreamBatch: one detection matrix per stream");
  std::vector<rt::FrameIn> in;
  std::vector<rt::Staged*> st;
  std::vector<int> who;
  std::vector<rt::PooledFrame> pin;  // the trackers whose lifecycle is a pooled device stream: one round for all of them
  std::vector<rt::PooledStream*> pst;
  std::vector<int> pwho;
  static const Eigen::MatrixXf kNone;
  // every stream goes through its tracker's own checks — all of them before any tracker's bookkeeping, so that a rejected stream
  // leaves no other tracker a frame ahead of its device state
  for (size_t i = 0; i < trackers_.size(); ++i) trackers_[i]->validate_update(dets[i], img, i < embs.size() ? embs[i] : kNone);
  for (size_t i = 0; i < trackers_.size(); ++i) {
    const Eigen::MatrixXf& e = i < embs.size() ? embs[i] : kNone;
    if (!trackers_[i]->commit_update(dets[i], img)) continue;  // skipped frame: empty table
    if (trackers_[i]->pooled()) {
      pin.push_back(make_pooled_input(dets[i], img, e));
      pst.push_back(trackers_[i]->pooled());
      pwho.push_back(static_cast<int>(i));
      continue;
    }
    in.push_back(make_input(dets[i], img, e));
    st.push_back(trackers_[i]->staged());
    who.push_back(static_cast<int>(i));
  }
  if (!st.empty()) rt::run_frame(*trackers_[0]->device(), st.data(), in.data(), static_cast<int>(st.size()));
  std::vector<Eigen::MatrixXf> out(trackers_.size(), Eigen::MatrixXf(0, 8));
  for (size_t k = 0; k < st.size(); ++k) out[who[k]] = to_matrix(st[k]->rows());
  if (!pst.empty()) {
    std::vector<const float*> rows(pst.size(), nullptr);
    std::vector<int> counts(pst.size(), 0);
    rt::PooledStream::update_many(pst.data(), pin.data(), static_cast<int>(pst.size()), rows.data(), counts.data());
    for (size_t k = 0; k < pst.size(); ++k) out[pwho[k]] = to_matrix(rows[k], counts[k]);
  }
  return out;
}

// ---- public tracker classes --------------------------------------------------------------------
namespace trackers {
Sort::Sort(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class, int nr_classes,
           const std::string& asso_func, bool is_obb, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  validate_inputs_ = false;
  if (rt::pooling_enabled()) adopt_pooled(0, {det_thresh_, static_cast<float>(max_age_), static_cast<float>(max_obs_), static_cast<float>(min_hits_), iou_threshold_});
  else adopt(rt::make_sort(dev_, det_thresh_, max_age_, max_obs_, min_hits_, iou_threshold_));
}
ByteTrack::ByteTrack(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class,
                     int nr_classes, const std::string& asso_func, bool is_obb, float min_conf, float track_thresh,
                     float match_thresh, int track_buffer, int frame_rate, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  det_thresh_ = track_thresh;
  if (rt::pooling_enabled()) adopt_pooled(1, {min_conf, track_thresh, match_thresh, static_cast<float>(track_buffer), static_cast<float>(frame_rate)});
  else adopt(rt::make_bytetrack(dev_, min_conf, track_thresh, match_thresh, track_buffer, frame_rate, max_age_, max_obs_));
}
OCSort::OCSort(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class, int nr_classes,
               const std::string& asso_func, bool is_obb, float min_conf, int delta_t, float inertia, bool use_byte,
               float Q_xy_scaling, float Q_s_scaling, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  int asso = rt::asso_kind(asso_func);
  if (asso < 0) {  // like the reference, the constructor accepts any name and update() throws (AssociationFunction, iou.hpp:405-407)
    asso_error_ = (asso_func == "iou_obb" || asso_func == "centroid_obb")
                      ? "motcpp_amd: oriented-box association mode '" + asso_func + "' is out of scope"
                      : "Invalid association mode: " + asso_func;
    asso = 0;
  }
  if (rt::pooling_enabled() && delta_t >= 0 && delta_t <= 64)
    adopt_pooled(2, {det_thresh_, static_cast<float>(max_age_), static_cast<float>(max_obs_), static_cast<float>(min_hits_), iou_threshold_, min_conf,
                     static_cast<float>(delta_t), inertia, use_byte ? 1.f : 0.f, Q_xy_scaling, Q_s_scaling, static_cast<float>(asso)});
  else
    adopt(rt::make_ocsort(dev_, det_thresh_, max_age_, max_obs_, min_hits_, iou_threshold_, min_conf, delta_t, inertia, use_byte,
                          Q_xy_scaling, Q_s_scaling, asso));
}
BotSort::BotSort(const std::string& reid_weights, bool /*use_half*/, bool /*use_gpu*/, float det_thresh, int max_age, int max_obs,
                 int min_hits, float iou_threshold, bool per_class, int nr_classes, const std::string& asso_func, bool is_obb,
                 float track_high_thresh, float track_low_thresh, float new_track_thresh, int track_buffer, float match_thresh,
                 float proximity_thresh, float appearance_thresh, const std::string& /*cmc_method*/, int frame_rate,
                 bool fuse_first_associate, bool with_reid, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  if (!reid_weights.empty())
    throw std::invalid_argument("motcpp_amd: ReID model inference is outside the hot path; pass embeddings to update()");
  skip_empty_ = true;
  if (rt::pooling_enabled())
    adopt_pooled(3, {track_high_thresh, track_low_thresh, new_track_thresh, static_cast<float>(track_buffer), match_thresh, proximity_thresh,
                     appearance_thresh, static_cast<float>(frame_rate), fuse_first_associate ? 1.f : 0.f, with_reid ? 1.f : 0.f});
  else
    adopt(rt::make_botsort(dev_, track_high_thresh, track_low_thresh, new_track_thresh, track_buffer, match_thresh,
                           proximity_thresh, appearance_thresh, frame_rate, fuse_first_associate, with_reid, max_age_, max_obs_));
}
DeepOCSort::DeepOCSort(const std::string& /*reid_weights*/, bool /*use_half*/, bool /*use_gpu*/, float det_thresh, int max_age, int max_obs,
                       int min_hits, float iou_threshold, bool per_class, int nr_classes, const std::string& asso_func, bool is_obb,
                       int delta_t, float inertia, float w_association_emb, float alpha_fixed_emb, float aw_param, bool embedding_off,
                       bool cmc_off, bool aw_off, float Q_xy_scaling, float Q_s_scaling, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  int asso = rt::asso_kind(asso_func);
  if (asso < 0) {  // the association function is built inside update() (deepocsort.cpp:762): an unknown name throws there
    asso_error_ = (asso_func == "iou_obb" || asso_func == "centroid_obb")
                      ? "motcpp_amd: oriented-box association mode '" + asso_func + "' is out of scope"
                      : "Invalid association mode: " + asso_func;
    asso = 0;
  }
  adopt(rt::make_deepocsort(dev_, det_thresh_, max_age_, max_obs_, min_hits_, iou_threshold_, delta_t, inertia, w_association_emb,
                            alpha_fixed_emb, aw_param, embedding_off, cmc_off, aw_off, Q_xy_scaling, Q_s_scaling, asso));
}
StrongSORT::StrongSORT(const std::string& reid_weights, bool /*use_half*/, bool /*use_gpu*/, float det_thresh, int max_age, int max_obs, int min_hits,
                       float iou_threshold, bool per_class, int nr_classes, const std::string& asso_func, bool is_obb, float min_conf,
                       float max_cos_dist, float max_iou_dist, int n_init, int nn_budget, float mc_lambda, float ema_alpha, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  if (!reid_weights.empty())
    throw std::invalid_argument("motcpp_amd: ReID model inference is outside the hot path; pass embeddings to update()");
  // (the tracker gets the constructor's max_age, strongsort.cpp:841-842; BaseTracker only adjusts max_obs)
  adopt(rt::make_strongsort(dev_, min_conf, max_cos_dist, max_iou_dist, n_init, nn_budget, mc_lambda, ema_alpha, max_age));
}
BoostTrackTracker::BoostTrackTracker(const std::string& reid_weights, bool /*use_half*/, bool /*use_gpu*/, float det_thresh, int max_age, int max_obs,
                                     int min_hits, float iou_threshold, bool per_class, int nr_classes, const std::string& asso_func, bool is_obb,
                                     bool /*use_ecc*/, int min_box_area, float aspect_ratio_thresh, const std::string& /*cmc_method*/, float lambda_iou,
                                     float lambda_mhd, float lambda_shape, bool use_dlo_boost, bool use_duo_boost, float dlo_boost_coef,
                                     bool /*s_sim_corr*/, bool /*use_rich_s*/, bool use_sb, bool use_vt, bool with_reid, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  if (!reid_weights.empty())
    throw std::invalid_argument("motcpp_amd: ReID model inference is outside the hot path; pass embeddings to update() (with_reid = true)");
  rt::BoostParams q;
  q.det_thresh = det_thresh_; q.max_age = max_age_; q.min_hits = min_hits_; q.iou_threshold = iou_threshold_; q.min_box_area = min_box_area;
  q.aspect_ratio_thresh = aspect_ratio_thresh; q.lambda_iou = lambda_iou; q.lambda_mhd = lambda_mhd; q.lambda_shape = lambda_shape;
  q.use_dlo = use_dlo_boost; q.use_duo = use_duo_boost; q.dlo_coef = dlo_boost_coef; q.use_sb = use_sb; q.use_vt = use_vt; q.with_reid = with_reid;
  adopt(rt::make_boosttrack(dev_, q));
}
HybridSort::HybridSort(const std::string& reid_weights, bool /*use_half*/, bool /*use_gpu*/, float det_thresh, int max_age, int max_obs, int min_hits,
                       float iou_threshold, bool per_class, int nr_classes, const std::string& asso_func, bool is_obb, float low_thresh, int /*delta_t*/,
                       float /*inertia*/, bool use_byte, bool /*use_custom_kf*/, int /*longterm_bank_length*/, float /*alpha*/, bool /*adapfs*/,
                       float track_thresh, float EG_weight_high_score, float EG_weight_low_score, bool TCM_first_step, bool TCM_byte_step,
                       float TCM_byte_step_weight, float /*high_score_matching_thresh*/, bool /*with_longterm_reid*/, float /*longterm_reid_weight*/,
                       bool /*with_longterm_reid_correction*/, float /*longterm_reid_correction_thresh*/, float /*longterm_reid_correction_thresh_low*/,
                       const std::string& /*cmc_method*/, bool with_reid, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  if (!reid_weights.empty()) throw std::invalid_argument("motcpp_amd: ReID model inference is outside the hot path");
  rt::HybridParams q;
  q.det_thresh = det_thresh_; q.max_age = max_age_; q.min_hits = min_hits_; q.iou_threshold = iou_threshold_;
  // with ReID the reference only knows hmiou and IoU (:755-759); without, everything but hmiou and ct_dist is IoU (:645-665)
  if (asso_func == "ct_dist") throw std::invalid_argument("motcpp_amd: HybridSORT's ct_dist measure is not built");
  q.asso = (asso_func == "hmiou") ? 1 : 0;
  q.low_thresh = low_thresh; q.use_byte = use_byte; q.track_thresh = track_thresh; q.eg_high = EG_weight_high_score; q.eg_low = EG_weight_low_score;
  q.tcm_first = TCM_first_step; q.tcm_byte = TCM_byte_step; q.tcm_byte_weight = TCM_byte_step_weight; q.with_reid = with_reid;
  adopt(rt::make_hybridsort(dev_, q));
}
UCMCTrack::UCMCTrack(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class, int nr_classes,
                     const std::string& asso_func, bool is_obb, double a1, double a2, double wx, double wy, double vmax, double dt, float high_score,
                     const std::vector<double>& Ki, const std::vector<double>& Ko, int device_index)
    : DeviceTracker(det_thresh, max_age, max_obs, min_hits, iou_threshold, per_class, nr_classes, asso_func, is_obb, device_index) {
  rt::UcmcParams q;
  q.det_thresh = det_thresh_; q.max_age = max_age_; q.a1 = a1; q.a2 = a2; q.wx = wx; q.wy = wy; q.vmax = vmax; q.dt = dt; q.high_score = high_score;
  if (!Ki.empty() && !Ko.empty() && Ki.size() == 12 && Ko.size() == 16) {  // (any other size: the mapper stays invalid, ucmc.cpp:60-62)
    q.has_camera = true;
    for (int k = 0; k < 12; ++k) q.Ki[k] = Ki[k];
    for (int k = 0; k < 16; ++k) q.Ko[k] = Ko[k];
  }
  adopt(rt::make_ucmc(dev_, q));
}
void DeepOCSort::set_camera_motion(const Eigen::MatrixXf& warp) {
  if (warp.rows() != 2 || warp.cols() != 3) throw std::invalid_argument("DeepOCSort::set_camera_motion: the warp must be 2 x 3");
  const float w[6] = {warp(0, 0), warp(0, 1), warp(0, 2), warp(1, 0), warp(1, 1), warp(1, 2)};
  camera_motion(w);
}
void BotSort::set_camera_motion(const Eigen::MatrixXf& warp) {
  if (warp.rows() != 2 || warp.cols() != 3) throw std::invalid_argument("BotSort::set_camera_motion: the warp must be 2 x 3");
  const float w[6] = {warp(0, 0), warp(0, 1), warp(0, 2), warp(1, 0), warp(1, 1), warp(1, 2)};
  camera_motion(w);
}
}  // namespace trackers

// ---- DeviceLifecycleBatch: ByteTrack (mot_bt_*), SORT (mot_sort_*), OC-SORT (mot_oc_*), BoT-SORT (mot_bot_*) on the device ----------
struct DeviceLifecycleBatch::Impl {
  std::shared_ptr<rt::Device> dev;
  int kind = 0;  // 0 ByteTrack, 1 SORT, 2 OC-SORT, 3 BoT-SORT
  int emb_dim = 0;
  mot_bt_batch* bt = nullptr;
  mot_sort_batch* so = nullptr;
  mot_oc_batch* oc = nullptr;
  mot_bot_batch* bot = nullptr;
  void* d_dets = nullptr;
  void* d_embs = nullptr;
  std::vector<float> soa, out, embs, warps;
  std::vector<unsigned char> has_warp;
  std::vector<int> counts, out_counts;
  ~Impl() {  // (also runs when the constructor below throws after the batch exists)
    if (bt) mot_bt_destroy(bt);
    if (so) mot_sort_destroy(so);
    if (oc) mot_oc_destroy(oc);
    if (bot) mot_bot_destroy(bot);
    if (d_dets) mot_free(dev->ctx, d_dets);
    if (d_embs) mot_free(dev->ctx, d_embs);
  }
};
DeviceLifecycleBatch::DeviceLifecycleBatch(int kind, int nstreams, int cap_tracks, int max_dets, const float* p, int device_index, int emb_dim)
    : impl_(std::make_unique<Impl>()), n_(nstreams), cap_(cap_tracks), maxd_(max_dets) {
  impl_->dev = rt::Device::shared(device_index);
  impl_->kind = kind;
  impl_->emb_dim = emb_dim;
  mot_ctx* ctx = impl_->dev->ctx;
  std::lock_guard<std::mutex> dev_lock(impl_->dev->frame_mu);
  int rc = MOT_ERR_INVALID;
  if (kind == 0) rc = mot_bt_create(ctx, nstreams, cap_tracks, max_dets, p, &impl_->bt);
  else if (kind == 1) rc = mot_sort_create(ctx, nstreams, cap_tracks, max_dets, p, &impl_->so);
  else if (kind == 2) rc = mot_oc_create(ctx, nstreams, cap_tracks, max_dets, p, &impl_->oc);
  else if (kind == 3) rc = mot_bot_create(ctx, nstreams, cap_tracks, max_dets, emb_dim, p, &impl_->bot);
  if (rc != MOT_OK) throw std::runtime_error(std::string("motcpp_amd: device-lifecycle batch creation failed: ") + mot_ctx_last_error(ctx));
  impl_->soa.assign(static_cast<size_t>(nstreams) * 6 * max_dets, 0.f);
  impl_->counts.assign(nstreams, 0);
  impl_->out_counts.assign(nstreams, 0);
  if (mot_malloc(ctx, impl_->soa.size() * sizeof(float), &impl_->d_dets) != MOT_OK) throw std::runtime_error("motcpp_amd: device allocation failed");
  if (kind == 3 && emb_dim > 0) {
    impl_->embs.assign(static_cast<size_t>(nstreams) * max_dets * emb_dim, 0.f);
    if (mot_malloc(ctx, impl_->embs.size() * sizeof(float), &impl_->d_embs) != MOT_OK) throw std::runtime_error("motcpp_amd: device allocation failed");
  }
}
DeviceLifecycleBatch::~DeviceLifecycleBatch() = default;
void DeviceLifecycleBatch::reset() {
  std::lock_guard<std::mutex> dev_lock(impl_->dev->frame_mu);
  int rc = MOT_OK;
  if (impl_->bt) rc = mot_bt_reset(impl_->bt);
  if (impl_->so) rc = mot_sort_reset(impl_->so);
  if (impl_->oc) rc = mot_oc_reset(impl_->oc);
  if (impl_->bot) rc = mot_bot_reset(impl_->bot);
  if (rc != MOT_OK) throw std::runtime_error(mot_ctx_last_error(impl_->dev->ctx));
}
std::vector<Eigen::MatrixXf> DeviceLifecycleBatch::update(const std::vector<Eigen::MatrixXf>& dets, const std::vector<Eigen::MatrixXf>& embs,
                                                          const std::vector<Eigen::MatrixXf>& warps) {
  if (static_cast<int>(dets.size()) != n_) throw std::invalid_argument("device-lifecycle batch: one detection matrix per stream");
  Impl& I = *impl_;
  const bool use_embs = I.kind == 3 && I.emb_dim > 0 && !embs.empty();
  const bool use_warps = I.kind == 3 && !warps.empty();
  if ((!embs.empty() || !warps.empty()) && I.kind != 3) throw std::invalid_argument("device-lifecycle batch: only BoT-SORT takes embeddings / camera-motion warps");
  if (use_embs && static_cast<int>(embs.size()) != n_) throw std::invalid_argument("device-lifecycle batch: one embedding matrix per stream");
  if (use_warps && static_cast<int>(warps.size()) != n_) throw std::invalid_argument("device-lifecycle batch: one warp (or an empty matrix) per stream");
  for (int s = 0; s < n_; ++s) {
    const Eigen::MatrixXf& d = dets[s];
    const int n = static_cast<int>(d.rows());
    if (n > 0 && d.cols() != 6) throw std::invalid_argument("device-lifecycle batch: detections must be N x 6");
    if (n > maxd_) throw std::invalid_argument("device-lifecycle batch: more detections than max_dets");
    I.counts[s] = n;
    float* dst = I.soa.data() + static_cast<size_t>(s) * 6 * maxd_;
    for (int k = 0; k < 6; ++k)  // a column-major N x 6 matrix IS the SoA layout
      if (n > 0) std::memcpy(dst + static_cast<size_t>(k) * maxd_, d.data() + static_cast<size_t>(k) * n, sizeof(float) * n);
    if (use_embs) {
      const Eigen::MatrixXf& e = embs[s];
      if (n > 0 && (e.rows() != n || e.cols() != I.emb_dim)) throw std::invalid_argument("device-lifecycle batch: embeddings must be N x emb_dim, one row per detection");
      float* ed = I.embs.data() + static_cast<size_t>(s) * maxd_ * I.emb_dim;
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < I.emb_dim; ++k) ed[static_cast<size_t>(i) * I.emb_dim + k] = e(i, k);
    }
  }
  if (use_warps) {
    I.warps.assign(static_cast<size_t>(n_) * 6, 0.f);
    I.has_warp.assign(n_, 0);
    for (int s = 0; s < n_; ++s) {
      const Eigen::MatrixXf& w = warps[s];
      if (w.rows() == 0 && w.cols() == 0) continue;
      if (w.rows() != 2 || w.cols() != 3) throw std::invalid_argument("device-lifecycle batch: a camera-motion warp is 2 x 3");
      for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) I.warps[static_cast<size_t>(s) * 6 + r * 3 + c] = w(r, c);
      I.has_warp[s] = 1;
    }
  }
  std::lock_guard<std::mutex> dev_lock(I.dev->frame_mu);  // the context's stream is shared with the other trackers of this GPU
  mot_ctx* ctx = I.dev->ctx;
  if (mot_memcpy_h2d(ctx, I.d_dets, I.soa.data(), I.soa.size() * sizeof(float)) != MOT_OK) throw std::runtime_error(mot_ctx_last_error(ctx));
  if (use_embs && mot_memcpy_h2d(ctx, I.d_embs, I.embs.data(), I.embs.size() * sizeof(float)) != MOT_OK) throw std::runtime_error(mot_ctx_last_error(ctx));
  const int cap_out = cap_;  // a stream never reports more rows than it has tracks
  I.out.resize(static_cast<size_t>(n_) * cap_out * 8);
  const float* dd = static_cast<const float*>(I.d_dets);
  int rc = MOT_OK, total = 0;
  const bool packed = I.kind >= 2;
  if (I.kind == 0) rc = mot_bt_step(I.bt, dd, I.counts.data(), I.out.data(), I.out_counts.data(), cap_out);
  else if (I.kind == 1) rc = mot_sort_step(I.so, dd, I.counts.data(), I.out.data(), I.out_counts.data(), cap_out);
  else if (I.kind == 2) rc = mot_oc_step_packed(I.oc, dd, I.counts.data(), I.out.data(), n_ * cap_out, I.out_counts.data(), &total);
  else rc = mot_bot_step_packed(I.bot, dd, I.counts.data(), use_embs ? static_cast<const float*>(I.d_embs) : nullptr,
                                use_warps ? I.warps.data() : nullptr, use_warps ? I.has_warp.data() : nullptr, I.out.data(), n_ * cap_out,
                                I.out_counts.data(), &total);
  if (rc != MOT_OK) throw std::runtime_error(std::string("motcpp_amd: device-lifecycle step failed: ") + mot_ctx_last_error(ctx));
  std::vector<Eigen::MatrixXf> res;
  res.reserve(n_);
  size_t off = 0;
  for (int s = 0; s < n_; ++s) {
    const int m = I.out_counts[s];
    Eigen::MatrixXf t(m, 8);
    const float* rows = packed ? I.out.data() + off * 8 : I.out.data() + static_cast<size_t>(s) * cap_out * 8;
    for (int i = 0; i < m; ++i)
      for (int k = 0; k < 8; ++k) t(i, k) = rows[static_cast<size_t>(i) * 8 + k];
    off += static_cast<size_t>(m);
    res.push_back(std::move(t));
  }
  return res;
}
namespace {
struct P5 { float v[5]; };
struct P14 { float v[14]; };
}
ByteTrackDeviceBatch::ByteTrackDeviceBatch(int nstreams, int cap_tracks, int max_dets, float min_conf, float track_thresh, float match_thresh,
                                           int track_buffer, int frame_rate, int device_index)
    : DeviceLifecycleBatch(0, nstreams, cap_tracks, max_dets,
                           P5{{min_conf, track_thresh, match_thresh, static_cast<float>(track_buffer), static_cast<float>(frame_rate)}}.v, device_index) {}
SortDeviceBatch::SortDeviceBatch(int nstreams, int cap_tracks, int max_dets, float det_thresh, int max_age, int min_hits, float iou_threshold,
                                 int device_index)
    : DeviceLifecycleBatch(1, nstreams, cap_tracks, max_dets,
                           P5{{det_thresh, static_cast<float>(max_age), 50.f, static_cast<float>(min_hits), iou_threshold}}.v, device_index) {}
namespace {
float asso_or_throw(const std::string& name) {
  const int k = rt::asso_kind(name);
  if (k < 0) throw std::invalid_argument("Invalid or unsupported association function: " + name);
  return static_cast<float>(k);
}
}  // namespace
OCSortDeviceBatch::OCSortDeviceBatch(int nstreams, int cap_tracks, int max_dets, float det_thresh, int max_age, int min_hits, float iou_threshold,
                                     float min_conf, int delta_t, float inertia, bool use_byte, float q_xy, float q_s, const std::string& asso_func,
                                     int frame_width, int frame_height, int device_index)
    : DeviceLifecycleBatch(2, nstreams, cap_tracks, max_dets,
                           P14{{det_thresh, static_cast<float>(max_age), 50.f, static_cast<float>(min_hits), iou_threshold, min_conf,
                                static_cast<float>(delta_t), inertia, use_byte ? 1.f : 0.f, q_xy, q_s, asso_or_throw(asso_func),
                                static_cast<float>(frame_width), static_cast<float>(frame_height)}}.v, device_index) {}
BotSortDeviceBatch::BotSortDeviceBatch(int nstreams, int cap_tracks, int max_dets, int emb_dim, float hi, float lo, float newt, int track_buffer,
                                       float match, float prox, float app, int frame_rate, bool fuse_first, bool with_reid, int device_index)
    : DeviceLifecycleBatch(3, nstreams, cap_tracks, max_dets,
                           P14{{hi, lo, newt, static_cast<float>(track_buffer), match, prox, app, static_cast<float>(frame_rate),
                                fuse_first ? 1.f : 0.f, with_reid ? 1.f : 0.f, 0.f, 0.f, 0.f, 0.f}}.v, device_index, emb_dim) {}

// ---- utils:: primitive seam ----------------------------------------------------------------------
namespace utils {
namespace {
std::vector<float> row_major(const Eigen::MatrixXf& m, int cols) {
  std::vector<float> v(static_cast<size_t>(m.rows()) * cols);
  for (Eigen::Index i = 0; i < m.rows(); ++i)
    for (int k = 0; k < cols; ++k) v[static_cast<size_t>(i) * cols + k] = m(i, k);
  return v;
}
void chk(rt::Device& d, int rc, const char* what) { d.check(rc, what); }
}  // namespace

LinearAssignmentResult linear_assignment(const Eigen::MatrixXf& cost, float thresh, int device_index) {
  LinearAssignmentResult r;
  const int n = static_cast<int>(cost.rows()), m = static_cast<int>(cost.cols());
  if (n == 0 || m == 0) {
    for (int i = 0; i < n; ++i) r.unmatched_a.push_back(i);
    for (int j = 0; j < m; ++j) r.unmatched_b.push_back(j);
    return r;
  }
  auto dev = rt::Device::shared(device_index);
  std::lock_guard<std::mutex> dev_lock(dev->frame_mu);  // the context's stream and scratch are shared with the trackers on this GPU
  std::vector<float> c = row_major(cost, m);
  std::vector<int> x(n), y(m);
  chk(*dev, mot_lap_solve_host(dev->ctx, c.data(), n, m, thresh, MOT_LAP_PLAIN, nullptr, 0.f, x.data(), y.data(), nullptr), "mot_lap_solve_host");
  for (int i = 0; i < n; ++i) {
    if (x[i] < 0) r.unmatched_a.push_back(i);
    else r.matches.push_back({i, x[i]});
  }
  for (int j = 0; j < m; ++j) if (y[j] < 0) r.unmatched_b.push_back(j);
  return r;
}
static Eigen::MatrixXf iou_mode(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int mode, int device_index,
                                int assoc = MOT_ASSOC_IOU, int fw = 1, int fh = 1) {
  const int n = static_cast<int>(a.rows()), m = static_cast<int>(b.rows());
  Eigen::MatrixXf out(n, m);
  if (n == 0 || m == 0) { out.setZero(); return out; }  // Zero(N, M), iou.hpp:68-70,127-129
  auto dev = rt::Device::shared(device_index);
  std::lock_guard<std::mutex> dev_lock(dev->frame_mu);  // the context's stream and scratch are shared with the trackers on this GPU
  std::vector<float> ra = row_major(a, 4), rb = row_major(b, 4), c(static_cast<size_t>(n) * m);
  chk(*dev, mot_assoc_cost_host(dev->ctx, ra.data(), n, rb.data(), m, nullptr, mode, assoc, fw, fh, c.data()), "mot_assoc_cost_host");
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) out(i, j) = c[static_cast<size_t>(i) * m + j];
  return out;
}
Eigen::MatrixXf iou_batch(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int device_index) { return iou_mode(a, b, MOT_COST_IOU, device_index); }
Eigen::MatrixXf iou_distance(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int device_index) { return iou_mode(a, b, MOT_COST_IOU_DIST, device_index); }
Eigen::MatrixXf hmiou_batch(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int d) { return iou_mode(a, b, MOT_COST_IOU, d, MOT_ASSOC_HMIOU); }
Eigen::MatrixXf giou_batch(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int d) { return iou_mode(a, b, MOT_COST_IOU, d, MOT_ASSOC_GIOU); }
Eigen::MatrixXf ciou_batch(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int d) { return iou_mode(a, b, MOT_COST_IOU, d, MOT_ASSOC_CIOU); }
Eigen::MatrixXf diou_batch(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int d) { return iou_mode(a, b, MOT_COST_IOU, d, MOT_ASSOC_DIOU); }
Eigen::MatrixXf centroid_batch(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, int w, int h, int d) {
  return iou_mode(a, b, MOT_COST_IOU, d, MOT_ASSOC_CENTROID, w, h);
}
AssociationFunction::AssociationFunction(int w, int h, const std::string& asso_mode, int device_index)
    : frame_width_(w), frame_height_(h), kind_(rt::asso_kind(asso_mode)), device_(device_index) {
  if (kind_ < 0) {
    if (asso_mode == "iou_obb" || asso_mode == "centroid_obb")
      throw std::invalid_argument("motcpp_amd: oriented-box association mode '" + asso_mode + "' is out of scope");
    throw std::invalid_argument("Invalid association mode: " + asso_mode);
  }
}
Eigen::MatrixXf AssociationFunction::operator()(const Eigen::MatrixXf& a, const Eigen::MatrixXf& b) const {
  return iou_mode(a, b, MOT_COST_IOU, device_, kind_, frame_width_, frame_height_);
}
Eigen::MatrixXf embedding_distance(const Eigen::MatrixXf& t, const Eigen::MatrixXf& d, const std::string& metric, int device_index) {
  // "cosine" (matching.cpp:79-92) on the fp32 matrix cores, "euclidean" (matching.cpp:93-101) on the vector ALUs
  const int metric_id = (metric == "cosine") ? MOT_EMB_COSINE : ((metric == "euclidean") ? MOT_EMB_EUCLIDEAN : -1);
  if (metric_id < 0) throw std::invalid_argument("Unknown metric: " + metric);
  const int n = static_cast<int>(t.rows()), m = static_cast<int>(d.rows()), dim = static_cast<int>(t.cols());
  Eigen::MatrixXf out(n, m);
  if (n == 0 || m == 0) return out;
  auto dev = rt::Device::shared(device_index);
  std::lock_guard<std::mutex> dev_lock(dev->frame_mu);  // the context's stream and scratch are shared with the trackers on this GPU
  std::vector<float> ra = row_major(t, dim), rb = row_major(d, dim), c(static_cast<size_t>(n) * m);
  chk(*dev, mot_embedding_cost_host(dev->ctx, metric_id, ra.data(), n, rb.data(), m, dim, c.data()), "mot_embedding_cost_host");
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) out(i, j) = c[static_cast<size_t>(i) * m + j];
  return out;
}
Eigen::MatrixXf fuse_iou(const Eigen::MatrixXf& reid, const Eigen::MatrixXf& a, const Eigen::MatrixXf& b, const Eigen::MatrixXf&, int device_index) {
  const int n = static_cast<int>(reid.rows()), m = static_cast<int>(reid.cols());
  if (n == 0 || m == 0) return reid;  // matching.cpp:113-115
  if (a.rows() != n || b.rows() != m || a.cols() < 4 || b.cols() < 4) throw std::invalid_argument("fuse_iou: boxes do not match the cost matrix");
  auto dev = rt::Device::shared(device_index);
  std::lock_guard<std::mutex> dev_lock(dev->frame_mu);
  std::vector<float> rr = row_major(reid, m), ra = row_major(a, 4), rb = row_major(b, 4), c(static_cast<size_t>(n) * m);
  chk(*dev, mot_fuse_iou_host(dev->ctx, rr.data(), ra.data(), n, rb.data(), m, c.data()), "mot_fuse_iou_host");
  Eigen::MatrixXf out(n, m);
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) out(i, j) = c[static_cast<size_t>(i) * m + j];
  return out;
}

namespace {
Eigen::MatrixXf gate_call(const std::string& filter, int mode, const Eigen::MatrixXf* cost, const Eigen::MatrixXf& means,
                          const Eigen::MatrixXf& covs, const Eigen::MatrixXf& meas, bool only_position, int metric, float lambda,
                          float gated_cost, int device_index) {
  const int kind = (filter == "xyah") ? MOT_KF_XYAH : ((filter == "xywh") ? MOT_KF_XYWH : -1);
  if (kind < 0) throw std::invalid_argument("gating: unknown filter '" + filter + "' (xyah | xywh)");
  const int n = static_cast<int>(means.rows()), m = static_cast<int>(meas.rows());
  if (means.cols() != 8 || covs.rows() != n || covs.cols() != 64 || (m && meas.cols() < 4))
    throw std::invalid_argument("gating: means n x 8, covariances n x 64, measurements m x 4 expected");
  if (cost && (cost->rows() != n || cost->cols() != m)) throw std::invalid_argument("gating: cost matrix is not tracks x measurements");
  Eigen::MatrixXf out(n, m);
  if (n == 0 || m == 0) return out;
  auto dev = rt::Device::shared(device_index);
  std::lock_guard<std::mutex> dev_lock(dev->frame_mu);
  std::vector<float> rm = row_major(means, 8), rc = row_major(covs, 64), rz = row_major(meas, 4), c(static_cast<size_t>(n) * m), ci;
  if (cost) ci = row_major(*cost, m);
  chk(*dev, mot_gate_cost_host(dev->ctx, kind, mode, n, m, rm.data(), rc.data(), rz.data(), cost ? ci.data() : nullptr, only_position ? 1 : 0,
                               metric, lambda, gated_cost, c.data()), "mot_gate_cost_host");
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) out(i, j) = c[static_cast<size_t>(i) * m + j];
  return out;
}
}  // namespace

Eigen::MatrixXf gating_distance(const std::string& filter, const Eigen::MatrixXf& means, const Eigen::MatrixXf& covs, const Eigen::MatrixXf& meas,
                                bool only_position, const std::string& metric, int device_index) {
  const int metric_id = (metric == "maha") ? 0 : ((metric == "gaussian") ? 1 : -1);
  if (metric_id < 0) throw std::invalid_argument("Invalid metric: " + metric);  // kalman_filter.cpp:172-174
  return gate_call(filter, MOT_GATE_DISTANCE, nullptr, means, covs, meas, only_position, metric_id, 0.f, 0.f, device_index);
}
Eigen::MatrixXf fuse_motion(const std::string& filter, const Eigen::MatrixXf& cost, const Eigen::MatrixXf& means, const Eigen::MatrixXf& covs,
                            const Eigen::MatrixXf& meas, bool only_position, float lambda, int device_index) {
  if (cost.rows() == 0 || cost.cols() == 0) return cost;  // matching.hpp:67-69
  return gate_call(filter, MOT_GATE_FUSE_MOTION, &cost, means, covs, meas, only_position, 0, lambda, 0.f, device_index);
}
Eigen::MatrixXf gate_cost_matrix(const std::string& filter, const Eigen::MatrixXf& cost, const Eigen::MatrixXf& means, const Eigen::MatrixXf& covs,
                                 const Eigen::MatrixXf& meas, float mc_lambda, float gated_cost, bool only_position, int device_index) {
  return gate_call(filter, MOT_GATE_STRONGSORT, &cost, means, covs, meas, only_position, 0, mc_lambda, gated_cost, device_index);
}
}  // namespace utils

}  // namespace motcpp
