// Common base of the MI355X-backed tracker classes: BaseTracker's contract on top of a stage machine
// (motcpp::rt::Staged) whose numeric work runs in HIP kernels behind the C ABI of motcpp_amd.h.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "tracker.hpp"

namespace motcpp {
namespace rt {
class Staged;
class Device;
class PooledStream;
}  // namespace rt

class DeviceTracker : public BaseTracker {
 public:
  ~DeviceTracker() override;
  Eigen::MatrixXf update(const Eigen::MatrixXf& dets, const cv::Mat& img,
                         const Eigen::MatrixXf& embs = Eigen::MatrixXf()) override;
  void reset() override;
  rt::Staged* staged() const { return impl_.get(); }        // the host stage machine (nullptr for a pooled tracker)
  rt::PooledStream* pooled() const { return pooled_.get(); }  // the pooled device stream (nullptr for a host-lifecycle tracker)
  // parity hook (pooled trackers): ids, Kalman means [n][d] and covariances [n][d*d] of the live tracks in list order; returns n
  int dump_states(std::vector<int>* ids, std::vector<float>* mean, std::vector<float>* cov);
  const std::shared_ptr<rt::Device>& device() const { return dev_; }
  // the checks and bookkeeping update() does before any track is touched (check_inputs, the asso_func error, detection
  // format, frame counter); false = this frame is skipped. StreamBatch calls it per stream. Throws what update() throws.
  bool prepare_update(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs);
  // the two halves of prepare_update: everything that can throw (no side effect), then the bookkeeping (cannot throw). StreamBatch
  // validates EVERY stream before it commits any, so that an exception leaves no tracker a frame ahead of its device state.
  void validate_update(const Eigen::MatrixXf& dets, const cv::Mat& img, const Eigen::MatrixXf& embs) const;
  bool commit_update(const Eigen::MatrixXf& dets, const cv::Mat& img);
  // Threading: like the reference, a tracker instance is not re-entrant, and different instances are meant to be updated from
  // different host threads (one tracker per camera thread, include/motcpp/tracker.hpp:67-69 of the reference). Since round 4
  // Sort / ByteTrack / OCSort / BotSort objects are streams of shared device-lifecycle batches (host/pool.hpp): update() calls
  // that arrive together from different threads are merged into ONE launch sequence on the GPU and each caller gets its own
  // table back — T threads with one tracker each run at the batched rate, not one frame at a time. MOTCPP_LIFECYCLE=host keeps
  // the per-object host stage machines of rounds 1-3 (DeepOCSort always uses one); their frames serialise on a per-GPU mutex.

 protected:
  DeviceTracker(float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold, bool per_class,
                int nr_classes, const std::string& asso_func, bool is_obb, int device_index);
  void adopt(rt::Staged* impl);
  void adopt_pooled(int c_kind, const std::vector<float>& c_params);  // kind / parameter vector as in motcpp_c.h
  bool camera_motion(const float* warp2x3);
  bool validate_inputs_ = true;  // SORT does not call check_inputs (sort.cpp:102-110)
  bool skip_empty_ = false;      // BoT-SORT returns before touching any state when dets is empty (botsort.cpp:267-269)
  std::shared_ptr<rt::Device> dev_;
  std::string asso_error_;       // OC-SORT with an unknown asso_func: update() throws this (ocsort.cpp:413 -> iou.hpp:405-407)

 private:
  std::unique_ptr<rt::Staged> impl_;
  std::unique_ptr<rt::PooledStream> pooled_;
};

// Lock-step driver for many independent streams on one GPU: every stage of every tracker is batched into
// one kernel launch per kernel family (the data-parallel axis of the hot path).
class StreamBatch {
 public:
  explicit StreamBatch(std::vector<DeviceTracker*> trackers);
  // dets[s]: N_s x 6 column-major; returns per-stream M_s x 8 tables
  std::vector<Eigen::MatrixXf> update(const std::vector<Eigen::MatrixXf>& dets, const cv::Mat& img,
                                      const std::vector<Eigen::MatrixXf>& embs = {});
  size_t size() const { return trackers_.size(); }

 private:
  std::vector<DeviceTracker*> trackers_;
};

// S independent ByteTrack streams whose whole update() runs on the GPU (C ABI: mot_bt_*, motcpp_amd/csrc/bt_device.hip):
// per frame the host enqueues a fixed sequence of launches and copies the output tables back — no per-stream host work.
// Same semantics per stream as trackers::ByteTrack(…).update(dets, img) with the default BaseTracker arguments; ids are
// per stream. cap_tracks bounds tracked + lost tracks of a stream, max_dets the detections of a frame (std::runtime_error
// when a stream exceeds them).
class DeviceLifecycleBatch {
 public:
  virtual ~DeviceLifecycleBatch();
  DeviceLifecycleBatch(const DeviceLifecycleBatch&) = delete;
  DeviceLifecycleBatch& operator=(const DeviceLifecycleBatch&) = delete;
  // dets[s]: N_s x 6 column-major [x1,y1,x2,y2,conf,cls]; returns per-stream M_s x 8 tables [x1,y1,x2,y2,id,conf,cls,det_ind].
  // BoT-SORT batches also take embs[s] (N_s x emb_dim, one row per detection; empty vector: no features this frame) and
  // warps[s] (2 x 3 camera-motion warp of stream s for this frame, or a 0 x 0 matrix for none; empty vector: none at all).
  std::vector<Eigen::MatrixXf> update(const std::vector<Eigen::MatrixXf>& dets, const std::vector<Eigen::MatrixXf>& embs = {},
                                      const std::vector<Eigen::MatrixXf>& warps = {});
  void reset();
  size_t size() const { return static_cast<size_t>(n_); }

 protected:
  DeviceLifecycleBatch(int kind, int nstreams, int cap_tracks, int max_dets, const float* params, int device_index, int emb_dim = 0);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
  int n_, cap_, maxd_;
};

class ByteTrackDeviceBatch : public DeviceLifecycleBatch {
 public:
  ByteTrackDeviceBatch(int nstreams, int cap_tracks, int max_dets, float min_conf = 0.1f, float track_thresh = 0.45f,
                       float match_thresh = 0.8f, int track_buffer = 25, int frame_rate = 30, int device_index = 0);
};

// The same for trackers::Sort (C ABI: mot_sort_*, motcpp_amd/csrc/sort_device.hip); reset() keeps the ids counting (sort.cpp:97-100)
class SortDeviceBatch : public DeviceLifecycleBatch {
 public:
  SortDeviceBatch(int nstreams, int cap_tracks, int max_dets, float det_thresh = 0.3f, int max_age = 1, int min_hits = 3,
                  float iou_threshold = 0.3f, int device_index = 0);
};

// trackers::OCSort on the device (C ABI: mot_oc_*, motcpp_amd/csrc/oc_device.hip); asso_func as in OCSort ("iou", "hmiou", "giou", ...)
class OCSortDeviceBatch : public DeviceLifecycleBatch {
 public:
  OCSortDeviceBatch(int nstreams, int cap_tracks, int max_dets, float det_thresh = 0.2f, int max_age = 30, int min_hits = 3,
                    float iou_threshold = 0.3f, float min_conf = 0.1f, int delta_t = 3, float inertia = 0.2f, bool use_byte = false,
                    float Q_xy_scaling = 0.01f, float Q_s_scaling = 0.0001f, const std::string& asso_func = "iou", int frame_width = 1920,
                    int frame_height = 1080, int device_index = 0);
};

// trackers::BotSort on the device (C ABI: mot_bot_*, motcpp_amd/csrc/bot_device.hip); emb_dim = 0: no appearance features
class BotSortDeviceBatch : public DeviceLifecycleBatch {
 public:
  BotSortDeviceBatch(int nstreams, int cap_tracks, int max_dets, int emb_dim, float track_high_thresh = 0.5f, float track_low_thresh = 0.1f,
                     float new_track_thresh = 0.6f, int track_buffer = 30, float match_thresh = 0.8f, float proximity_thresh = 0.5f,
                     float appearance_thresh = 0.25f, int frame_rate = 30, bool fuse_first_associate = false, bool with_reid = true,
                     int device_index = 0);
};

}  // namespace motcpp
