// Pooled tracker streams: what makes the reference's OWN surface — one BaseTracker object per camera, each updated from its own
// host thread (include/motcpp/tracker.hpp:67-69, docs/guides/architecture.md:242-255) — run as batched GPU work.
//
// Every tracker object owns one STREAM of a device-lifecycle batch (mot_bt_* / mot_sort_* / mot_oc_* / mot_bot_*: the whole update()
// as a fixed launch sequence for all streams of the batch). Objects with the same parameters on the same GPU share batches
// ("segments"); update() calls that arrive together are merged by a combiner — the first caller becomes the leader of a ROUND,
// gives the others a short window to join, and runs ONE launch sequence in which the streams that did not call sit the frame out
// (mot_*_enqueue_frame) — and every caller gets its own table back from page-locked memory the kernels wrote directly
// (mot_*_collect_view). While a round runs, the next one fills: callers copy their detections into the next round's page-locked
// staging buffer in parallel, so the leader's serial part is one host-to-device copy and the launches.
//
// Capacities (cap_tracks, max_dets of a batch) are fixed when it is created; the BaseTracker surface has none. Segments therefore
// come in LEVELS (512 x 256, 2048 x 1024, 8192 x 4096, 32768 x 16384 tracks x detections); an object starts on the level its first
// frame needs and MOVES up (mot_*_move_stream) before a frame that could overflow its level — live tracks after the last frame plus
// the births the new detections can cause — so the device never sees MOT_ERR_CAPACITY.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "motcpp_amd.h"

namespace motcpp::rt {

enum PoolKind { kPoolByteTrack = 0, kPoolSort = 1, kPoolOCSort = 2, kPoolBotSort = 3 };

class Segment;
class StreamPool;

struct PoolStats {       // per process, all pools (bench / tests)
  long rounds = 0;       // launch sequences run
  long frames = 0;       // stream-frames processed
  long moves = 0;        // streams moved to a larger level
  long max_round = 0;    // most streams merged into one round
  // leader wall time, microseconds: the batching window, waiting for the callers' copies / the previous table's readers, the round itself
  // (uploads + launches + wait for the GPU), and inside that the host time of queueing the launch sequence
  double us_window = 0, us_gather = 0, us_run = 0, us_enqueue = 0;
};
PoolStats pool_stats(bool reset);
// MOTCPP_LIFECYCLE=host: the tracker classes keep their lifecycle in host stage machines (rounds 1-3: one frame at a time per GPU)
bool pooling_enabled();
// motcpp_c.h's parameter vector of tracker kind c_kind (0 SORT, 1 ByteTrack, 2 OC-SORT, 3 BoT-SORT; missing tail = reference defaults)
// -> the PoolKind and the device lifecycle's parameter vector
int pooled_params(int c_kind, const float* p, int np, std::vector<float>* out);

struct PooledFrame {
  const float* dets = nullptr;  // column-major n x 6 (leading dimension ld)
  int n = 0, ld = 0;
  const float* embs = nullptr;  // n x emb_dim: column-major with leading dimension emb_ld, or row-major (emb_ld = row stride); nullptr: none
  int emb_ld = 0, emb_dim = 0;
  bool embs_rowmajor = false;
  int img_w = 0, img_h = 0;
};

class PooledStream {
 public:
  // params: the device lifecycle's parameter vector (mot_bt_create: 5, mot_sort_create: 5, mot_oc_create: 14 — frame size filled in at
  // the first update —, mot_bot_create: 10)
  PooledStream(int device, int kind, const float* params, int nparams);
  ~PooledStream();
  PooledStream(const PooledStream&) = delete;
  PooledStream& operator=(const PooledStream&) = delete;

  // One frame; returns the number of output rows, `rows` points at them (row-major [m][8]) in memory owned by this object that stays
  // valid until its next call.
  int update(const PooledFrame& f, const float** rows);
  // The same for k objects at once from ONE host thread (motcpp::StreamBatch): the ones that share a segment join the same round.
  static void update_many(PooledStream* const* streams, const PooledFrame* frames, int k, const float** rows, int* counts);
  void reset();
  void set_camera_motion(const float* warp2x3);  // BoT-SORT: the warp of the NEXT frame (nullptr withdraws it)
  // parity hooks: ids / Kalman states (and BoT-SORT's smooth features) of the live tracks in list order; returns the track count
  int dump(std::vector<int>* ids, std::vector<float>* mean, std::vector<float>* cov, std::vector<float>* feats, std::vector<unsigned char>* has_feat);
  int state_dim() const { return (kind_ == kPoolSort || kind_ == kPoolOCSort) ? 7 : 8; }
  int level() const;

 private:
  void attach(int level, int emb_dim);
  void prepare(const PooledFrame& f, void* req);
  void unprepare(void* req);
  int finish(void* req, const float** rows);
  int device_, kind_;
  std::vector<float> params_;
  std::shared_ptr<StreamPool> pool_;
  Segment* seg_ = nullptr;
  int s_ = -1;
  int alive_ = 0;          // live tracks after the last frame
  bool fresh_ = true;      // the slot still needs its reset before the first frame
  bool reset_pending_ = false;  // reset() was called: the stream's tracks go at the head of its next round
  bool have_warp_ = false;
  float warp_[6] = {0, 0, 0, 0, 0, 0};
  std::vector<float> rows_;  // this object's rows of the last frame (copied out of the round's page-locked table)
};

}  // namespace motcpp::rt
