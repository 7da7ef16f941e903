// Stage-machine interface shared by the four tracker implementations and the frame drivers.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "runtime.hpp"
#include "team.hpp"

namespace motcpp::rt {

struct FrameIn {
  const float* dets = nullptr;  // column-major n x 6 (ld = leading dimension)
  int n = 0, ld = 0;
  const float* embs = nullptr;  // column-major n x emb_dim (ld = emb_ld) or nullptr
  int emb_ld = 0, emb_dim = 0;
  bool embs_rowmajor = false;   // true: n x emb_dim row-major (ld = emb_dim)
  int img_w = 0, img_h = 0;
  const float* d_dets = nullptr;  // optional: the same detections already resident in HBM, SoA [6][d_ld]
  int d_ld = 0;
  const float* d_embs = nullptr;  // optional: the embeddings already resident in HBM, row-major n x emb_dim (nothing is uploaded)
};

// Set of small non-negative ints (track ids) with O(1) clear: a generation stamp per id, no hashing, no allocation
// in steady state. The tracker lifecycles build several id sets per frame; std::unordered_set dominated host time.
class IdSet {
 public:
  void clear() { ++gen_; }
  bool insert(int id) {  // true if newly inserted
    if (id >= static_cast<int>(stamp_.size())) stamp_.resize(static_cast<size_t>(id) * 2 + 64, 0);
    if (stamp_[id] == gen_) return false;
    stamp_[id] = gen_;
    return true;
  }
  bool count(int id) const { return id < static_cast<int>(stamp_.size()) && stamp_[id] == gen_; }
 private:
  std::vector<int> stamp_;
  int gen_ = 1;
};

struct LapRecord {
  std::vector<int> x, y;
};

class Staged {
 public:
  virtual ~Staged() = default;
  virtual void begin(const FrameIn& in) = 0;  // queue the first stage of a frame
  virtual bool advance() = 0;                 // consume the flushed stage, queue the next; false = frame finished
  virtual void reset() = 0;
  virtual Core& core() = 0;
  // Camera-motion warp (2x3 row-major, what the reference's cmc_->apply(img, dets) returns) for the NEXT frame only;
  // nullptr withdraws it. false: this tracker has no camera-motion step (only BoT-SORT has, botsort.cpp:317-324).
  virtual bool set_camera_motion(const float* /*warp2x3*/) { return false; }
  const std::vector<float>& rows() const { return rows_; }  // output rows [x1,y1,x2,y2,id,conf,cls,det_ind]
  const std::vector<LapRecord>& laps() const { return laps_; }
  bool record_laps = true;  // parity hook; switched off for throughput runs
  // parity hook: slots of the live tracks in list order with their ids (states are read back by the caller)
  virtual void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const = 0;
  // parity hook (BoT-SORT): device address of the smooth-feature slab [slot][dim] (nullptr: none), which tracks have one
  virtual const float* feature_slab(int* dim, std::vector<char>* has) const { *dim = 0; (void)has; return nullptr; }
  // parity hook (UCMCTrack): rows of doubles [id, state, death, birth, det_idx, age, x(4), P(16)] in list order; false: not that tracker
  virtual bool f64_states(std::vector<double>* rows) { (void)rows; return false; }
  // parity hook for trackers whose states are not records of the Core slab (HybridSORT): rows of `width` floats [id, x, P]
  virtual bool f32_states(std::vector<float>* rows, int* width) { (void)rows; (void)width; return false; }

 protected:
  void record(const Core::Lap& l) {
    if (!record_laps) return;
    LapRecord r;
    r.x.assign(l.x.h, l.x.h + l.n);
    r.y.assign(l.y.h, l.y.h + l.m);
    laps_.push_back(std::move(r));
  }
  void push_row(const float* box4, int ld, int col, int id, float conf, int cls, int det_ind) {
    rows_.push_back(box4[col]);
    rows_.push_back(box4[static_cast<size_t>(ld) + col]);
    rows_.push_back(box4[static_cast<size_t>(2) * ld + col]);
    rows_.push_back(box4[static_cast<size_t>(3) * ld + col]);
    rows_.push_back(static_cast<float>(id));
    rows_.push_back(conf);
    rows_.push_back(static_cast<float>(cls));
    rows_.push_back(static_cast<float>(det_ind));
  }
  std::vector<float> rows_;
  std::vector<LapRecord> laps_;
};

// Runs one frame for a set of trackers sharing a Device in lockstep: one kernel launch per kernel family per stage.
// errors == nullptr: the first failure of any stream aborts the frame for all (an exception; StreamBatch's contract: one call, one result).
// errors != nullptr ([count] strings, empty on entry): a stream whose begin() / advance() throws sits the rest of the frame out with its own
// message in errors[i]; the other streams finish their frame (merged update() calls of unrelated tracker objects: the reference's exceptions
// are per object, src/tracker.cpp:108-125). A failure of the device itself (a flush) is everybody's either way.
void run_frame(Device& dev, Staged* const* trackers, const FrameIn* inputs, int count, Team* team = nullptr, std::string* errors = nullptr);

// update() calls of host-lifecycle tracker OBJECTS that arrive together from different host threads (round 5: DeepOCSort, StrongSORT,
// UCMCTrack, BoostTrack, HybridSort — the trackers without a device lifecycle): the first caller of a round leads it and steps every
// joined stage machine in ONE run_frame (3-5 flushes per frame however many cameras), the others sleep until their rows are there.
// Before, every call took the Device's frame mutex for its whole update(): T threads ran one frame at a time. Same combiner as the
// pooled device streams (host/pool.cpp): futex rounds, a window of at most 100 us when the GPU was idle, group commit otherwise.
void run_frame_combined(const std::shared_ptr<Device>& dev, Staged* tracker, const FrameIn& input);

// factories (parameter vectors: same layout as documented in include/motcpp_c.h)
Staged* make_sort(std::shared_ptr<Device>, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold);
Staged* make_bytetrack(std::shared_ptr<Device>, float min_conf, float track_thresh, float match_thresh, int track_buffer,
                       int frame_rate, int max_age, int max_obs);
Staged* make_ocsort(std::shared_ptr<Device>, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold,
                    float min_conf, int delta_t, float inertia, bool use_byte, float q_xy, float q_s, int asso = 0);
// "iou" | "hmiou" | "giou" | "ciou" | "diou" | "centroid" -> mot_assoc, or -1 (AssociationFunction::get_asso_func, iou.hpp:385-408;
// the oriented-box modes are out of scope)
int asso_kind(const std::string& name);
Staged* make_deepocsort(std::shared_ptr<Device>, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold,
                        int delta_t, float inertia, float w_emb, float alpha_fixed, float aw_param, bool emb_off, bool cmc_off,
                        bool aw_off, float q_xy, float q_s, int asso);
// StrongSORT (src/trackers/strongsort.cpp); nn_budget > 0
Staged* make_strongsort(std::shared_ptr<Device>, float min_conf, float max_cos_dist, float max_iou_dist, int n_init, int nn_budget,
                        float mc_lambda, float ema_alpha, int max_age);
// UCMCTrack (src/trackers/ucmc.cpp): the reference's double-precision parameters; Ki 3 x 4 / Ko 4 x 4 row-major when has_camera
struct UcmcParams {
  float det_thresh = 0.3f;
  int max_age = 30;
  double a1 = 100.0, a2 = 100.0, wx = 5.0, wy = 5.0, vmax = 10.0, dt = 1.0 / 30.0;
  float high_score = 0.5f;
  bool has_camera = false;
  double Ki[12] = {}, Ko[16] = {};
};
Staged* make_ucmc(std::shared_ptr<Device>, const UcmcParams& p);
// BoostTrack (src/trackers/boosttrack.cpp), motion-only configuration
struct BoostParams {
  float det_thresh = 0.6f;
  int max_age = 60, min_hits = 3;
  float iou_threshold = 0.3f;
  int min_box_area = 10;
  float aspect_ratio_thresh = 1.6f, lambda_iou = 0.5f, lambda_mhd = 0.25f, lambda_shape = 0.25f;
  bool use_dlo = true, use_duo = true;
  float dlo_coef = 0.65f;
  bool use_sb = false, use_vt = false;
  bool with_reid = false;  // embeddings come with update(); without them the tracker runs motion-only (as the reference does, :539-551)
};
Staged* make_boosttrack(std::shared_ptr<Device>, const BoostParams& p);
// HybridSORT (src/trackers/hybridsort.cpp) as the reference runs it; with_reid: only without embeddings (its all-zero features)
struct HybridParams {
  float det_thresh = 0.7f;
  int max_age = 30, min_hits = 3;
  float iou_threshold = 0.15f;
  int asso = 1;  // 0 IoU (also giou / ciou / diou there), 1 hmiou
  float low_thresh = 0.1f;
  bool use_byte = true;
  float track_thresh = 0.5f, eg_high = 4.6f, eg_low = 1.3f;
  bool tcm_first = true, tcm_byte = true;
  float tcm_byte_weight = 1.0f;
  bool with_reid = false;
};
Staged* make_hybridsort(std::shared_ptr<Device>, const HybridParams& p);
Staged* make_botsort(std::shared_ptr<Device>, float track_high, float track_low, float new_track, int track_buffer,
                     float match_thresh, float proximity, float appearance, int frame_rate, bool fuse_first, bool with_reid,
                     int max_age, int max_obs);

}  // namespace motcpp::rt
