// ByteTrack on the MI355X hot path: host lifecycle of src/trackers/bytetrack.cpp:166-706 (list algebra,
// states, ids — including the reference's copy-semantics quirks) with every numeric
// step on the device: XYAH Kalman predict/update/initiate, box conversion, IoU(+score fusion) matrices,
// duplicate detection and the three linear assignments. Track state never leaves HBM.
//
// Stages per frame (one flush each):
//   0  det prepare, predict pool copies into scratch slots, IoU+fuse cost, LAP#1 (match_thresh)
//   1  boxes of the un-predicted remaining/unconfirmed originals, IoU costs, LAP#2 (0.5), LAP#3 (0.7)
//   2  Kalman updates / initiations, boxes of active+lost, duplicate pairs (iou_dist < 0.15)
#include <algorithm>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

enum St { New = 0, Tracked = 1, Lost = 2, Removed = 3 };

struct Trk {
  int id = 0, slot = -1;
  int state = New;
  bool activated = false;
  int tracklet_len = 0, frame_id = 0, start_frame = 0;
  float conf = 0.f;
  int cls = 0, det_ind = 0;
};

class ByteTrackGpu final : public Staged {
 public:
  ByteTrackGpu(std::shared_ptr<Device> dev, float min_conf, float track_thresh, float match_thresh, int track_buffer,
               int frame_rate, int /*max_age*/, int /*max_obs*/)
      : core_(std::move(dev), MOT_KF_XYAH), min_conf_(min_conf), track_thresh_(track_thresh), match_thresh_(match_thresh) {
    max_time_lost_ = static_cast<int>(frame_rate / 30.0f * track_buffer);  // bytetrack.cpp:141-142
    det_thresh_ = track_thresh_;                                             // :145
  }
  Core& core() override { return core_; }
  void reset() override {
    frame_count_ = 0;
    active_.clear(); lost_.clear();
    core_.clear_slots();
  }
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : active_) { ids->push_back(t.id); slots->push_back(t.slot); }
    for (const Trk& t : lost_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    ++frame_count_;
    stage_ = 0;
    high_.clear(); second_.clear();
    for (int i = 0; i < in.n; ++i) {
      const float c = in.dets[static_cast<size_t>(4) * in.ld + i];
      if (c > track_thresh_) high_.push_back(i);
    }
    for (int i = 0; i < in.n; ++i) {
      const float c = in.dets[static_cast<size_t>(4) * in.ld + i];
      if (c > min_conf_ && c < track_thresh_) second_.push_back(i);
    }
    det_conf_.assign(in.n, 0.f); det_cls_.assign(in.n, 0);
    for (int i = 0; i < in.n; ++i) {
      det_conf_[i] = in.dets[static_cast<size_t>(4) * in.ld + i];
      det_cls_[i] = static_cast<int>(in.dets[static_cast<size_t>(5) * in.ld + i]);
    }
    unconf_idx_.clear(); tracked_idx_.clear();
    for (size_t i = 0; i < active_.size(); ++i) (active_[i].activated ? tracked_idx_ : unconf_idx_).push_back(static_cast<int>(i));
    // pool = tracked ∪ lost (by id), as copies predicted into scratch slots (:251-265)
    pool_.clear();
    IdSet& seen = set_a_;
    seen.clear();
    for (int i : tracked_idx_) { pool_.push_back({i, true}); seen.insert(active_[i].id); }
    for (size_t i = 0; i < lost_.size(); ++i)
      if (seen.insert(lost_[i].id)) pool_.push_back({static_cast<int>(i), false});
    const int np = static_cast<int>(pool_.size());
    core_.reserve(static_cast<int>(high_.size()) + 8, np + 8);
    dets_ = core_.upload_dets(in.dets, in.n, in.ld, MOT_DET_XYAH, in.d_dets, in.d_ld);

    lap1_ = Core::Lap();
    if (np > 0) {
      std::vector<int>&src = w_src_, &dst = w_dst_;
      std::vector<uint8_t>& fl = w_fl_;
      src.resize(np); dst.resize(np); fl.resize(np);
      for (int i = 0; i < np; ++i) {
        const Trk& t = trk(pool_[i]);
        src[i] = t.slot; dst[i] = core_.scratch_slot(i);
        fl[i] = (t.state != Tracked) ? MOT_KF_ZERO_V7 : 0;
      }
      pool_box_ = core_.predict(src, &dst, &fl, nullptr);
      if (!high_.empty()) {
        high_d_ = core_.ints(high_);
        Core::IouArgs a;
        a.a = pool_box_; a.lda = np; a.n = np;
        a.b = dets_.d_box; a.ldb = dets_.n; a.bidx = high_d_.d; a.m = static_cast<int>(high_.size());
        a.bconf = dets_.d_conf(); a.mode = MOT_COST_IOU_DIST_FUSE;
        lap1_ = core_.lap_geom(a, match_thresh_);
      }
    }
  }

  bool advance() override {
    while (true) {
      if (stage_ == 0) { after_first(); stage_ = 1; if (queued_) return true; continue; }
      if (stage_ == 1) { after_second(); stage_ = 2; return true; }
      if (stage_ == 2) { if (!finish()) return true; stage_ = 3; return false; }
      return false;
    }
  }

 private:
  struct PoolRef { int idx; bool in_active; };
  Trk& trk(const PoolRef& r) { return r.in_active ? active_[r.idx] : lost_[r.idx]; }

  void apply_match(Trk& t, int det) {  // STrack::update :71-89 / re_activate :55-69 (Kalman part queued separately)
    if (t.state == Tracked) { t.frame_id = frame_count_; ++t.tracklet_len; }
    else { t.tracklet_len = 0; t.frame_id = frame_count_; }
    t.state = Tracked; t.activated = true;
    t.conf = det_conf_[det]; t.cls = det_cls_[det]; t.det_ind = det;
  }

  void after_first() {
    queued_ = false;
    const int np = static_cast<int>(pool_.size()), nd = static_cast<int>(high_.size());
    const int32_t* x;
    const int32_t* y;
    if (lap1_.queued) {
      record(lap1_);
      x = lap1_.x.h;
      y = lap1_.y.h;
    } else {
      w_x_.assign(np, -1); w_y_.assign(nd, -1);
      x = w_x_.data(); y = w_y_.data();
      if (record_laps) laps_.push_back(LapRecord{std::vector<int>(np, -1), std::vector<int>(nd, -1)});  // utils::linear_assignment's empty-side early return (matching.cpp:20-27)
    }
    upd_src_.clear(); upd_dst_.clear(); upd_meas_.clear();
    refind_.clear();
    u_det_.clear();
    std::vector<int>& u_track = w_utrack_;
    u_track.clear();
    for (int i = 0; i < np; ++i) {
      if (x[i] < 0) { u_track.push_back(i); continue; }
      Trk& t = trk(pool_[i]);
      const bool was_tracked = (t.state == Tracked);
      upd_src_.push_back(core_.scratch_slot(i)); upd_dst_.push_back(t.slot); upd_meas_.push_back(high_[x[i]]);
      apply_match(t, high_[x[i]]);
      if (!was_tracked) refind_.push_back(t.id);
    }
    for (int j = 0; j < nd; ++j) if (y[j] < 0) u_det_.push_back(j);

    // second association: un-predicted originals of the still-Tracked, still-unmatched pool members (:367-442)
    r_tracked_.clear(); r_pool_.clear();
    for (int i : u_track)
      if (pool_[i].in_active && active_[pool_[i].idx].state == Tracked) { r_tracked_.push_back(pool_[i].idx); r_pool_.push_back(i); }
    lap2_ = Core::Lap(); lap3_ = Core::Lap();
    if (!second_.empty() && !r_tracked_.empty()) {
      std::vector<int>& slots = w_slots_;
      slots.clear();
      for (int ai : r_tracked_) slots.push_back(active_[ai].slot);
      float* rb = core_.boxes(slots, nullptr);
      second_d_ = core_.ints(second_);
      Core::IouArgs a;
      a.a = rb; a.lda = static_cast<int>(slots.size()); a.n = a.lda;
      a.b = dets_.d_box; a.ldb = dets_.n; a.bidx = second_d_.d; a.m = static_cast<int>(second_.size());
      a.mode = MOT_COST_IOU_DIST;
      lap2_ = core_.lap_geom(a, 0.5f);
      queued_ = true;
    }
    // unconfirmed tracks (stored, un-predicted state) vs. leftover high detections (:455-542)
    if (!unconf_idx_.empty() && !u_det_.empty()) {
      std::vector<int>&slots = w_slots2_, &rem = w_rem_;
      slots.clear(); rem.clear();
      for (int ai : unconf_idx_) slots.push_back(active_[ai].slot);
      for (int j : u_det_) rem.push_back(high_[j]);
      float* ub = core_.boxes(slots, nullptr);
      rem_d_ = core_.ints(rem);
      Core::IouArgs a;
      a.a = ub; a.lda = static_cast<int>(slots.size()); a.n = a.lda;
      a.b = dets_.d_box; a.ldb = dets_.n; a.bidx = rem_d_.d; a.m = static_cast<int>(rem.size());
      a.bconf = dets_.d_conf(); a.mode = MOT_COST_IOU_DIST_FUSE;
      lap3_ = core_.lap_geom(a, 0.7f);
      queued_ = true;
    }
  }

  void after_second() {
    std::vector<Trk>& lost_new = w_lost_new_;
    std::vector<int>& removed_ids = w_removed_;
    lost_new.clear(); removed_ids.clear();
    if (lap2_.queued) {
      record(lap2_);
      for (int i = 0; i < lap2_.n; ++i) {
        Trk& t = active_[r_tracked_[i]];
        const int j = lap2_.x.h[i];
        if (j >= 0) {
          upd_src_.push_back(core_.scratch_slot(r_pool_[i])); upd_dst_.push_back(t.slot); upd_meas_.push_back(second_[j]);
          apply_match(t, second_[j]);
        } else if (t.state != Lost) {
          t.state = Lost;
          lost_new.push_back(t);
        }
      }
    }
    std::vector<int>& u_det_final = w_udet_final_;
    u_det_final.clear();
    if (lap3_.queued) {
      record(lap3_);
      for (int j = 0; j < lap3_.m; ++j) if (lap3_.y.h[j] < 0) u_det_final.push_back(u_det_[j]);
      for (int i = 0; i < lap3_.n; ++i) {
        Trk& t = active_[unconf_idx_[i]];
        const int j = lap3_.x.h[i];
        if (j >= 0) {
          const int det = high_[u_det_[j]];
          upd_src_.push_back(t.slot); upd_dst_.push_back(t.slot); upd_meas_.push_back(det);  // un-predicted state (:524-527)
          apply_match(t, det);
        } else {
          t.state = Removed;
          removed_ids.push_back(t.id);
        }
      }
    } else {
      u_det_final = u_det_;
    }
    // new tracks (:546-554)
    std::vector<Trk>& fresh = w_fresh_;
    std::vector<int>&init_dst = w_init_dst_, &init_meas = w_init_meas_;
    fresh.clear(); init_dst.clear(); init_meas.clear();
    for (int j : u_det_final) {
      const int det = high_[j];
      if (det_conf_[det] >= det_thresh_) {
        Trk t;
        t.id = ++next_id_;
        t.slot = core_.new_slot();
        t.conf = det_conf_[det]; t.cls = det_cls_[det]; t.det_ind = det;
        t.tracklet_len = 0; t.state = Tracked;
        if (frame_count_ == 1) t.activated = true;
        t.frame_id = frame_count_; t.start_frame = frame_count_;
        init_dst.push_back(t.slot); init_meas.push_back(det);
        fresh.push_back(t);
      }
    }
    for (Trk& t : lost_)  // :557-562
      if (frame_count_ - t.frame_id > max_time_lost_) { t.state = Removed; removed_ids.push_back(t.id); }

    // list algebra (:565-580). Copies in the reference == moves here: one slot per id.
    std::vector<Trk>& na = w_na_;
    na.clear();
    IdSet& active_ids = set_a_;
    active_ids.clear();
    for (const Trk& t : active_)
      if (t.state == Tracked) { na.push_back(t); active_ids.insert(t.id); }
      else if (t.state == Removed) dead_slots_.push_back(t.slot);
    for (const Trk& t : fresh) { na.push_back(t); active_ids.insert(t.id); }
    for (int id : refind_)
      for (const Trk& t : lost_)
        if (t.id == id && active_ids.insert(id)) na.push_back(t);
    std::vector<Trk>& nl = w_nl_;
    nl.clear();
    IdSet& rm = set_b_;
    rm.clear();
    for (int id : removed_ids) rm.insert(id);
    for (const Trk& t : lost_) {
      if (active_ids.count(t.id)) continue;
      if (rm.count(t.id)) { dead_slots_.push_back(t.slot); continue; }
      nl.push_back(t);
    }
    for (const Trk& t : lost_new)
      if (!rm.count(t.id)) nl.push_back(t);
    active_.swap(na);  // (swap, not move: both buffers keep their capacity for the next frame)
    lost_.swap(nl);

    core_.initiate(init_dst, init_meas, dets_);
    core_.update(upd_src_, upd_dst_, upd_meas_, dets_);
    queue_boxes_and_dups(4 * static_cast<int>(active_.size() + lost_.size()) + 64);
  }

  void queue_boxes_and_dups(int cap) {
    std::vector<int>&sa = w_sa_, &sl = w_sl_;
    sa.clear(); sl.clear();
    for (const Trk& t : active_) sa.push_back(t.slot);
    for (const Trk& t : lost_) sl.push_back(t.slot);
    abox_ = Span<float>();
    float* da = core_.boxes(sa, &abox_);
    pairs_cap_ = 0;
    if (!sa.empty() && !sl.empty()) {
      float* dl = core_.boxes(sl, nullptr);
      pairs_ = core_.dev().down->alloc<int32_t>(static_cast<size_t>(2) * cap);
      npairs_ = core_.dev().zdown->alloc<int32_t>(4);  // zeroed on the device ahead of this stage's kernels
      pairs_cap_ = cap;
      mot_iou_task t{};
      t.n = static_cast<int>(sa.size()); t.m = static_cast<int>(sl.size());
      t.a = da; t.lda = t.n; t.b = dl; t.ldb = t.m; t.mode = MOT_COST_IOU_DIST;
      t.pairs = pairs_.d; t.npairs = npairs_.d; t.pairs_cap = cap; t.pair_thresh = 0.15f;
      core_.dev().q().iou.push_back(t);
    }
  }

  bool finish() {
    // remove_duplicate_stracks (:659-706)
    if (pairs_cap_ > 0) {
      const int np = npairs_.h[0];
      if (np > pairs_cap_) {  // overflow: redo with an exact-size buffer
        queue_boxes_and_dups(np + 16);
        return false;
      }
      std::vector<char>&dupa = w_dupa_, &dupb = w_dupb_;
      dupa.assign(active_.size(), 0); dupb.assign(lost_.size(), 0);
      for (int k = 0; k < np; ++k) {
        const int i = pairs_.h[2 * k], j = pairs_.h[2 * k + 1];
        const int tp = active_[i].frame_id - active_[i].start_frame;
        const int tq = lost_[j].frame_id - lost_[j].start_frame;
        if (tp > tq) dupb[j] = 1; else dupa[i] = 1;
      }
      std::vector<Trk>&ra = w_na_, &rb = w_nl_;
      std::vector<int>& keep_cols = w_cols_;
      ra.clear(); rb.clear(); keep_cols.clear();
      for (size_t i = 0; i < active_.size(); ++i) {
        if (!dupa[i]) { ra.push_back(active_[i]); keep_cols.push_back(static_cast<int>(i)); }
        else dead_slots_.push_back(active_[i].slot);
      }
      for (size_t j = 0; j < lost_.size(); ++j) {
        if (!dupb[j]) rb.push_back(lost_[j]);
        else dead_slots_.push_back(lost_[j].slot);
      }
      emit(ra, keep_cols, static_cast<int>(active_.size()));
      active_.swap(ra); lost_.swap(rb);
    } else {
      std::vector<int>& cols = w_cols_;
      cols.resize(active_.size());
      for (size_t i = 0; i < cols.size(); ++i) cols[i] = static_cast<int>(i);
      emit(active_, cols, static_cast<int>(active_.size()));
    }
    for (int s : dead_slots_) core_.release_slot(s);
    dead_slots_.clear();
    return true;
  }
  void emit(const std::vector<Trk>& list, const std::vector<int>& cols, int ld) {
    for (size_t i = 0; i < list.size(); ++i)
      if (list[i].activated) push_row(abox_.h, ld, cols[i], list[i].id, list[i].conf, list[i].cls, list[i].det_ind);
  }

  Core core_;
  IdSet set_a_, set_b_;
  float min_conf_, track_thresh_, match_thresh_, det_thresh_;
  int max_time_lost_;
  int frame_count_ = 0, next_id_ = 0;
  std::vector<Trk> active_, lost_;
  // frame scratch
  int stage_ = 0;
  bool queued_ = false;
  Core::Dets dets_;
  std::vector<int> high_, second_, unconf_idx_, tracked_idx_, u_det_, r_tracked_, r_pool_, refind_;
  std::vector<float> det_conf_;
  std::vector<int> det_cls_;
  std::vector<PoolRef> pool_;
  float* pool_box_ = nullptr;
  Span<int32_t> high_d_, second_d_, rem_d_, pairs_, npairs_;
  Span<float> abox_;
  int pairs_cap_ = 0;
  Core::Lap lap1_, lap2_, lap3_;
  std::vector<int> upd_src_, upd_dst_, upd_meas_, dead_slots_;
  // reusable work buffers: the lifecycle allocates nothing in steady state
  std::vector<int> w_src_, w_dst_, w_x_, w_y_, w_utrack_, w_slots_, w_slots2_, w_rem_, w_removed_, w_udet_final_, w_init_dst_,
      w_init_meas_, w_sa_, w_sl_, w_cols_;
  std::vector<uint8_t> w_fl_;
  std::vector<char> w_dupa_, w_dupb_;
  std::vector<Trk> w_lost_new_, w_fresh_, w_na_, w_nl_;
};

}  // namespace

Staged* make_bytetrack(std::shared_ptr<Device> dev, float min_conf, float track_thresh, float match_thresh, int track_buffer,
                       int frame_rate, int max_age, int max_obs) {
  return new ByteTrackGpu(std::move(dev), min_conf, track_thresh, match_thresh, track_buffer, frame_rate, max_age, max_obs);
}

}  // namespace motcpp::rt
