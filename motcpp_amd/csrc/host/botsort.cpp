// BoT-SORT on the MI355X hot path: host lifecycle of src/trackers/botsort.cpp:260-764 (no CMC, embeddings
// passed in) with the numeric work on the device: XYWH Kalman predict(in place)/update/initiate, the cosine
// distance matrix on the fp32 matrix cores, the gated IoU/appearance cost, three LAPs and the EMA feature
// maintenance. Track state and track features never leave HBM.
//
// Stages: 0 det/feature prepare, predict pool, cosine + gated cost, LAP#1 | 1 LAP#2 (0.5) and the
// unconfirmed association LAP#3 (0.7) | 2 Kalman/feature updates, new tracks, boxes of the rows to emit.

#include "staged.hpp"

namespace motcpp::rt {
namespace {

enum St { New = 0, Tracked = 1, Lost = 2, Removed = 3 };

struct Trk {
  int id = 0, slot = -1, state = New;
  bool activated = false, has_feat = false;
  int frame_id = 0, start_frame = 0, end_frame = 0, tracklet_len = 0;
  float conf = 0.f;
  int cls = 0, det_ind = -1;
};

class BotSortGpu final : public Staged {
 public:
  BotSortGpu(std::shared_ptr<Device> dev, float hi, float lo, float newt, int track_buffer, float match, float prox, float app,
             int frame_rate, bool fuse_first, bool with_reid, int /*max_age*/, int /*max_obs*/)
      : core_(std::move(dev), MOT_KF_XYWH), hi_(hi), lo_(lo), newt_(newt), match_(match), prox_(prox), app_(app),
        fuse_first_(fuse_first), with_reid_(with_reid) {
    max_time_lost_ = static_cast<int>(frame_rate / 30.0f * track_buffer);  // botsort.cpp:236-237
  }
  ~BotSortGpu() override { if (feat_) mot_free(core_.dev().ctx, feat_); }
  Core& core() override { return core_; }
  void reset() override { frame_count_ = 0; active_.clear(); lost_.clear(); next_id_ = 0; core_.clear_slots(); }  // :252-258
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : active_) { ids->push_back(t.id); slots->push_back(t.slot); }
    for (const Trk& t : lost_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }

  const float* feature_slab(int* dim, std::vector<char>* has) const override {
    *dim = D_;
    for (const Trk& t : active_) has->push_back(t.has_feat ? 1 : 0);
    for (const Trk& t : lost_) has->push_back(t.has_feat ? 1 : 0);
    return feat_;
  }

  bool set_camera_motion(const float* w) override {
    has_warp_ = (w != nullptr);
    if (w) {  // Matrix3f::Identity() with the top two rows replaced (:320-321)
      for (int i = 0; i < 6; ++i) warp_[i] = w[i];
      warp_[6] = 0.0f; warp_[7] = 0.0f; warp_[8] = 1.0f;
    }
    return true;
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    const bool warp_now = has_warp_;  // consumed by this frame, used or not
    has_warp_ = false;
    idle_ = (in.n == 0);  // :267-269: nothing happens, not even frame_count++ (nor the CMC step)
    if (idle_) return;
    ++frame_count_;
    stage_ = 0;
    first_.clear(); second_.clear();
    conf_.assign(in.n, 0.f); cls_.assign(in.n, 0);
    for (int i = 0; i < in.n; ++i) {
      conf_[i] = in.dets[static_cast<size_t>(4) * in.ld + i];
      cls_[i] = static_cast<int>(in.dets[static_cast<size_t>(5) * in.ld + i]);
      if (conf_[i] > hi_) first_.push_back(i);
      else if (conf_[i] > lo_) second_.push_back(i);
    }
    unconf_.clear(); pool_.clear();
    IdSet& seen = set_a_;
    seen.clear();
    for (size_t i = 0; i < active_.size(); ++i) {
      if (!active_[i].activated) unconf_.push_back(static_cast<int>(i));
      else { pool_.push_back({static_cast<int>(i), true}); seen.insert(active_[i].id); }
    }
    for (size_t i = 0; i < lost_.size(); ++i)
      if (seen.insert(lost_[i].id)) pool_.push_back({static_cast<int>(i), false});

    core_.reserve(static_cast<int>(first_.size()) + 8, 8);
    dets_ = core_.upload_dets(in.dets, in.n, in.ld, MOT_DET_XYWH, in.d_dets, in.d_ld);
    // appearance: raw rows for every detection, L2-normalised copies for the association (:38-46)
    have_emb_ = with_reid_ && (in.embs != nullptr || in.d_embs != nullptr) && in.emb_dim > 0;
    if (have_emb_) {
      if (D_ == 0) D_ = in.emb_dim;
      if (D_ != in.emb_dim) throw Error("BotSort: embedding dimension changed between frames");
      ensure_feat_slab();
      if (in.d_embs) emb_raw_ = in.d_embs;  // already in HBM
      else {
        Span<float> raw = core_.dev().up->alloc<float>(static_cast<size_t>(in.n) * D_);
        if (in.embs_rowmajor) std::memcpy(raw.h, in.embs, sizeof(float) * static_cast<size_t>(in.n) * D_);
        else
          for (int i = 0; i < in.n; ++i)
            for (int k = 0; k < D_; ++k) raw.h[static_cast<size_t>(i) * D_ + k] = in.embs[static_cast<size_t>(k) * in.emb_ld + i];
        emb_raw_ = raw.d;
      }
      emb_norm_ = core_.dev().tmp->alloc<float>(static_cast<size_t>(in.n) * D_).d;
      mot_feat_task t{};
      t.n = in.n; t.d = D_; t.feat = emb_norm_; t.ldf = D_; t.src = emb_raw_; t.lds = D_; t.mode = 0; t.alpha = 0.9f;
      core_.dev().q().feat_set.push_back(t);
    }
    if (warp_now && !unconf_.empty()) {  // multi_gmc(unconfirmed) :323 — these are not predicted
      std::vector<int> us;
      for (int ai : unconf_) us.push_back(active_[ai].slot);
      core_.warp(us, warp_);
    }
    const int np = static_cast<int>(pool_.size());
    lap1_ = Core::Lap();
    pool_box_ = nullptr;
    if (np > 0) {
      std::vector<int> slots(np);
      for (int i = 0; i < np; ++i) slots[i] = trk(pool_[i]).slot;
      // multi_predict :54-58 in place, then multi_gmc :60-91 on the predicted states when a warp was supplied (:317-324)
      pool_box_ = core_.predict(slots, nullptr, nullptr, nullptr, warp_now ? warp_ : nullptr);
      if (!first_.empty()) {
        first_d_ = core_.ints(first_);
        lap1_ = queue_assoc(pool_box_, np, nullptr, slots, first_d_, static_cast<int>(first_.size()), fuse_first_ ? 1 : 0, match_);
      }
    }
  }

  bool advance() override {
    if (idle_) return false;
    if (stage_ == 0) { after_first(); stage_ = 1; return true; }
    if (stage_ == 1) { after_second(); stage_ = 2; return true; }
    if (stage_ == 2) { emit(); stage_ = 3; }
    return false;
  }

 private:
  struct PoolRef { int idx; bool in_active; };
  Trk& trk(const PoolRef& r) { return r.in_active ? active_[r.idx] : lost_[r.idx]; }

  void ensure_feat_slab() {
    const int need = core_.cap();
    if (feat_cap_ >= need) return;
    void* nf = nullptr;
    core_.dev().check(mot_malloc(core_.dev().ctx, sizeof(float) * static_cast<size_t>(need) * D_, &nf), "feature slab alloc");
    core_.dev().check(mot_memset(core_.dev().ctx, nf, 0, sizeof(float) * static_cast<size_t>(need) * D_), "feature slab clear");
    if (feat_) {
      core_.dev().check(mot_memcpy_d2d(core_.dev().ctx, nf, feat_, sizeof(float) * static_cast<size_t>(feat_cap_) * D_), "feature slab copy");
      core_.dev().check(mot_ctx_sync(core_.dev().ctx), "feature slab sync");
      mot_free(core_.dev().ctx, feat_);
    }
    feat_ = static_cast<float*>(nf);
    feat_cap_ = need;
  }

  // gated IoU/appearance cost (:433-466 / :591-623) + LAP. rows: boxes [4][ld] (+gather), feature rows = slots
  Core::Lap queue_assoc(const float* boxes, int ld, const int32_t* aidx, const std::vector<int>& slots, const Span<int32_t>& didx,
                        int m, int fuse, float thresh) {
    const int n = static_cast<int>(slots.size());
    const float* emb = nullptr;
    int lde = 0;
    if (with_reid_ && have_emb_) {
      Span<int32_t> sl = core_.ints(slots);
      lde = round_up(m, 4);
      float* out = core_.dev().tmp->alloc<float>(static_cast<size_t>(n) * lde).d;
      mot_cos_task t{};
      t.n = n; t.m = m; t.d = D_; t.a = feat_; t.lda = D_; t.aidx = sl.d; t.b = emb_norm_; t.ldb = D_; t.bidx = didx.d;
      t.out = out; t.ldo = lde;
      t.norm_a = core_.dev().tmp->alloc<float>(n).d; t.norm_b = core_.dev().tmp->alloc<float>(m).d;
      core_.dev().q().cos.push_back(t);
      emb = out;
    }
    else if (with_reid_) lde = -1;  // no features this frame: the reference's cosine term is the constant 1 (matching.cpp:79-92, D = 0)
    Core::IouArgs a;
    a.a = boxes; a.lda = ld; a.aidx = aidx; a.n = n;
    a.b = dets_.d_box; a.ldb = dets_.n; a.bidx = didx.d; a.m = m; a.bconf = dets_.d_conf();
    a.mode = MOT_COST_BOTSORT; a.emb = emb; a.lde = lde; a.prox = prox_; a.app = app_; a.fuse = fuse;
    return core_.lap_geom(a, thresh);
  }

  void apply_match(Trk& t, int det, bool first_stage_det) {  // BotSTrack::update :133-156 / re_activate :111-131
    if (t.state == Tracked) ++t.tracklet_len; else t.tracklet_len = 0;
    t.frame_id = frame_count_; t.end_frame = frame_count_;
    t.state = Tracked; t.activated = true;
    t.conf = conf_[det]; t.cls = cls_[det]; t.det_ind = det;
    upd_slot_.push_back(t.slot); upd_meas_.push_back(det);
    if (first_stage_det && have_emb_) {  // update_features :158-169 (second-stage detections carry no feature)
      (t.has_feat ? ema_slot_ : set_slot_).push_back(t.slot);
      (t.has_feat ? ema_det_ : set_det_).push_back(det);
      t.has_feat = true;
    }
  }

  void after_first() {
    const int np = static_cast<int>(pool_.size()), nd = static_cast<int>(first_.size());
    std::vector<int> x(np, -1), y(nd, -1);
    if (lap1_.queued) { record(lap1_); x.assign(lap1_.x.h, lap1_.x.h + np); y.assign(lap1_.y.h, lap1_.y.h + nd); }
    else if (record_laps) laps_.push_back(LapRecord{x, y});
    upd_slot_.clear(); upd_meas_.clear(); ema_slot_.clear(); ema_det_.clear(); set_slot_.clear(); set_det_.clear();
    act_ids_.clear(); lost_new_.clear();
    std::vector<int> u_track;
    u_det_.clear();
    for (int i = 0; i < np; ++i) {
      if (x[i] < 0) { u_track.push_back(i); continue; }
      Trk& t = trk(pool_[i]);
      apply_match(t, first_[x[i]], true);
      act_ids_.insert(t.id);  // activated_stracks / refind_stracks (all end up Tracked)
    }
    for (int j = 0; j < nd; ++j) if (y[j] < 0) u_det_.push_back(j);
    // second association :497-563
    r_tracked_.clear();
    for (int i : u_track)
      if (trk(pool_[i]).state == Tracked) r_tracked_.push_back(i);
    lap2_ = Core::Lap(); lap3_ = Core::Lap();
    if (!r_tracked_.empty() && !second_.empty()) {
      r_tracked_d_ = core_.ints(r_tracked_);
      second_d_ = core_.ints(second_);
      Core::IouArgs a;
      a.a = pool_box_; a.lda = np; a.aidx = r_tracked_d_.d; a.n = static_cast<int>(r_tracked_.size());
      a.b = dets_.d_box; a.ldb = dets_.n; a.bidx = second_d_.d; a.m = static_cast<int>(second_.size());
      a.mode = MOT_COST_IOU_DIST;
      lap2_ = core_.lap_geom(a, 0.5f);
    }
    // unconfirmed :565-647
    if (!unconf_.empty() && !u_det_.empty()) {
      std::vector<int> slots, rem;
      for (int ai : unconf_) slots.push_back(active_[ai].slot);
      for (int j : u_det_) rem.push_back(first_[j]);
      float* ub = core_.boxes(slots, nullptr);
      rem_d_ = core_.ints(rem);
      lap3_ = queue_assoc(ub, static_cast<int>(slots.size()), nullptr, slots, rem_d_, static_cast<int>(rem.size()), 1, 0.7f);
    }
  }

  void after_second() {
    if (lap2_.queued) {
      record(lap2_);
      for (int i = 0; i < lap2_.n; ++i) {
        Trk& t = trk(pool_[r_tracked_[i]]);
        const int j = lap2_.x.h[i];
        if (j >= 0) { apply_match(t, second_[j], false); act_ids_.insert(t.id); }
        else if (t.state != Lost) { t.state = Lost; lost_new_.push_back(t.id); }
      }
    }
    std::vector<int> u_det_unc;  // indices into the filtered list detections[u_det_]
    if (lap3_.queued) {
      record(lap3_);
      for (int i = 0; i < lap3_.n; ++i) {
        Trk& t = active_[unconf_[i]];
        const int j = lap3_.x.h[i];
        if (j >= 0) { apply_match(t, first_[u_det_[j]], true); act_ids_.insert(t.id); }
        else t.state = Removed;
      }
      for (int j = 0; j < lap3_.m; ++j) if (lap3_.y.h[j] < 0) u_det_unc.push_back(j);
    } else {
      for (size_t j = 0; j < u_det_.size(); ++j) u_det_unc.push_back(static_cast<int>(j));
    }
    // new tracks :649-667
    std::vector<int> init_dst, init_meas;
    std::vector<Trk> fresh;
    for (int idx : u_det_unc) {
      const int det = first_[u_det_[idx]];
      if (conf_[det] < newt_) continue;
      Trk t;
      t.id = ++next_id_;
      t.slot = core_.new_slot();
      t.conf = conf_[det]; t.cls = cls_[det]; t.det_ind = det;
      t.tracklet_len = 0; t.state = Tracked;
      if (frame_count_ == 1) t.activated = true;
      t.frame_id = frame_count_; t.end_frame = frame_count_; t.start_frame = frame_count_;
      init_dst.push_back(t.slot); init_meas.push_back(det);
      if (have_emb_) { set_slot_.push_back(t.slot); set_det_.push_back(det); t.has_feat = true; }
      fresh.push_back(t);
      act_ids_.insert(t.id);
    }
    for (Trk& t : lost_)  // :669-676
      if (frame_count_ - t.end_frame > max_time_lost_) t.state = Removed;

    // prepare_output :678-764. Re-found lost tracks are dropped from lost_ and never re-enter active_.
    IdSet& active_ids = set_a_;
    active_ids.clear();
    for (const Trk& t : active_) if (act_ids_.count(t.id) && t.state == Tracked) active_ids.insert(t.id);
    for (const Trk& t : lost_) if (act_ids_.count(t.id) && t.state == Tracked) active_ids.insert(t.id);
    for (const Trk& t : fresh) active_ids.insert(t.id);
    std::vector<Trk> new_lost;
    IdSet& lost_ids = set_b_;
    lost_ids.clear();
    for (const Trk& t : lost_) {
      if (!active_ids.count(t.id) && t.state != Removed) { new_lost.push_back(t); lost_ids.insert(t.id); }
      else dead_.push_back(t.slot);
    }
    for (int id : lost_new_)
      for (const Trk& t : active_)
        if (t.id == id && !active_ids.count(id) && lost_ids.insert(id)) new_lost.push_back(t);
    std::vector<Trk> new_active;
    for (const Trk& t : active_) {
      if (t.state == Tracked) new_active.push_back(t);
      else if (t.state == Removed) dead_.push_back(t.slot);
    }
    for (const Trk& t : fresh) new_active.push_back(t);
    active_ = std::move(new_active);
    lost_ = std::move(new_lost);

    core_.initiate(init_dst, init_meas, dets_);
    core_.update(upd_slot_, upd_slot_, upd_meas_, dets_);
    if (have_emb_) {
      if (core_.cap() > feat_cap_) throw Error("BotSort: feature slab smaller than the Kalman slab");
      queue_feat(set_slot_, set_det_, 0);
      queue_feat(ema_slot_, ema_det_, 1);
    }
    out_idx_.clear();
    std::vector<int> slots;
    for (size_t i = 0; i < active_.size(); ++i)
      if (active_[i].activated) { out_idx_.push_back(static_cast<int>(i)); slots.push_back(active_[i].slot); }
    obox_ = Span<float>();
    core_.boxes(slots, &obox_);
  }
  void queue_feat(const std::vector<int>& slots, const std::vector<int>& dets, int mode) {
    if (slots.empty()) return;
    Span<int32_t> s = core_.ints(slots), d = core_.ints(dets);
    mot_feat_task t{};
    t.n = static_cast<int>(slots.size()); t.d = D_; t.feat = feat_; t.ldf = D_; t.slot = s.d; t.src = emb_raw_; t.lds = D_; t.sidx = d.d;
    t.mode = mode; t.alpha = 0.9f;
    (mode ? core_.dev().q().feat_ema : core_.dev().q().feat_set).push_back(t);
  }
  void emit() {
    const int n = static_cast<int>(out_idx_.size());
    for (int k = 0; k < n; ++k) {
      const Trk& t = active_[out_idx_[k]];
      push_row(obox_.h, n, k, t.id, t.conf, t.cls, t.det_ind);
    }
    for (int s : dead_) core_.release_slot(s);
    dead_.clear();
  }

  Core core_;
  float hi_, lo_, newt_, match_, prox_, app_;
  bool fuse_first_, with_reid_;
  int max_time_lost_;
  int frame_count_ = 0, next_id_ = 0, stage_ = 0, D_ = 0, feat_cap_ = 0;
  bool idle_ = false, have_emb_ = false;
  float* feat_ = nullptr;
  const float* emb_raw_ = nullptr;
  float* emb_norm_ = nullptr;
  std::vector<Trk> active_, lost_;
  std::vector<PoolRef> pool_;
  float warp_[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  bool has_warp_ = false;
  std::vector<int> first_, second_, unconf_, u_det_, r_tracked_, cls_, out_idx_, lost_new_, dead_;
  std::vector<int> upd_slot_, upd_meas_, ema_slot_, ema_det_, set_slot_, set_det_;
  std::vector<float> conf_;
  IdSet act_ids_, set_a_, set_b_;
  Core::Dets dets_;
  float* pool_box_ = nullptr;
  Span<int32_t> first_d_, second_d_, rem_d_, r_tracked_d_;
  Span<float> obox_;
  Core::Lap lap1_, lap2_, lap3_;
};

}  // namespace

Staged* make_botsort(std::shared_ptr<Device> dev, float track_high, float track_low, float new_track, int track_buffer,
                     float match_thresh, float proximity, float appearance, int frame_rate, bool fuse_first, bool with_reid,
                     int max_age, int max_obs) {
  return new BotSortGpu(std::move(dev), track_high, track_low, new_track, track_buffer, match_thresh, proximity, appearance,
                        frame_rate, fuse_first, with_reid, max_age, max_obs);
}

}  // namespace motcpp::rt
