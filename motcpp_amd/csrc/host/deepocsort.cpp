// DeepOC-SORT on the MI355X hot path: host lifecycle of src/trackers/deepocsort.cpp:589-945 — OC-SORT's
// observation history and velocity directions plus an embedding per track — with the numeric work on the device: XYSR
// Kalman predict / update / initiate / camera-motion warp, the IoU + velocity-direction cost (mot_ocsort_cost), the
// embedding similarity dets_embs . trk_embs^T on the fp32 matrix cores (mot_embedding_cost, MOT_EMB_DOT), its zeroing
// and adaptive weighting into the cost (mot_deepoc_cost), the trivial-case shortcut + LAP, the -IoU OCR rematch with its
// max-IoU gate, and the embedding maintenance (normalise at birth, EMA with a per-detection alpha: mot_feat_update
// modes 2 / 3). Reproduced quirks: in the LAP branch every unmatched detection and track enters the unmatched lists twice
// (:456-503), so the OCR stage sees duplicated rows / columns and each unmatched detection spawns two tracks; the NaN-row
// rule of :683-690; ids from 1. ReID inference and the image registration are outside the path: embeddings (N x D) come
// with update(), the 2 x 3 warp of the next frame through set_camera_motion().
//
// Stages: -1 (only with a warp) camera-motion correction of the stored states | 0 predict + similarity + first
// association | 2 OCR rematch | 4.. Kalman / embedding updates and spawns (in rounds when a slot is updated twice).
#include <cmath>
#include <map>
#include <unordered_set>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

struct DObs { float v[5]; };

struct DTrk {
  int id = 0, slot = -1, age = 0, hits = 0, hit_streak = 0, tsu = 0, cls = 0, det_ind = 0;
  float conf = 0.f;
  DObs last_obs{{-1, -1, -1, -1, -1}};
  std::vector<std::pair<int, DObs>> observations;  // age -> observation: the last delta_t + 2 entries (the only ones ever read)
  float vel[2] = {0.f, 0.f};
};

class DeepOCSortGpu final : public Staged {
 public:
  DeepOCSortGpu(std::shared_ptr<Device> dev, float det_thresh, int max_age, int /*max_obs*/, int min_hits, float iou_threshold,
                int delta_t, float inertia, float w_emb, float alpha_fixed, float aw_param, bool emb_off, bool cmc_off, bool aw_off,
                float q_xy, float q_s, int asso)
      : core_(std::move(dev), MOT_KF_XYSR), det_thresh_(det_thresh), max_age_(max_age), min_hits_(min_hits), thr_(iou_threshold),
        delta_t_(delta_t), inertia_(inertia), w_emb_(w_emb), alpha_fixed_(alpha_fixed), aw_param_(aw_param), emb_off_(emb_off),
        cmc_off_(cmc_off), aw_off_(aw_off), asso_(asso) {
    core_.q[0] = 0.01f * q_xy;  // the tracker scales the constructor's already-scaled entries again (deepocsort.cpp:88-90)
    core_.q[1] = 0.01f * q_xy;
    core_.q[2] = 0.0001f * q_s;
  }
  ~DeepOCSortGpu() override { if (feat_) mot_free(core_.dev().ctx, feat_); }
  Core& core() override { return core_; }
  void reset() override { frame_count_ = 0; next_id_ = 0; trk_.clear(); core_.clear_slots(); has_warp_ = false; }
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const DTrk& t : trk_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }
  const float* feature_slab(int* dim, std::vector<char>* has) const override {
    *dim = emb_off_ ? 0 : D_;
    for (size_t i = 0; i < trk_.size(); ++i) has->push_back(1);
    return emb_off_ ? nullptr : feat_;
  }
  bool set_camera_motion(const float* w) override {
    has_warp_ = (w != nullptr);
    if (w) for (int i = 0; i < 6; ++i) warp_[i] = w[i];
    return true;
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    ++frame_count_;
    frame_diag_ = static_cast<float>(std::sqrt(static_cast<double>(in.img_w * in.img_w + in.img_h * in.img_h)));
    high_.clear();
    raw_.assign(static_cast<size_t>(6) * in.n, 0.f);
    n_ = in.n;
    for (int k = 0; k < 6; ++k)
      for (int i = 0; i < in.n; ++i) raw_[static_cast<size_t>(k) * in.n + i] = in.dets[static_cast<size_t>(k) * in.ld + i];
    for (int i = 0; i < in.n; ++i)
      if (conf(i) > det_thresh_) high_.push_back(i);  // :601-611 (no second, low-score stage)
    const int nd = static_cast<int>(high_.size());
    // embeddings :619-633: the supplied rows (all detections; rows of the kept ones are gathered by index) — or none
    have_emb_ = !emb_off_ && nd > 0;
    if (have_emb_) {
      if ((in.embs == nullptr && in.d_embs == nullptr) || in.emb_dim <= 0)
        throw Error("DeepOcSort: embeddings are required (ReID inference is outside this library): pass embs or construct with embedding_off");
      if (D_ == 0) D_ = in.emb_dim;
      if (D_ != in.emb_dim) throw Error("DeepOcSort: embedding dimension changed between frames");
      if (in.d_embs) emb_raw_ = in.d_embs;
      else {
        Span<float> raw = core_.dev().up->alloc<float>(static_cast<size_t>(in.n) * D_);
        if (in.embs_rowmajor) std::memcpy(raw.h, in.embs, sizeof(float) * static_cast<size_t>(in.n) * D_);
        else
          for (int i = 0; i < in.n; ++i)
            for (int k = 0; k < D_; ++k) raw.h[static_cast<size_t>(i) * D_ + k] = in.embs[static_cast<size_t>(k) * in.emb_ld + i];
        emb_raw_ = raw.d;
      }
    }
    core_.reserve(2 * nd + 8, 8);
    ensure_feat_slab();
    dets_ = core_.upload_dets(in.dets, in.n, in.ld, MOT_DET_XYSR, in.d_dets, in.d_ld);
    // dets_alpha :646-648
    alpha_.assign(in.n, alpha_fixed_);
    for (int i : high_) {
      const float trust = (conf(i) - det_thresh_) / (1.0f - det_thresh_);
      alpha_[i] = alpha_fixed_ + (1.0f - alpha_fixed_) * (1.0f - trust);
    }
    upd_.clear(); um_dets_.clear(); um_trks_.clear();
    assoc_ = Core::Lap(); rematch_ = Core::Lap();
    pbox_ = Span<float>();
    const bool warp_now = has_warp_ && !cmc_off_;
    has_warp_ = false;
    if (warp_now && !trk_.empty()) {  // :633-643, before the prediction: its own flush (the runtime predicts before it warps)
      const float m[2][2] = {{warp_[0], warp_[1]}, {warp_[3], warp_[4]}}, t[2] = {warp_[2], warp_[5]};
      std::vector<int> slots;
      for (DTrk& tr : trk_) { affine_obs(tr, m, t); slots.push_back(tr.slot); }
      const float w9[9] = {warp_[0], warp_[1], warp_[2], warp_[3], warp_[4], warp_[5], 0.f, 0.f, 1.f};
      core_.warp(slots, w9);
      stage_ = -1;
      return;
    }
    stage_ = 0;
    predict_and_first();
  }

  bool advance() override {
    while (true) {
      switch (stage_) {
        case -1: stage_ = 0; predict_and_first(); return true;
        case 0: {
          if (nt0_ == 0) { stage_ = 1; continue; }
          const int nt = nt0_;
          std::vector<int> del;
          for (int i = 0; i < nt; ++i)
            if (std::isnan(pbox_.h[i]) || std::isnan(pbox_.h[nt + i]) || std::isnan(pbox_.h[2 * nt + i]) || std::isnan(pbox_.h[3 * nt + i])) del.push_back(i);
          if (!del.empty()) {
            for (auto it = del.rbegin(); it != del.rend(); ++it) { core_.release_slot(trk_[*it].slot); trk_.erase(trk_.begin() + *it); }
            assoc_ = Core::Lap();
            stage_ = 1;
            if (!trk_.empty()) { queue_first(static_cast<int>(trk_.size())); return true; }  // rows = FIRST nt' predicted boxes (:690)
            continue;
          }
          stage_ = 1;
          continue;
        }
        case 1: {
          if (trk_.empty()) {  // :652-664 / :692-705: every kept detection starts a track, nothing is emitted
            um_dets_.clear(); um_trks_.clear();
            for (int i = 0; i < static_cast<int>(high_.size()); ++i) um_dets_.push_back(i);
            upd_.clear();
            silent_ = true;
            stage_ = 4;
            continue;
          }
          silent_ = false;
          after_first();
          stage_ = 3;
          if (!um_dets_.empty() && !um_trks_.empty()) { queue_rematch(); return true; }
          rematch_ = Core::Lap();
          continue;
        }
        case 3: {
          if (rematch_.queued) after_rematch();
          stage_ = 4;
          continue;
        }
        case 4: {
          finish_lists();
          round_ = 0;
          stage_ = 5;
          queue_round();
          return true;
        }
        case 5: {
          ++round_;
          if (round_ < n_rounds_) { queue_round(); return true; }
          emit();
          stage_ = 6;
          return false;
        }
        default:
          return false;
      }
    }
  }

 private:
  float conf(int i) const { return raw_[static_cast<size_t>(4) * n_ + i]; }
  int cls(int i) const { return static_cast<int>(raw_[static_cast<size_t>(5) * n_ + i]); }
  void box(int i, float b[4]) const { for (int k = 0; k < 4; ++k) b[k] = raw_[static_cast<size_t>(k) * n_ + i]; }

  static DObs k_previous_obs(const DTrk& t, int k) {  // :24-47
    if (t.observations.empty()) return DObs{{-1, -1, -1, -1, -1}};
    for (int i = 0; i < k; ++i) {
      const int key = t.age - (k - i);
      for (const auto& o : t.observations)
        if (o.first == key) return o.second;
    }
    return t.observations.back().second;
  }
  static void speed_direction(const float* b1, const float* b2, float out[2]) {  // :239-250
    const float cx1 = (b1[0] + b1[2]) / 2.0f, cy1 = (b1[1] + b1[3]) / 2.0f;
    const float cx2 = (b2[0] + b2[2]) / 2.0f, cy2 = (b2[1] + b2[3]) / 2.0f;
    const float dy = cy2 - cy1, dx = cx2 - cx1;
    const float norm = std::sqrt(dy * dy + dx * dx) + 1e-6f;
    out[0] = dy / norm; out[1] = dx / norm;
  }
  // apply_affine_correction :189-236, the observation part (the filter state goes through mot_kf_warp)
  void affine_obs(DTrk& t, const float m[2][2], const float tr[2]) const {
    auto move = [&](float* o) {
      const float x1 = o[0], y1 = o[1], x2 = o[2], y2 = o[3];
      o[0] = (m[0][0] * x1 + m[0][1] * y1) + tr[0];
      o[1] = (m[1][0] * x1 + m[1][1] * y1) + tr[1];
      o[2] = (m[0][0] * x2 + m[0][1] * y2) + tr[0];
      o[3] = (m[1][0] * x2 + m[1][1] * y2) + tr[1];
    };
    if (t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3] > 0) move(t.last_obs.v);
    for (int dt = delta_t_; dt >= 0; --dt)
      for (auto& o : t.observations)
        if (o.first == t.age - dt && o.second.v[0] + o.second.v[1] + o.second.v[2] + o.second.v[3] > 0) move(o.second.v);
  }
  void apply(DTrk& t, int det) {  // update :101-145 (Kalman and embedding parts queued)
    t.det_ind = det;
    t.conf = conf(det); t.cls = cls(det);
    float b[4];
    box(det, b);
    const float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
    if (ls >= 0) {
      const DObs pb = k_previous_obs(t, delta_t_);
      if (pb.v[0] + pb.v[1] + pb.v[2] + pb.v[3] >= 0) speed_direction(pb.v, b, t.vel);
      else speed_direction(t.last_obs.v, b, t.vel);
    }
    for (int k = 0; k < 4; ++k) t.last_obs.v[k] = b[k];
    t.last_obs.v[4] = t.conf;
    if (!t.observations.empty() && t.observations.back().first == t.age) t.observations.back().second = t.last_obs;
    else t.observations.push_back({t.age, t.last_obs});
    if (static_cast<int>(t.observations.size()) > delta_t_ + 2) t.observations.erase(t.observations.begin());
    t.tsu = 0; ++t.hits; ++t.hit_streak;
    upd_.push_back({t.slot, det});
  }

  void ensure_feat_slab() {
    if (emb_off_ || D_ == 0) return;
    const int need = core_.cap();
    if (feat_cap_ >= need) return;
    void* nf = nullptr;
    core_.dev().check(mot_malloc(core_.dev().ctx, sizeof(float) * static_cast<size_t>(need) * D_, &nf), "embedding slab alloc");
    core_.dev().check(mot_memset(core_.dev().ctx, nf, 0, sizeof(float) * static_cast<size_t>(need) * D_), "embedding slab clear");
    if (feat_) {
      core_.dev().check(mot_memcpy_d2d(core_.dev().ctx, nf, feat_, sizeof(float) * static_cast<size_t>(feat_cap_) * D_), "embedding slab copy");
      core_.dev().check(mot_ctx_sync(core_.dev().ctx), "embedding slab sync");
      mot_free(core_.dev().ctx, feat_);
    }
    feat_ = static_cast<float*>(nf);
    feat_cap_ = need;
  }

  void predict_and_first() {
    const int nt = static_cast<int>(trk_.size());
    nt0_ = nt;
    if (nt == 0) return;
    std::vector<int> slots(nt);
    std::vector<uint8_t> fl(nt, MOT_KF_OCSORT_CLAMP);
    for (int i = 0; i < nt; ++i) {  // predict :152-168
      DTrk& t = trk_[i];
      slots[i] = t.slot;
      ++t.age;
      if (t.tsu > 0) t.hit_streak = 0;
      ++t.tsu;
    }
    pbox_d_ = core_.predict(slots, nullptr, &fl, &pbox_);
    queue_first(nt);
  }

  void queue_first(int nt) {
    std::vector<float> vel(static_cast<size_t>(2) * nt), prev(static_cast<size_t>(5) * nt);
    std::vector<int> slots(nt);
    for (int i = 0; i < nt; ++i) {
      vel[i] = trk_[i].vel[0]; vel[nt + i] = trk_[i].vel[1];
      const DObs k = k_previous_obs(trk_[i], delta_t_);
      for (int c = 0; c < 5; ++c) prev[static_cast<size_t>(c) * nt + i] = k.v[c];
      slots[i] = trk_[i].slot;
    }
    const int nd = static_cast<int>(high_.size());
    assoc_nt_ = nt;
    if (nd == 0) { assoc_ = Core::Lap(); return; }
    Span<float> dv = core_.floats(vel), dp = core_.floats(prev);
    high_d_ = core_.ints(high_);
    const int ld = round_up(nt, 4);
    float* cost = core_.dev().tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
    iou_d_ = core_.dev().tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
    {
      mot_ocsort_task t{};
      t.nd = nd; t.nt = nt; t.dbox = dets_.d_box; t.ldd = dets_.n; t.didx = high_d_.d; t.dconf = dets_.d_conf();
      t.tbox = pbox_d_; t.ldt = nt0_; t.vel = dv.d; t.ldv = nt; t.prev = dp.d; t.ldp = nt; t.vdc_weight = inertia_;
      t.cost = cost; t.iou = iou_d_; t.ldc = ld;
      t.assoc = asso_; t.frame_diag = frame_diag_;
      core_.dev().q().oc.push_back(t);
    }
    if (have_emb_) {  // emb_cost = dets_embs * trk_embs^T (:746-757), zeroed / weighted into the cost (:419-441)
      if (core_.cap() > feat_cap_) throw Error("DeepOcSort: embedding slab smaller than the Kalman slab");
      Span<int32_t> sl = core_.ints(slots);
      float* emb = core_.dev().tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
      mot_cos_task c{};
      c.n = nd; c.m = nt; c.d = D_; c.a = emb_raw_; c.lda = D_; c.aidx = high_d_.d; c.b = feat_; c.ldb = D_; c.bidx = sl.d;
      c.out = emb; c.ldo = ld;
      core_.dev().q().dot.push_back(c);
      mot_deep_task d{};
      d.nd = nd; d.nt = nt; d.emb = emb; d.lde = ld; d.iou = iou_d_; d.ldi = ld; d.cost = cost; d.ldc = ld;
      d.rw = core_.dev().tmp->alloc<float>(nd).d; d.cw = core_.dev().tmp->alloc<float>(nt).d;
      d.w = w_emb_; d.aw_param = aw_param_; d.aw_off = aw_off_ ? 1 : 0;
      core_.dev().q().deep.push_back(d);
    }
    assoc_ = core_.lap(cost, ld, nd, nt, -thr_, MOT_LAP_OCSORT, iou_d_, ld, thr_, true);
  }

  void after_first() {  // deepocsort_assoc::associate :443-507
    const int nd = static_cast<int>(high_.size()), nt = static_cast<int>(trk_.size());
    um_dets_.clear(); um_trks_.clear(); upd_.clear();
    std::vector<char> md(nd, 0), mt(nt, 0);
    std::vector<std::pair<int, int>> matches;
    if (assoc_.queued) {
      const int path = assoc_.info.h[0];
      if (path == 0) record(assoc_);
      for (int i = 0; i < nd; ++i) {
        const int j = assoc_.x.h[i];
        if (j < 0) continue;
        if (path == 1 || assoc_.xval.h[i] >= thr_) { matches.push_back({i, j}); md[i] = 1; mt[j] = 1; }
        else { um_dets_.push_back(i); um_trks_.push_back(j); }
      }
      if (path == 0) {  // :484-489: the assignment's own unmatched lists — the sweep below then adds them a second time
        for (int i = 0; i < nd; ++i) if (assoc_.x.h[i] < 0) um_dets_.push_back(i);
        for (int j = 0; j < nt; ++j) if (assoc_.y.h[j] < 0) um_trks_.push_back(j);
      }
    }
    for (int i = 0; i < nd; ++i) if (!md[i]) um_dets_.push_back(i);
    for (int j = 0; j < nt; ++j) if (!mt[j]) um_trks_.push_back(j);
    for (const auto& m : matches) apply(trk_[m.second], high_[m.first]);
  }

  void queue_rematch() {  // OCR :796-872 over the (duplicated) lists
    const int nl = static_cast<int>(um_trks_.size());
    std::vector<float> lt(static_cast<size_t>(4) * nl);
    for (int i = 0; i < nl; ++i)
      for (int c = 0; c < 4; ++c) lt[static_cast<size_t>(c) * nl + i] = trk_[um_trks_[i]].last_obs.v[c];
    Span<float> dlt = core_.floats(lt);
    std::vector<int> didx;
    for (int d : um_dets_) didx.push_back(high_[d]);
    left_d_ = core_.ints(didx);
    Core::IouArgs a;
    a.a = dets_.d_box; a.lda = dets_.n; a.aidx = left_d_.d; a.n = static_cast<int>(didx.size());
    a.b = dlt.d; a.ldb = nl; a.m = nl;
    a.mode = MOT_COST_NEG_IOU; a.assoc = asso_; a.frame_diag = frame_diag_;
    rematch_ = core_.lap_geom(a, -thr_, MOT_LAP_GATE_MIN, -thr_, true);
  }
  void after_rematch() {
    if (rematch_.info.h[0] == 2) return;
    record(rematch_);
    std::unordered_set<int> rmd, rmt;
    for (int i = 0; i < rematch_.n; ++i) {
      const int j = rematch_.x.h[i];
      if (j < 0) continue;
      if (-rematch_.xval.h[i] < thr_) continue;
      const int di = um_dets_[i], ti = um_trks_[j];
      apply(trk_[ti], high_[di]);
      rmd.insert(di); rmt.insert(ti);
    }
    std::vector<int> kd, kt;
    for (int d : um_dets_) if (!rmd.count(d)) kd.push_back(d);
    for (int t : um_trks_) if (!rmt.count(t)) kt.push_back(t);
    um_dets_ = kd; um_trks_ = kt;
  }

  void finish_lists() {
    for (int t : um_trks_) trk_[t].det_ind = 0;  // update(None) :875-877
    init_dst_.clear(); init_meas_.clear();
    for (int d : um_dets_) {  // :880-890 — once per list entry: a detection listed twice starts two tracks
      DTrk t;
      t.id = ++next_id_;
      t.slot = core_.new_slot();
      const int det = high_[d];
      t.conf = conf(det); t.cls = cls(det); t.det_ind = det;
      init_dst_.push_back(t.slot); init_meas_.push_back(det);
      trk_.push_back(t);
    }
    ensure_feat_slab();
    // Kalman + embedding updates, split into rounds when a slot is updated more than once in a frame
    std::map<int, int> seen;
    rounds_.clear();
    for (const auto& u : upd_) {
      const int r = seen[u.first]++;
      if (static_cast<int>(rounds_.size()) <= r) rounds_.resize(r + 1);
      rounds_[r].push_back(u);
    }
    n_rounds_ = std::max<int>(1, static_cast<int>(rounds_.size()));
    emit_idx_.clear();
    std::vector<int> need_state;
    if (!silent_) {
      for (int i = static_cast<int>(trk_.size()) - 1; i >= 0; --i) {
        const DTrk& t = trk_[i];
        if (t.tsu < 1 && (t.hit_streak >= min_hits_ || frame_count_ <= min_hits_)) {
          emit_idx_.push_back(i);
          const float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
          if (ls < 0) need_state.push_back(i);
        }
      }
    }
    need_state_ = need_state;
  }
  void queue_feat(const std::vector<int>& slots, const std::vector<int>& dets, int mode) {
    if (!have_emb_ || slots.empty()) return;
    if (core_.cap() > feat_cap_) throw Error("DeepOcSort: embedding slab smaller than the Kalman slab");
    Span<int32_t> s = core_.ints(slots), d = core_.ints(dets);
    mot_feat_task t{};
    t.n = static_cast<int>(slots.size()); t.d = D_; t.feat = feat_; t.ldf = D_; t.slot = s.d; t.src = emb_raw_; t.lds = D_; t.sidx = d.d;
    t.mode = mode; t.alpha = alpha_fixed_;
    if (mode == 3) {
      std::vector<float> al(slots.size());
      for (size_t i = 0; i < dets.size(); ++i) al[i] = alpha_[dets[i]];
      t.alpha_i = core_.floats(al).d;
    }
    (mode == 3 ? core_.dev().q().feat_ema : core_.dev().q().feat_set).push_back(t);
  }
  void queue_round() {
    if (round_ == 0) {
      core_.initiate(init_dst_, init_meas_, dets_);
      queue_feat(init_dst_, init_meas_, 2);  // ctor :73-79: the detection's embedding, normalised where its norm exceeds 1e-6
    }
    if (round_ < static_cast<int>(rounds_.size())) {
      std::vector<int> s, m;
      for (const auto& u : rounds_[round_]) { s.push_back(u.first); m.push_back(u.second); }
      core_.update(s, s, m, dets_);
      queue_feat(s, m, 3);  // update_emb :132-150 with dets_alpha
    }
    if (round_ == n_rounds_ - 1 && !need_state_.empty()) {
      std::vector<int> slots;
      for (int i : need_state_) slots.push_back(trk_[i].slot);
      core_.boxes(slots, &sbox_);
    }
  }
  void emit() {
    if (!silent_) {
      const int ns = static_cast<int>(need_state_.size());
      for (int i : emit_idx_) {
        const DTrk& t = trk_[i];
        float b[4] = {t.last_obs.v[0], t.last_obs.v[1], t.last_obs.v[2], t.last_obs.v[3]};
        if (b[0] + b[1] + b[2] + b[3] < 0) {
          for (int k = 0; k < ns; ++k)
            if (need_state_[k] == i) for (int c = 0; c < 4; ++c) b[c] = sbox_.h[static_cast<size_t>(c) * ns + k];
        }
        push_row(b, 1, 0, t.id, t.conf, t.cls, t.det_ind);
      }
      for (int i = static_cast<int>(trk_.size()) - 1; i >= 0; --i)
        if (trk_[i].tsu > max_age_) { core_.release_slot(trk_[i].slot); trk_.erase(trk_.begin() + i); }
    }
  }

  Core core_;
  float det_thresh_;
  int max_age_, min_hits_;
  float thr_;
  int delta_t_;
  float inertia_, w_emb_, alpha_fixed_, aw_param_;
  bool emb_off_, cmc_off_, aw_off_;
  int frame_count_ = 0, next_id_ = 0, stage_ = 0, n_ = 0, nt0_ = 0, assoc_nt_ = 0, round_ = 0, n_rounds_ = 1, D_ = 0, feat_cap_ = 0;
  bool silent_ = false, have_emb_ = false, has_warp_ = false;
  float warp_[6] = {1, 0, 0, 0, 1, 0};
  float* feat_ = nullptr;
  const float* emb_raw_ = nullptr;
  std::vector<DTrk> trk_;
  std::vector<float> raw_, alpha_;
  std::vector<int> high_, um_dets_, um_trks_, init_dst_, init_meas_, emit_idx_, need_state_;
  std::vector<std::pair<int, int>> upd_;
  std::vector<std::vector<std::pair<int, int>>> rounds_;
  Core::Dets dets_;
  Span<float> pbox_, sbox_;
  float* pbox_d_ = nullptr;
  float* iou_d_ = nullptr;
  int asso_ = MOT_ASSOC_IOU;
  float frame_diag_ = 1.f;
  Span<int32_t> high_d_, left_d_;
  Core::Lap assoc_, rematch_;
};

}  // namespace

Staged* make_deepocsort(std::shared_ptr<Device> dev, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold,
                        int delta_t, float inertia, float w_emb, float alpha_fixed, float aw_param, bool emb_off, bool cmc_off,
                        bool aw_off, float q_xy, float q_s, int asso) {
  return new DeepOCSortGpu(std::move(dev), det_thresh, max_age, max_obs, min_hits, iou_threshold, delta_t, inertia, w_emb, alpha_fixed,
                           aw_param, emb_off, cmc_off, aw_off, q_xy, q_s, asso);
}

}  // namespace motcpp::rt
