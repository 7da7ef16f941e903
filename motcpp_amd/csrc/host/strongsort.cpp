// StrongSORT on the MI355X hot path: host lifecycle of src/trackers/strongsort.cpp:595-1012 (Tracker::predict / update / match,
// StrongSORT::update) with every float on the device — XYAH Kalman predict / NSA update / initiate, the nearest-sample cosine
// distance (raw inner products on the fp32 matrix cores, reduced by ss_nn_kernel), the motion gate (gate_kernel), the tlwh IoU cost
// (ss_iou_kernel), two assignments and the feature / sample-library maintenance. Track states, smoothed features and the sample
// library (nn_budget rows per track) never leave HBM.
//
// Stages: 0 detections, predict, sample x feature inner products, nearest-sample cost, gate + clamp, assignment A |
//         1 IoU cost of the candidates, assignment B | 2 Kalman updates / births, feature EMA, sample library, boxes of the rows to emit.
// The reference's quirks are kept (they decide ids): an EMPTY index list means "all of them" in matching_cascade / min_cost_matching
// (:358-368, :440-447), so with no confirmed track the appearance stage runs over every track, the candidates of the IoU stage then
// list those tracks TWICE (the unconfirmed ones + the unmatched ones of stage A with time_since_update == 1, :745-757) — a row
// per copy, and the copy that stays unmatched puts its (matched) track on the unmatched list, where a tentative track is deleted
// (:189-197) — and an empty list of unmatched detections makes the IoU stage see all detections again (:765-776).
#include <cmath>
#include <cstdlib>
#include <string>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

enum St { Tentative = 1, Confirmed = 2, Deleted = 3 };

struct Trk {
  int id = 0, slot = -1, state = Tentative;
  int hits = 1, age = 1, tsu = 0;
  float conf = 0.f;
  int cls = 0, det_ind = -1;
  bool has_feat = false;
  int n_samples = 0, head = 0;  // ring of the last nn_budget smoothed features (rows slot * budget + k of the sample slab)
};

class StrongSortGpu final : public Staged {
 public:
  StrongSortGpu(std::shared_ptr<Device> dev, float min_conf, float max_cos, float max_iou, int n_init, int budget, float lambda, float alpha, int max_age)
      : core_(std::move(dev), MOT_KF_XYAH), min_conf_(min_conf), max_cos_(max_cos), max_iou_(max_iou), n_init_(n_init),
        budget_(budget > 0 ? budget : 1), unbounded_(budget <= 0), lambda_(lambda), alpha_(alpha), max_age_(max_age) {
    // Track::Track :61-76: under GITHUB_ACTIONS=true (and outside the mot-metrics-benchmark job) a new track is Confirmed at once
    const char* ga = std::getenv("GITHUB_ACTIONS");
    const char* gj = std::getenv("GITHUB_JOB");
    born_confirmed_ = ga && std::string(ga) == "true" && (!gj || std::string(gj) != "mot-metrics-benchmark");
    core_.box_style = MOT_KF_BOX_TLWH_SUM;  // Track::to_tlbr :102-111
    if (unbounded_) throw Error("StrongSORT: nn_budget <= 0 (an unbounded sample library) is not supported on the device; give a budget");
  }
  ~StrongSortGpu() override {
    if (feat_) mot_free(core_.dev().ctx, feat_);
    if (samp_) mot_free(core_.dev().ctx, samp_);
  }
  Core& core() override { return core_; }
  void reset() override { tracks_.clear(); next_id_ = 1; core_.clear_slots(); }  // Tracker::reset :812-816
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : tracks_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }
  const float* feature_slab(int* dim, std::vector<char>* has) const override {
    *dim = D_;
    for (const Trk& t : tracks_) has->push_back(t.has_feat ? 1 : 0);
    return feat_;
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    stage_ = 0;
    // detections with conf >= min_conf, in input order (:873-877); det_ind = the row in the caller's matrix
    keep_.clear();
    for (int i = 0; i < in.n; ++i)
      if (in.dets[static_cast<size_t>(4) * in.ld + i] >= min_conf_) keep_.push_back(i);
    nd_ = static_cast<int>(keep_.size());
    conf_.resize(nd_); cls_.resize(nd_);
    std::vector<float> cm(static_cast<size_t>(6) * (nd_ > 0 ? nd_ : 1));
    for (int j = 0; j < nd_; ++j) {
      for (int k = 0; k < 6; ++k) cm[static_cast<size_t>(k) * nd_ + j] = in.dets[static_cast<size_t>(k) * in.ld + keep_[j]];
      conf_[j] = cm[static_cast<size_t>(4) * nd_ + j];
      cls_[j] = static_cast<int>(cm[static_cast<size_t>(5) * nd_ + j]);
    }
    core_.reserve(nd_ + 8, 8);
    dets_ = core_.upload_dets(cm.data(), nd_, nd_, MOT_DET_TLWH);
    // features of the kept detections, re-normalised rows (cosine_distance :317-331 = Track::update's feat / |feat| :167-171)
    have_emb_ = nd_ > 0 && (in.embs != nullptr) && in.emb_dim > 0;
    feat_ok_.assign(nd_, 0);
    if (have_emb_) {
      if (D_ == 0) D_ = in.emb_dim;
      if (D_ != in.emb_dim) throw Error("StrongSORT: embedding dimension changed between frames");
      ensure_slabs();
      Span<float> raw = core_.dev().up->alloc<float>(static_cast<size_t>(nd_) * D_);
      for (int j = 0; j < nd_; ++j) {
        float* dst = raw.h + static_cast<size_t>(j) * D_;
        const int i = keep_[j];
        double ss = 0.0;
        for (int k = 0; k < D_; ++k) {
          const float v = in.embs_rowmajor ? in.embs[static_cast<size_t>(i) * in.emb_ld + k] : in.embs[static_cast<size_t>(k) * in.emb_ld + i];
          dst[k] = v;
          ss += static_cast<double>(v) * v;
        }
        feat_ok_[j] = std::sqrt(ss) > 1e-10 ? 1 : 0;  // (:84, :168: a zero feature is no feature; decided here, the rows themselves stay on the device)
      }
      emb_norm_ = core_.dev().tmp->alloc<float>(static_cast<size_t>(nd_) * D_).d;
      mot_feat_task t{};
      t.n = nd_; t.d = D_; t.feat = emb_norm_; t.ldf = D_; t.src = raw.d; t.lds = D_; t.mode = 5; t.alpha = 0.f;
      core_.dev().q().feat_set.push_back(t);
    }
    // Tracker::predict :595-599
    const int nt = static_cast<int>(tracks_.size());
    if (nt > 0) {
      std::vector<int> slots(nt);
      for (int i = 0; i < nt; ++i) { slots[i] = tracks_[i].slot; ++tracks_[i].age; ++tracks_[i].tsu; }
      core_.predict(slots, nullptr, nullptr, nullptr);
    }
    // stage A: matching_cascade over the confirmed tracks (:729-741) — all tracks when there is none
    ta_.clear();
    for (int i = 0; i < nt; ++i) if (tracks_[i].state == Confirmed) ta_.push_back(i);
    confirmed_empty_ = ta_.empty();
    if (ta_.empty()) for (int i = 0; i < nt; ++i) ta_.push_back(i);
    lapA_ = Core::Lap();
    a_const_ = false;
    if (!ta_.empty() && nd_ > 0) {
      if (!have_emb_) a_const_ = true;  // :688-690: no feature in the frame — a constant 1e5 matrix, not even gated: nothing can match
      else queue_stage_a();
    }
  }

  bool advance() override {
    if (stage_ == 0) { after_a(); stage_ = 1; return true; }
    if (stage_ == 1) { after_b(); stage_ = 2; return true; }
    if (stage_ == 2) { emit(); stage_ = 3; }
    return false;
  }

 private:
  void ensure_slabs() {
    const int need = core_.cap();
    if (slab_cap_ >= need) return;
    Device& dv = core_.dev();
    void *nf = nullptr, *ns = nullptr;
    const size_t fb = sizeof(float) * static_cast<size_t>(need) * D_, sb = fb * budget_;
    dv.check(mot_malloc(dv.ctx, fb, &nf), "feature slab alloc");
    dv.check(mot_malloc(dv.ctx, sb, &ns), "sample slab alloc");
    if (feat_) {
      dv.check(mot_memcpy_d2d(dv.ctx, nf, feat_, sizeof(float) * static_cast<size_t>(slab_cap_) * D_), "feature slab copy");
      dv.check(mot_memcpy_d2d(dv.ctx, ns, samp_, sizeof(float) * static_cast<size_t>(slab_cap_) * D_ * budget_), "sample slab copy");
      dv.check(mot_ctx_sync(dv.ctx), "slab sync");
      mot_free(dv.ctx, feat_); mot_free(dv.ctx, samp_);
    }
    feat_ = static_cast<float*>(nf); samp_ = static_cast<float*>(ns);
    slab_cap_ = need;
  }

  // gated_metric :666-718: nearest-sample cosine distance, gate_cost_matrix, then min_cost_matching's clamp, then the assignment
  void queue_stage_a() {
    Device& dv = core_.dev();
    const int n = static_cast<int>(ta_.size()), m = nd_;
    std::vector<int> soff(n + 1, 0), rows, slots(n);
    for (int i = 0; i < n; ++i) {
      const Trk& t = tracks_[ta_[i]];
      slots[i] = t.slot;
      for (int k = 0; k < t.n_samples; ++k) rows.push_back(t.slot * budget_ + k);  // (the minimum does not care about the ring's order)
      soff[i + 1] = static_cast<int>(rows.size());
    }
    const int R = static_cast<int>(rows.size());
    const int ld = round_up(m, 4);
    float* cost = dv.tmp->alloc<float>(static_cast<size_t>(n) * ld).d;
    mot_ss_nn_task nn{};
    nn.n = n; nn.m = m; nn.soff = core_.ints(soff).d; nn.cost = cost; nn.ldc = ld; nn.ldd = ld; nn.dots = cost;  // (R == 0: never read)
    if (R > 0) {
      float* dots = dv.tmp->alloc<float>(static_cast<size_t>(R) * ld).d;
      mot_cos_task c{};
      c.n = R; c.m = m; c.d = D_; c.a = samp_; c.lda = D_; c.aidx = core_.ints(rows).d; c.b = emb_norm_; c.ldb = D_; c.out = dots; c.ldo = ld;
      dv.q().dot.push_back(c);
      nn.dots = dots;
    }
    dv.q().ss_nn.push_back(nn);
    mot_gate_task g{};
    g.n = n; g.m = m; g.mean = core_.d_mean(); g.src = core_.ints(slots).d; g.meas = dets_.d_meas; g.ldm = dets_.n;
    g.cost = cost; g.ldc = ld; g.out = cost; g.ldo = ld; g.mode = MOT_GATE_STRONGSORT | MOT_GATE_CLAMP; g.only_position = 0; g.metric = 0;
    g.lambda = lambda_; g.gated_cost = 1e5f; g.clamp_above = max_cos_;
    dv.q().gate.push_back(g);
    lapA_ = core_.lap(cost, ld, n, m, max_cos_);
  }

  void after_a() {
    const int n = static_cast<int>(ta_.size()), m = nd_;
    // min_cost_matching :343-420 on stage A's rows
    ut_a_.clear(); ud_a_.clear(); match_a_.clear();
    std::vector<int> x(n, -1), y(m, -1);
    if (lapA_.queued) { record(lapA_); x.assign(lapA_.x.h, lapA_.x.h + n); y.assign(lapA_.y.h, lapA_.y.h + m); }
    else if (a_const_ && record_laps) laps_.push_back(LapRecord{x, y});  // (the reference still solves the constant matrix: nothing matches)
    if (n == 0 || m == 0) {  // :370-372: returned before any cost: the lists as they came in (empty ones replaced by "all")
      for (int i = 0; i < n; ++i) ut_a_.push_back(ta_[i]);
      for (int j = 0; j < m; ++j) ud_a_.push_back(j);
    } else {
      for (int i = 0; i < n; ++i) {
        if (x[i] >= 0) match_a_.push_back({ta_[i], x[i]});  // (a matched pair's cost is below the threshold: the solver only pairs those)
        else ut_a_.push_back(ta_[i]);
      }
      for (int j = 0; j < m; ++j) if (y[j] < 0) ud_a_.push_back(j);
    }
    // candidates of the IoU stage :745-757
    tb_.clear(); ua_rest_.clear();
    const int nt = static_cast<int>(tracks_.size());
    for (int i = 0; i < nt; ++i) if (tracks_[i].state != Confirmed) tb_.push_back(i);
    for (int k : ut_a_) (tracks_[k].tsu == 1 ? tb_ : ua_rest_).push_back(k);
    if (tb_.empty()) for (int i = 0; i < nt; ++i) tb_.push_back(i);     // :358-361
    db_ = ud_a_;
    if (db_.empty()) for (int j = 0; j < m; ++j) db_.push_back(j);      // :362-365
    lapB_ = Core::Lap();
    if (!tb_.empty() && !db_.empty()) {
      Device& dv = core_.dev();
      const int nb = static_cast<int>(tb_.size()), mb = static_cast<int>(db_.size());
      std::vector<int> slots(nb);
      std::vector<uint8_t> stale(nb);
      for (int i = 0; i < nb; ++i) { slots[i] = tracks_[tb_[i]].slot; stale[i] = tracks_[tb_[i]].tsu > 1 ? 1 : 0; }
      const int ld = round_up(mb, 4);
      float* cost = dv.tmp->alloc<float>(static_cast<size_t>(nb) * ld).d;
      mot_ss_iou_task t{};
      t.n = nb; t.m = mb; t.mean = core_.d_mean(); t.src = core_.ints(slots).d; t.stale = core_.bytes(stale).d;
      t.dtlwh = dets_.d_box; t.ldd = dets_.n; t.didx = core_.ints(db_).d; t.cost = cost; t.ldc = ld; t.max_dist = max_iou_;
      dv.q().ss_iou.push_back(t);
      lapB_ = core_.lap(cost, ld, nb, mb, max_iou_);
    }
  }

  void after_b() {
    const int nb = static_cast<int>(tb_.size()), mb = static_cast<int>(db_.size());
    std::vector<std::pair<int, int>> match_b;
    std::vector<int> ut_b, ud_b;
    if (lapB_.queued) {
      record(lapB_);
      for (int i = 0; i < nb; ++i) {
        const int j = lapB_.x.h[i];
        if (j >= 0) match_b.push_back({tb_[i], db_[j]});
        else ut_b.push_back(tb_[i]);
      }
      for (int j = 0; j < mb; ++j) if (lapB_.y.h[j] < 0) ud_b.push_back(db_[j]);
    } else {
      ut_b = tb_; ud_b = db_;
    }
    // :778-806: stage A's matches, then stage B's that repeat neither a track nor a detection
    std::vector<std::pair<int, int>> matches = match_a_;
    IdSet& mt = set_a_;
    IdSet& md = set_b_;
    mt.clear(); md.clear();
    for (const auto& p : match_a_) { mt.insert(p.first); md.insert(p.second); }
    for (const auto& p : match_b)
      if (!mt.count(p.first) && !md.count(p.second)) { matches.push_back(p); mt.insert(p.first); md.insert(p.second); }
    std::vector<char> missed(tracks_.size(), 0);
    for (int k : ua_rest_) missed[k] = 1;
    for (int k : ut_b) missed[k] = 1;
    // Tracker::update :601-651
    upd_slot_.clear(); upd_meas_.clear(); ema_slot_.clear(); ema_det_.clear(); set_slot_.clear(); set_det_.clear();
    for (const auto& p : matches) {  // Track::update :147-187
      Trk& t = tracks_[p.first];
      const int j = p.second;
      t.conf = conf_[j]; t.cls = cls_[j]; t.det_ind = keep_[j];
      upd_slot_.push_back(t.slot); upd_meas_.push_back(j);
      if (have_emb_ && feat_ok_[j]) {
        (t.has_feat ? ema_slot_ : set_slot_).push_back(t.slot);
        (t.has_feat ? ema_det_ : set_det_).push_back(j);
        t.has_feat = true;
      }
      ++t.hits; t.tsu = 0;
      if (t.state == Tentative && t.hits >= n_init_) t.state = Confirmed;
    }
    for (size_t k = 0; k < tracks_.size(); ++k) {  // mark_missed :189-197, ascending track index (a std::set in the reference)
      if (!missed[k]) continue;
      Trk& t = tracks_[k];
      if (t.state == Tentative) t.state = Deleted;
      else if (t.tsu > max_age_) t.state = Deleted;
    }
    std::vector<int> init_dst, init_meas;
    for (int j : ud_b) {  // initiate_track :808-810
      Trk t;
      t.id = next_id_++;
      t.slot = core_.new_slot();
      t.state = born_confirmed_ ? Confirmed : Tentative;
      t.conf = conf_[j]; t.cls = cls_[j]; t.det_ind = keep_[j];
      init_dst.push_back(t.slot); init_meas.push_back(j);
      if (have_emb_ && feat_ok_[j]) { set_slot_.push_back(t.slot); set_det_.push_back(j); t.has_feat = true; }
      tracks_.push_back(t);
    }
    std::vector<Trk> alive;
    for (const Trk& t : tracks_) {
      if (t.state == Deleted) dead_.push_back(t.slot);
      else alive.push_back(t);
    }
    tracks_ = std::move(alive);
    core_.initiate(init_dst, init_meas, dets_);
    queue_update();
    if (D_ > 0) {
      if (core_.cap() > slab_cap_) ensure_slabs();
      queue_feat(set_slot_, set_det_, 6, core_.dev().q().feat_set);   // a first feature is the normalised detection feature as it is (:84-91, :180-182)
      queue_feat(ema_slot_, ema_det_, 4, core_.dev().q().feat_ema);   // :172-179
      // partial_fit :203-237 with the confirmed tracks' smoothed features (:627-650): one more sample per frame, the last nn_budget kept
      bool any = false;
      for (const Trk& t : tracks_) any = any || (t.state == Confirmed && t.has_feat);
      if (any) {
        std::vector<int> dst, src;
        for (Trk& t : tracks_) {
          if (t.state != Confirmed) { t.n_samples = 0; t.head = 0; continue; }  // (samples of tracks that are not active targets are dropped :226-234)
          if (!t.has_feat) continue;
          dst.push_back(t.slot * budget_ + t.head);
          src.push_back(t.slot);
          t.head = (t.head + 1) % budget_;
          if (t.n_samples < budget_) ++t.n_samples;
        }
        if (!dst.empty()) {
          mot_feat_task f{};
          f.n = static_cast<int>(dst.size()); f.d = D_; f.feat = samp_; f.ldf = D_; f.slot = core_.ints(dst).d; f.src = feat_; f.lds = D_; f.sidx = core_.ints(src).d;
          f.mode = 5; f.alpha = 0.f;  // stored re-normalised: cosine_distance normalises the samples on every call (:317-324), always to the same rows
          core_.dev().q().feat_late.push_back(f);
        }
      }
    }
    // rows to emit :976-994
    out_idx_.clear();
    std::vector<int> slots;
    for (size_t i = 0; i < tracks_.size(); ++i)
      if (tracks_[i].state == Confirmed && tracks_[i].tsu < 1) { out_idx_.push_back(static_cast<int>(i)); slots.push_back(tracks_[i].slot); }
    obox_ = Span<float>();
    core_.boxes(slots, &obox_);
  }
  void queue_update() {  // kf.update(mean, covariance, bbox, conf): the NSA rule takes the detection's confidence (:153)
    const int n = static_cast<int>(upd_slot_.size());
    if (n == 0) return;
    Span<int32_t> s = core_.ints(upd_slot_), m = core_.ints(upd_meas_);
    mot_kf_task t{};
    t.mean = core_.d_mean(); t.cov = core_.d_cov(); t.cap = core_.cap(); t.n = n; t.src = s.d; t.dst = s.d; t.meas = dets_.d_meas; t.ldm = dets_.n; t.midx = m.d;
    t.conf = dets_.d_conf();
    core_.dev().q().kf_upd[MOT_KF_XYAH].push_back(t);
  }
  void queue_feat(const std::vector<int>& slots, const std::vector<int>& dets, int mode, std::vector<mot_feat_task>& list) {
    if (slots.empty()) return;
    mot_feat_task t{};
    t.n = static_cast<int>(slots.size()); t.d = D_; t.feat = feat_; t.ldf = D_; t.slot = core_.ints(slots).d; t.src = emb_norm_; t.lds = D_; t.sidx = core_.ints(dets).d;
    t.mode = mode; t.alpha = alpha_;
    list.push_back(t);
  }
  void emit() {
    const int n = static_cast<int>(out_idx_.size());
    for (int k = 0; k < n; ++k) {
      const Trk& t = tracks_[out_idx_[k]];
      push_row(obox_.h, n, k, t.id, t.conf, t.cls, t.det_ind);
    }
    for (int s : dead_) core_.release_slot(s);
    dead_.clear();
  }

  Core core_;
  float min_conf_, max_cos_, max_iou_;
  int n_init_, budget_;
  bool unbounded_;
  float lambda_, alpha_;
  int max_age_;
  bool born_confirmed_ = false;
  int next_id_ = 1, stage_ = 0, D_ = 0, slab_cap_ = 0, nd_ = 0;
  bool have_emb_ = false, a_const_ = false, confirmed_empty_ = false;
  float* feat_ = nullptr;   // [slot][D] smoothed features
  float* samp_ = nullptr;   // [slot][budget][D] sample library, rows re-normalised
  float* emb_norm_ = nullptr;
  std::vector<Trk> tracks_;
  std::vector<int> keep_, cls_, ta_, tb_, db_, ut_a_, ud_a_, ua_rest_, out_idx_, dead_;
  std::vector<int> upd_slot_, upd_meas_, ema_slot_, ema_det_, set_slot_, set_det_;
  std::vector<std::pair<int, int>> match_a_;
  std::vector<float> conf_;
  std::vector<char> feat_ok_;
  IdSet set_a_, set_b_;
  Core::Dets dets_;
  Span<float> obox_;
  Core::Lap lapA_, lapB_;
};

}  // namespace

Staged* make_strongsort(std::shared_ptr<Device> dev, float min_conf, float max_cos_dist, float max_iou_dist, int n_init, int nn_budget,
                        float mc_lambda, float ema_alpha, int max_age) {
  return new StrongSortGpu(std::move(dev), min_conf, max_cos_dist, max_iou_dist, n_init, nn_budget, mc_lambda, ema_alpha, max_age);
}

}  // namespace motcpp::rt
