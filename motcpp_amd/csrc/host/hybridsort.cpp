// HybridSORT on the MI355X hot path: host lifecycle of src/trackers/hybridsort.cpp:825-1262 (HybridSort::update) as the reference runs it:
// its association functions are the "simplified" ones (:645-823) — the four corner velocities and the k-th previous observations are
// gathered and ignored — so a frame is three assignments on IoU-family costs against each track's LAST OBSERVED box (get_bbox
// :364-369; the Kalman box only while a track has none), and a Kalman update with an ALL-ZERO measurement for every track left
// unmatched (:1181-1188 -> :315-320). The nine-state filter, the pairwise cost matrices and the assignments run on the device
// (mot_hyb_task, csrc/hybrid_kernels.hip); states are 90-float records in this tracker's slab.
// Built: with_reid = false, and with_reid = true without embeddings (the reference's all-zero features, :868-871: every appearance
// distance is 1 — the first association's costs and threshold carry + EG_weight_high_score, the BYTE step's costs + EG_weight_low_score,
// which no BYTE pair survives). Embeddings are refused; the ECC camera-motion step is outside the path.
//
// Stages: 0 predict every track, Kalman boxes of the tracks without an observation | 1 first assignment | 2 BYTE assignment |
//         3 last-chance assignment | 4 filter updates (matched pairs; zero measurements), births, boxes of new tracks to report.
#include <cmath>
#include <string>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

struct Trk {
  int id = 0, slot = -1, age = 0, hits = 0, hit_streak = 0, tsu = 0, cls = 0, det_ind = -1;
  float conf = 0.f, conf_pre = 0.f;
  float last[4] = {-1.f, -1.f, -1.f, -1.f};
  bool no_obs() const { return ((last[0] + last[1]) + last[2]) + last[3] < 0; }  // last_observation_.head<4>().sum() < 0
};

class HybridSortGpu final : public Staged {
 public:
  HybridSortGpu(std::shared_ptr<Device> dev, const HybridParams& p) : core_(std::move(dev), MOT_KF_XYAH), p_(p) {}
  ~HybridSortGpu() override { if (slab_) mot_free(core_.dev().ctx, slab_); }
  Core& core() override { return core_; }
  void reset() override { tracks_.clear(); free_.clear(); next_slot_ = 0; frame_count_ = 0; next_id_ = 0; }  // :478-482
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : tracks_) { ids->push_back(t.id + 1); slots->push_back(t.slot); }
  }
  // parity hook: rows [id + 1, x(9), P(81)] in list order
  bool f32_states(std::vector<float>* rows, int* width) override {
    Device& dv = core_.dev();
    std::lock_guard<std::mutex> lk(dv.frame_mu);
    *width = 91;
    rows->clear();
    std::vector<float> h(static_cast<size_t>(next_slot_) * 90);
    if (next_slot_ > 0) {
      dv.check(mot_memcpy_d2h(dv.ctx, h.data(), slab_, h.size() * sizeof(float)), "hybridsort state download");
      dv.check(mot_ctx_sync(dv.ctx), "sync");
    }
    for (const Trk& t : tracks_) {
      rows->push_back(static_cast<float>(t.id + 1));
      for (int k = 0; k < 90; ++k) rows->push_back(h[static_cast<size_t>(t.slot) * 90 + k]);
    }
    return true;
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    ++frame_count_;
    if (in.embs != nullptr && in.emb_dim > 0 && in.n > 0)
      throw Error("HybridSORT: the ReID branch is not built (run without embeddings: with_reid = false, or its zero-feature behaviour)");
    Device& dv = core_.dev();
    n_ = in.n;
    stage_ = (n_ == 0) ? 5 : 0;
    raw_ = nullptr;
    keep_.clear(); second_.clear();
    if (n_ > 0) {
      Span<float> raw = dv.up->alloc<float>(static_cast<size_t>(6) * n_);
      for (int k = 0; k < 6; ++k)
        for (int i = 0; i < n_; ++i) raw.h[static_cast<size_t>(k) * n_ + i] = in.dets[static_cast<size_t>(k) * in.ld + i];
      raw_ = raw.d; hraw_ = raw.h;
      for (int i = 0; i < n_; ++i) {  // :884-886
        const float c = hraw_[static_cast<size_t>(4) * n_ + i];
        if (c > p_.low_thresh && c < p_.det_thresh) second_.push_back(i);
        if (c > p_.det_thresh) keep_.push_back(i);
      }
    }
    ensure_slab(static_cast<int>(keep_.size()));
    const int nt = static_cast<int>(tracks_.size());
    kbox_ = Span<float>(); kidx_.clear();
    if (nt > 0) {
      std::vector<int> slots(nt), noobs;
      for (int j = 0; j < nt; ++j) {  // HybridKalmanBoxTracker::predict :256-270 (the filter part runs on the device)
        Trk& t = tracks_[j];
        slots[j] = t.slot;
        ++t.age;
        if (t.tsu > 0) t.hit_streak = 0;
        ++t.tsu;
        if (n_ > 0 && t.no_obs()) { kidx_.push_back(j); noobs.push_back(t.slot); }
      }
      mot_hyb_task t{};
      t.n = nt; t.slab = slab_; t.slots = core_.ints(slots).d;
      dv.q().hyb[MOT_HYB_PREDICT].push_back(t);
      if (!noobs.empty()) {
        kbox_ = dv.down->alloc<float>(noobs.size() * 4);
        mot_hyb_task b{};
        b.n = static_cast<int>(noobs.size()); b.slab = slab_; b.slots = core_.ints(noobs).d; b.boxes = kbox_.d;
        dv.q().hyb[MOT_HYB_BOXES].push_back(b);
      }
    }
    if (n_ == 0) drop_dead();  // :832-846: an empty frame predicts, drops the dead tracks and reports nothing
  }

  bool advance() override {
    switch (stage_) {
      case 0: after_predict(); stage_ = 1; return true;
      case 1: after_first(); stage_ = 2; return true;
      case 2: after_byte(); stage_ = 3; return true;
      case 3: after_last(); stage_ = 4; return true;
      case 4: emit(); stage_ = 5; return false;
      default: return false;
    }
  }

 private:
  struct Pair {
    Span<float> sim;
    Core::Lap lap;
    int n = 0, m = 0, ld = 0;
    bool queued = false;
  };
  void ensure_slab(int births) {
    const int need = next_slot_ + births + 8;
    if (need <= cap_) return;
    Device& dv = core_.dev();
    int ncap = cap_ > 0 ? cap_ : 64;
    while (ncap < need) ncap *= 2;
    void* ns = nullptr;
    dv.check(mot_malloc(dv.ctx, sizeof(float) * 90 * ncap, &ns), "hybridsort slab alloc");
    if (slab_) {
      dv.check(mot_memcpy_d2d(dv.ctx, ns, slab_, sizeof(float) * 90 * next_slot_), "hybridsort slab copy");
      dv.check(mot_ctx_sync(dv.ctx), "slab sync");
      mot_free(dv.ctx, slab_);
    }
    slab_ = static_cast<float*>(ns);
    cap_ = ncap;
  }
  int take_slot() {
    if (!free_.empty()) { const int s = free_.back(); free_.pop_back(); return s; }
    return next_slot_++;
  }
  float det(int k, int i) const { return hraw_[static_cast<size_t>(k) * n_ + i]; }
  // similarity (downloaded) + cost (device) + assignment of detections `rows` (original indices) against boxes / scores of tracks
  Pair pair(const std::vector<int>& rows, const std::vector<std::array<float, 4>>& tbox, const std::vector<float>* tscore, bool hm, float score_w,
            float add_const, bool scale_first, float thresh) {
    Device& dv = core_.dev();
    Pair q;
    q.n = static_cast<int>(rows.size()); q.m = static_cast<int>(tbox.size()); q.ld = round_up(q.m, 4);
    Span<float> a = dv.up->alloc<float>(static_cast<size_t>(q.n) * 4), as = dv.up->alloc<float>(q.n);
    Span<float> b = dv.up->alloc<float>(static_cast<size_t>(q.m) * 4), bs = dv.up->alloc<float>(q.m);
    for (int i = 0; i < q.n; ++i) {
      for (int k = 0; k < 4; ++k) a.h[static_cast<size_t>(i) * 4 + k] = det(k, rows[i]);
      as.h[i] = det(4, rows[i]);
    }
    for (int j = 0; j < q.m; ++j) {
      for (int k = 0; k < 4; ++k) b.h[static_cast<size_t>(j) * 4 + k] = tbox[j][k];
      bs.h[j] = tscore ? (*tscore)[j] : 0.0f;
    }
    q.sim = dv.down->alloc<float>(static_cast<size_t>(q.n) * q.ld);
    float* cost = dv.tmp->alloc<float>(static_cast<size_t>(q.n) * q.ld).d;
    mot_hyb_task t{};
    t.n = q.n; t.m = q.m; t.ldc = q.ld; t.a = a.d; t.b = b.d; t.a_score = as.d; t.b_score = bs.d; t.sim = q.sim.d; t.cost = cost;
    t.hmiou = hm ? 1 : 0; t.score_w = score_w; t.add_const = add_const; t.scale_first = scale_first ? 1 : 0;
    dv.q().hyb[MOT_HYB_PAIR].push_back(t);
    q.lap = core_.lap(cost, q.ld, q.n, q.m, thresh);
    q.queued = true;
    return q;
  }
  float sim_max(const Pair& q) const {  // maxCoeff() of the similarity matrix
    float mx = 0.0f;
    for (int i = 0; i < q.n; ++i)
      for (int j = 0; j < q.m; ++j) {
        const float v = q.sim.h[static_cast<size_t>(i) * q.ld + j];
        if ((i == 0 && j == 0) || v > mx) mx = v;
      }
    return mx;
  }
  void matched(int trk, int d) {  // HybridKalmanBoxTracker::update with a box :272-313 (the filter update is queued for stage 4)
    Trk& t = tracks_[trk];
    for (int k = 0; k < 4; ++k) t.last[k] = det(k, d);
    t.tsu = 0; ++t.hits; ++t.hit_streak;
    t.cls = static_cast<int>(det(5, d)); t.det_ind = d;
    t.conf_pre = t.conf; t.conf = det(4, d);
    upd_slots_.push_back(t.slot); upd_dets_.push_back(d);
  }

  void after_predict() {
    const int nt = static_cast<int>(tracks_.size());
    tbox_.assign(nt, {}); last_.assign(nt, {}); tscore_.assign(nt, 0.f);
    auto clampf = [](float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); };
    size_t kb = 0;
    for (int j = 0; j < nt; ++j) {
      const Trk& t = tracks_[j];
      for (int k = 0; k < 4; ++k) last_[j][k] = t.last[k];
      if (kb < kidx_.size() && kidx_[kb] == j) { for (int k = 0; k < 4; ++k) tbox_[j][k] = kbox_.h[kb * 4 + k]; ++kb; }  // get_bbox :364-369
      else tbox_[j] = last_[j];
      tscore_[j] = (t.conf_pre == 0.0f) ? clampf(t.conf, 0.1f, p_.track_thresh) : clampf(t.conf - (t.conf_pre - t.conf), 0.1f, p_.track_thresh);  // :376-381
    }
    upd_slots_.clear(); upd_dets_.clear();
    ud_.clear(); ut_.clear();
    first_ = Pair();
    if (p_.tcm_first && !keep_.empty() && nt > 0) {
      const bool zero_reid = p_.with_reid && p_.eg_high > 0;
      const float thr = zero_reid ? (1.0f - p_.iou_threshold) * 1.0f + p_.eg_high : 1.0f - p_.iou_threshold;
      first_ = pair(keep_, tbox_, nullptr, p_.asso == 1, 0.0f, zero_reid ? 1.0f * p_.eg_high : 0.0f, zero_reid, thr);
    }
  }

  void after_first() {
    const int nt = static_cast<int>(tracks_.size()), nd = static_cast<int>(keep_.size());
    if (first_.queued) {  // associate_4_points_with_score :645-716
      record(first_.lap);
      std::vector<char> dm(nd, 0), tm(nt, 0);
      std::vector<std::pair<int, int>> ok;
      for (int i = 0; i < nd; ++i) {
        const int j = first_.lap.x.h[i];
        if (j < 0) continue;
        if (first_.sim.h[static_cast<size_t>(i) * first_.ld + j] >= p_.iou_threshold) { ok.push_back({i, j}); dm[i] = 1; tm[j] = 1; }
        else { ud_.push_back(i); ut_.push_back(j); }
      }
      // (as written, :681-691: a pair rejected by the IoU check is on the lists already and is appended once more here)
      for (int i = 0; i < nd; ++i) if (!dm[i]) ud_.push_back(i);
      for (int j = 0; j < nt; ++j) if (!tm[j]) ut_.push_back(j);
      for (const auto& m : ok) matched(m.second, keep_[m.first]);
    } else {
      for (int i = 0; i < nd; ++i) ud_.push_back(i);
      for (int j = 0; j < nt; ++j) ut_.push_back(j);
    }
    byte_ = Pair();
    if (p_.use_byte && !second_.empty() && !ut_.empty()) {  // :1052-1065
      std::vector<std::array<float, 4>> ub(ut_.size());
      std::vector<float> us(ut_.size());
      for (size_t k = 0; k < ut_.size(); ++k) { ub[k] = tbox_[ut_[k]]; us[k] = tscore_[ut_[k]]; }
      const bool zero_reid = p_.with_reid && p_.eg_low > 0;
      byte_ = pair(second_, ub, &us, false, p_.tcm_byte ? p_.tcm_byte_weight : 0.0f, zero_reid ? 1.0f * p_.eg_low : 0.0f, false, 1.0f - p_.iou_threshold);
    }
  }

  void after_byte() {
    if (byte_.queued && sim_max(byte_) > p_.iou_threshold) {  // :1067-1127
      record(byte_.lap);
      std::vector<char> gone(tracks_.size(), 0);
      for (int i = 0; i < byte_.n; ++i) {
        const int j = byte_.lap.x.h[i];
        if (j >= 0 && byte_.sim.h[static_cast<size_t>(i) * byte_.ld + j] >= p_.iou_threshold) { matched(ut_[j], second_[i]); gone[ut_[j]] = 1; }
      }
      std::vector<int> rest;
      for (int j : ut_) if (!gone[j]) rest.push_back(j);
      ut_.swap(rest);
    }
    last_pair_ = Pair();
    if (!ud_.empty() && !ut_.empty()) {  // :1130-1143: the unmatched detections against the LAST OBSERVED boxes of the unmatched tracks
      std::vector<int> rows(ud_.size());
      for (size_t k = 0; k < ud_.size(); ++k) rows[k] = keep_[ud_[k]];
      std::vector<std::array<float, 4>> lb(ut_.size());
      for (size_t k = 0; k < ut_.size(); ++k) lb[k] = last_[ut_[k]];
      last_pair_ = pair(rows, lb, nullptr, false, 0.0f, 0.0f, false, 1.0f - p_.iou_threshold);
    }
  }

  void after_last() {
    Device& dv = core_.dev();
    if (last_pair_.queued && sim_max(last_pair_) > p_.iou_threshold) {  // :1144-1178
      record(last_pair_.lap);
      std::vector<char> dgone(keep_.size(), 0), tgone(tracks_.size(), 0);
      for (int i = 0; i < last_pair_.n; ++i) {
        const int j = last_pair_.lap.x.h[i];
        if (j >= 0 && last_pair_.sim.h[static_cast<size_t>(i) * last_pair_.ld + j] >= p_.iou_threshold) {
          matched(ut_[j], keep_[ud_[i]]);
          dgone[ud_[i]] = 1; tgone[ut_[j]] = 1;
        }
      }
      std::vector<int> rd, rt;
      for (int i : ud_) if (!dgone[i]) rd.push_back(i);
      for (int j : ut_) if (!tgone[j]) rt.push_back(j);
      ud_.swap(rd); ut_.swap(rt);
    }
    for (int j : ut_) {  // :1181-1188: update(empty box) = a filter update with an all-zero measurement
      tracks_[j].conf_pre = 0.0f;
      upd_slots_.push_back(tracks_[j].slot); upd_dets_.push_back(-1);
    }
    if (!upd_slots_.empty()) {
      mot_hyb_task t{};
      t.n = static_cast<int>(upd_slots_.size()); t.slab = slab_; t.slots = core_.ints(upd_slots_).d; t.didx = core_.ints(upd_dets_).d; t.dets = raw_; t.ldd = n_;
      dv.q().hyb[MOT_HYB_UPDATE].push_back(t);
    }
    if (!ud_.empty()) {  // :1190-1210
      std::vector<int> slots(ud_.size()), dets(ud_.size());
      for (size_t k = 0; k < ud_.size(); ++k) {
        const int d = keep_[ud_[k]];
        Trk t;
        t.id = ++next_id_;  // next_id() :21-23 (the table shows id + 1)
        t.slot = take_slot(); t.conf = det(4, d); t.cls = static_cast<int>(det(5, d)); t.det_ind = d;
        slots[k] = t.slot; dets[k] = d;
        tracks_.push_back(t);
      }
      mot_hyb_task t{};
      t.n = static_cast<int>(ud_.size()); t.slab = slab_; t.slots = core_.ints(slots).d; t.didx = core_.ints(dets).d; t.dets = raw_; t.ldd = n_;
      dv.q().hyb[MOT_HYB_INIT].push_back(t);
    }
    // tracks to report that have no observation yet (new ones, while frame_count <= min_hits) show their Kalman box
    out_.clear(); obox_idx_.clear();
    std::vector<int> bslots;
    for (int j = static_cast<int>(tracks_.size()) - 1; j >= 0; --j) {  // :1212-1229: reverse order
      const Trk& t = tracks_[j];
      if (t.tsu < 1 && (t.hit_streak >= p_.min_hits || frame_count_ <= p_.min_hits)) {
        out_.push_back(j);
        if (t.no_obs()) { obox_idx_.push_back(static_cast<int>(out_.size()) - 1); bslots.push_back(t.slot); }
      }
    }
    obox_ = Span<float>();
    if (!bslots.empty()) {
      obox_ = dv.down->alloc<float>(bslots.size() * 4);
      mot_hyb_task b{};
      b.n = static_cast<int>(bslots.size()); b.slab = slab_; b.slots = core_.ints(bslots).d; b.boxes = obox_.d;
      dv.q().hyb[MOT_HYB_BOXES].push_back(b);
    }
  }

  void emit() {
    size_t kb = 0;
    for (size_t k = 0; k < out_.size(); ++k) {
      const Trk& t = tracks_[out_[k]];
      const float* b = t.last;
      if (kb < obox_idx_.size() && obox_idx_[kb] == static_cast<int>(k)) { b = obox_.h + kb * 4; ++kb; }
      rows_.push_back(b[0]); rows_.push_back(b[1]); rows_.push_back(b[2]); rows_.push_back(b[3]);
      rows_.push_back(static_cast<float>(t.id + 1)); rows_.push_back(t.conf);
      rows_.push_back(static_cast<float>(t.cls)); rows_.push_back(static_cast<float>(t.det_ind));
    }
    drop_dead();
  }
  void drop_dead() {  // :1231-1238
    size_t w = 0;
    for (size_t j = 0; j < tracks_.size(); ++j) {
      if (tracks_[j].tsu > p_.max_age) { free_.push_back(tracks_[j].slot); continue; }
      if (w != j) tracks_[w] = tracks_[j];
      ++w;
    }
    tracks_.resize(w);
  }

  Core core_;
  HybridParams p_;
  float* slab_ = nullptr;
  int cap_ = 0, next_slot_ = 0, next_id_ = 0, frame_count_ = 0;
  std::vector<int> free_;
  std::vector<Trk> tracks_;
  // per frame
  int stage_ = 0, n_ = 0;
  const float* raw_ = nullptr;
  const float* hraw_ = nullptr;
  std::vector<int> keep_, second_, kidx_, ud_, ut_, upd_slots_, upd_dets_, out_, obox_idx_;
  Span<float> kbox_, obox_;
  std::vector<std::array<float, 4>> tbox_, last_;
  std::vector<float> tscore_;
  Pair first_, byte_, last_pair_;
};

}  // namespace

Staged* make_hybridsort(std::shared_ptr<Device> dev, const HybridParams& p) { return new HybridSortGpu(std::move(dev), p); }

}  // namespace motcpp::rt
