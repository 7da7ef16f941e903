// OC-SORT on the MI355X hot path: host lifecycle of src/trackers/ocsort.cpp:285-738 (observation history,
// velocity directions, the Q4 duplicate-unmatched quirk, the NaN-row quirk of :360-364) with the numeric
// work on the device: XYSR Kalman predict(with the x6 clamp)/update/initiate, the IoU + velocity-direction
// cost matrix, the trivial-case shortcut + LAP, and the -IoU rematch LAPs with their max-IoU gates.
//
// Stages: 0 predict + first association | 1 (optional BYTE) | 2 OCR rematch | 3 Kalman updates / spawns.
#include <cmath>
#include <map>
#include <unordered_set>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

struct Obs5 { float v[5]; };

struct Trk {
  int id = 0, slot = -1, age = 0, hits = 0, hit_streak = 0, tsu = 0, cls = 0, det_ind = 0;
  float conf = 0.f;
  Obs5 last_obs{{-1, -1, -1, -1, -1}};
  // age -> observation. The reference keeps every entry (ocsort.cpp:118); only the last delta_t+1 ages and the newest
  // entry are ever looked up (:24-51), so a short age-ordered tail is equivalent.
  std::vector<std::pair<int, Obs5>> observations;
  float vel[2] = {0.f, 0.f};
};

class OCSortGpu final : public Staged {
 public:
  OCSortGpu(std::shared_ptr<Device> dev, float det_thresh, int max_age, int /*max_obs*/, int min_hits, float iou_threshold,
            float min_conf, int delta_t, float inertia, bool use_byte, float q_xy, float q_s, int asso)
      : core_(std::move(dev), MOT_KF_XYSR), det_thresh_(det_thresh), max_age_(max_age), min_hits_(min_hits), thr_(iou_threshold),
        min_conf_(min_conf), delta_t_(delta_t), inertia_(inertia), use_byte_(use_byte), asso_(asso) {
    core_.q[0] = 0.01f * q_xy;  // Q5: the tracker scales the constructor's already-scaled entries again (ocsort.cpp:77-79)
    core_.q[1] = 0.01f * q_xy;
    core_.q[2] = 0.0001f * q_s;
  }
  Core& core() override { return core_; }
  void reset() override { frame_count_ = 0; trk_.clear(); core_.clear_slots(); }
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : trk_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    ++frame_count_;
    stage_ = 0;
    // AssociationFunction(img_w, img_h, asso_func) is rebuilt every frame (ocsort.cpp:295-296,413): the centroid
    // measure is normalised by the frame diagonal, std::sqrt(int) -> double -> float (iou.hpp:329)
    frame_diag_ = static_cast<float>(std::sqrt(static_cast<double>(in.img_w * in.img_w + in.img_h * in.img_h)));
    high_.clear(); second_.clear();
    raw_.assign(static_cast<size_t>(6) * in.n, 0.f);
    n_ = in.n;
    for (int k = 0; k < 6; ++k)
      for (int i = 0; i < in.n; ++i) raw_[static_cast<size_t>(k) * in.n + i] = in.dets[static_cast<size_t>(k) * in.ld + i];
    for (int i = 0; i < in.n; ++i) {
      const float c = conf(i);
      if (c > min_conf_ && c < det_thresh_) second_.push_back(i);
      if (c > det_thresh_) high_.push_back(i);
    }
    const int nt = static_cast<int>(trk_.size());
    core_.reserve(2 * static_cast<int>(high_.size()) + 8, 8);
    dets_ = core_.upload_dets(in.dets, in.n, in.ld, MOT_DET_XYSR, in.d_dets, in.d_ld);
    pbox_ = Span<float>();
    nt0_ = nt;
    assoc_ = Core::Lap();
    if (nt > 0) {
      std::vector<int> slots(nt);
      std::vector<uint8_t> fl(nt, MOT_KF_OCSORT_CLAMP);
      for (int i = 0; i < nt; ++i) {  // KalmanBoxTracker::predict :132-148
        Trk& t = trk_[i];
        slots[i] = t.slot;
        ++t.age;
        if (t.tsu > 0) t.hit_streak = 0;
        ++t.tsu;
      }
      pbox_d_ = core_.predict(slots, nullptr, &fl, &pbox_);
      queue_first(nt);
    }
  }

  bool advance() override {
    while (true) {
      switch (stage_) {
        case 0: {
          const int nt = nt0_;
          std::vector<int> del;
          for (int i = 0; i < nt; ++i)
            if (std::isnan(pbox_.h[i]) || std::isnan(pbox_.h[nt + i]) || std::isnan(pbox_.h[2 * nt + i]) || std::isnan(pbox_.h[3 * nt + i])) del.push_back(i);
          if (!del.empty()) {
            for (auto it = del.rbegin(); it != del.rend(); ++it) { core_.release_slot(trk_[*it].slot); trk_.erase(trk_.begin() + *it); }
            assoc_ = Core::Lap();
            stage_ = 1;
            if (!trk_.empty()) { queue_first(static_cast<int>(trk_.size())); return true; }  // rows = FIRST nt' predicted boxes (:363-364)
            continue;
          }
          stage_ = 1;
          continue;
        }
        case 1: {
          if (trk_.empty()) {  // :366-383
            um_dets_.clear(); um_trks_.clear();
            for (int i = 0; i < static_cast<int>(high_.size()); ++i) um_dets_.push_back(i);
            upd_.clear();
            silent_ = true;
            stage_ = 4;
            continue;
          }
          silent_ = false;
          after_first();
          stage_ = 2;
          if (use_byte_ && !second_.empty() && !um_trks_.empty()) { queue_byte(); return true; }
          byte_ = Core::Lap();
          continue;
        }
        case 2: {
          if (byte_.queued) after_byte();
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

  static Obs5 k_previous_obs(const Trk& t, int k) {  // :24-51
    if (t.observations.empty()) return Obs5{{-1, -1, -1, -1, -1}};
    for (int i = 0; i < k; ++i) {
      const int key = t.age - (k - i);
      for (const auto& o : t.observations)
        if (o.first == key) return o.second;
    }
    return t.observations.back().second;
  }
  static void speed_direction(const float* b1, const float* b2, float out[2]) {  // :160-172
    const float cx1 = (b1[0] + b1[2]) / 2.0f, cy1 = (b1[1] + b1[3]) / 2.0f;
    const float cx2 = (b2[0] + b2[2]) / 2.0f, cy2 = (b2[1] + b2[3]) / 2.0f;
    const float dy = cy2 - cy1, dx = cx2 - cx1;
    const float norm = std::sqrt(dy * dy + dx * dx) + 1e-6f;
    out[0] = dy / norm; out[1] = dx / norm;
  }
  void apply(Trk& t, int det) {  // KalmanBoxTracker::update :89-130 (Kalman part queued)
    t.det_ind = det;
    t.conf = conf(det); t.cls = cls(det);
    float b[4];
    box(det, b);
    const float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
    if (ls >= 0) {
      const Obs5 pb = k_previous_obs(t, delta_t_);
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

  void queue_first(int nt) {
    // rows of `trks` are the first nt predicted boxes; velocities / k-previous observations come from the current tracks
    std::vector<float> vel(static_cast<size_t>(2) * nt), prev(static_cast<size_t>(5) * nt);
    for (int i = 0; i < nt; ++i) {
      vel[i] = trk_[i].vel[0]; vel[nt + i] = trk_[i].vel[1];
      const Obs5 k = k_previous_obs(trk_[i], delta_t_);
      for (int c = 0; c < 5; ++c) prev[static_cast<size_t>(c) * nt + i] = k.v[c];
    }
    const int nd = static_cast<int>(high_.size());
    if (nd == 0) { assoc_ = Core::Lap(); assoc_nt_ = nt; return; }
    Span<float> dv = core_.floats(vel), dp = core_.floats(prev);
    high_d_ = core_.ints(high_);
    const int ld = round_up(nt, 4);
    float* cost;
    {
      cost = core_.dev().tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
      iou_d_ = core_.dev().tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
      mot_ocsort_task t{};
      t.nd = nd; t.nt = nt; t.dbox = dets_.d_box; t.ldd = dets_.n; t.didx = high_d_.d; t.dconf = dets_.d_conf();
      t.tbox = pbox_d_; t.ldt = nt0_; t.vel = dv.d; t.ldv = nt; t.prev = dp.d; t.ldp = nt; t.vdc_weight = inertia_;
      t.cost = cost; t.iou = iou_d_; t.ldc = ld;
      t.assoc = asso_; t.frame_diag = frame_diag_;
      core_.dev().q().oc.push_back(t);
    }
    assoc_nt_ = nt;
    assoc_ = core_.lap(cost, ld, nd, nt, -thr_, MOT_LAP_OCSORT, iou_d_, ld, thr_, true);
  }

  void after_first() {  // ocsort_assoc::associate :610-738
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
        else { um_dets_.push_back(i); um_trks_.push_back(j); }  // Q4: swept in again below
      }
    }
    for (int i = 0; i < nd; ++i) if (!md[i]) um_dets_.push_back(i);
    for (int j = 0; j < nt; ++j) if (!mt[j]) um_trks_.push_back(j);
    for (const auto& m : matches) apply(trk_[m.second], high_[m.first]);
  }

  void queue_byte() {  // :430-472
    second_d_ = core_.ints(second_);
    um_trks_d_ = core_.ints(um_trks_);
    Core::IouArgs a;
    a.a = dets_.d_box; a.lda = dets_.n; a.aidx = second_d_.d; a.n = static_cast<int>(second_.size());
    a.b = pbox_d_; a.ldb = nt0_; a.bidx = um_trks_d_.d; a.m = static_cast<int>(um_trks_.size());
    a.mode = MOT_COST_NEG_IOU; a.assoc = asso_; a.frame_diag = frame_diag_;
    byte_ = core_.lap_geom(a, -thr_, MOT_LAP_GATE_MIN, -thr_, true);
  }
  void after_byte() {
    if (byte_.info.h[0] == 2) return;
    record(byte_);
    std::unordered_set<int> rm;
    for (int i = 0; i < byte_.n; ++i) {
      const int j = byte_.x.h[i];
      if (j < 0) continue;
      if (-byte_.xval.h[i] < thr_) continue;
      const int ti = um_trks_[j];
      apply(trk_[ti], second_[i]);
      rm.insert(ti);
    }
    std::vector<int> keep;
    for (int t : um_trks_) if (!rm.count(t)) keep.push_back(t);
    um_trks_ = keep;
  }
  void queue_rematch() {  // :475-540
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
    for (int t : um_trks_) trk_[t].det_ind = 0;  // "None" update (:543-545): det_ind = 0, Kalman no-op
    init_dst_.clear(); init_meas_.clear();
    for (int d : um_dets_) {  // spawn (:548-556); duplicates in um_dets_ spawn twice (Q4)
      Trk t;
      t.id = ++next_id_;
      t.slot = core_.new_slot();
      const int det = high_[d];
      t.conf = conf(det); t.cls = cls(det); t.det_ind = det;
      init_dst_.push_back(t.slot); init_meas_.push_back(det);
      trk_.push_back(t);
    }
    // Kalman updates, split into rounds when a slot is updated more than once in a frame (Q4 duplicates)
    std::map<int, int> seen;
    rounds_.clear();
    for (const auto& u : upd_) {
      const int r = seen[u.first]++;
      if (static_cast<int>(rounds_.size()) <= r) rounds_.resize(r + 1);
      rounds_[r].push_back(u);
    }
    n_rounds_ = std::max<int>(1, static_cast<int>(rounds_.size()));
    // rows to emit (:562-587, reverse order) and tracks whose box must come from the filter state
    emit_idx_.clear();
    std::vector<int> need_state;
    if (!silent_) {
      for (int i = static_cast<int>(trk_.size()) - 1; i >= 0; --i) {
        const Trk& t = trk_[i];
        if (t.tsu < 1 && (t.hit_streak >= min_hits_ || frame_count_ <= min_hits_)) {
          emit_idx_.push_back(i);
          const float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
          if (ls < 0) need_state.push_back(i);
        }
      }
    }
    need_state_ = need_state;
  }
  void queue_round() {
    if (round_ == 0) core_.initiate(init_dst_, init_meas_, dets_);
    if (round_ < static_cast<int>(rounds_.size())) {
      std::vector<int> s, m;
      for (const auto& u : rounds_[round_]) { s.push_back(u.first); m.push_back(u.second); }
      core_.update(s, s, m, dets_);
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
        const Trk& t = trk_[i];
        float b[4] = {t.last_obs.v[0], t.last_obs.v[1], t.last_obs.v[2], t.last_obs.v[3]};
        if (b[0] + b[1] + b[2] + b[3] < 0) {
          for (int k = 0; k < ns; ++k)
            if (need_state_[k] == i) for (int c = 0; c < 4; ++c) b[c] = sbox_.h[static_cast<size_t>(c) * ns + k];
        }
        push_row(b, 1, 0, t.id + 1, t.conf, t.cls, t.det_ind);
      }
      for (int i = static_cast<int>(trk_.size()) - 1; i >= 0; --i)
        if (trk_[i].tsu > max_age_) { core_.release_slot(trk_[i].slot); trk_.erase(trk_.begin() + i); }
    }
  }

  Core core_;
  float det_thresh_;
  int max_age_, min_hits_;
  float thr_, min_conf_;
  int delta_t_;
  float inertia_;
  bool use_byte_;
  int frame_count_ = 0, next_id_ = 0, stage_ = 0, n_ = 0, nt0_ = 0, assoc_nt_ = 0 , round_ = 0, n_rounds_ = 1;
  bool silent_ = false;
  std::vector<Trk> trk_;
  std::vector<float> raw_;
  std::vector<int> high_, second_, um_dets_, um_trks_, init_dst_, init_meas_, emit_idx_, need_state_;
  std::vector<std::pair<int, int>> upd_;
  std::vector<std::vector<std::pair<int, int>>> rounds_;
  Core::Dets dets_;
  Span<float> pbox_, sbox_;
  float* pbox_d_ = nullptr;
  float* iou_d_ = nullptr;
  int asso_ = MOT_ASSOC_IOU;
  float frame_diag_ = 1.f;
  Span<int32_t> high_d_, second_d_, um_trks_d_, left_d_;
  Core::Lap assoc_, byte_, rematch_;

};

}  // namespace

Staged* make_ocsort(std::shared_ptr<Device> dev, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold,
                    float min_conf, int delta_t, float inertia, bool use_byte, float q_xy, float q_s, int asso) {
  return new OCSortGpu(std::move(dev), det_thresh, max_age, max_obs, min_hits, iou_threshold, min_conf, delta_t, inertia, use_byte, q_xy, q_s, asso);
}

}  // namespace motcpp::rt
