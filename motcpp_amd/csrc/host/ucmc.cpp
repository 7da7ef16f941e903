// UCMCTrack on the MI355X hot path: host lifecycle of src/trackers/ucmc.cpp:261-572 (UCMCTrack::update, dataAssociation,
// associateTentative, initTentative, deleteOldTrackers, updateStatus) with the ground-plane filter on the device — the detections'
// mapping to the ground plane, the double-precision predict / Joseph update of [x, vx, y, vy], the Mahalanobis + log-determinant cost
// matrices (mot_ucmc_task, csrc/ucmc_kernels.hip) and the three assignments. Track states (20 doubles each) never leave HBM; what the
// host keeps per track is its bookkeeping (state, counters, the detection it holds).
//
// Stages: 0 map detections, predict, cost + assignment of the high-confidence detections against confirmed + coasted tracks |
//         1 cost + assignment of the low-confidence detections against the tracks left over, and of the high-confidence detections
//           left over against the tentative tracks (both depend on stage 0's result only) |
//         2 Kalman updates of every matched pair, births; the output rows are the matched detections' own boxes (:303-342).
// The reference updates a matched track's filter before it builds the next association's costs; the tracks of the later associations
// are exactly the ones no earlier association matched (or the tentative ones, which the first two never see), so applying all updates
// at the end changes nothing.
#include <cmath>
#include <string>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

enum St { Tentative = 0, Confirmed = 1, Coasted = 2, Deleted = 3 };

struct Trk {
  int id = 0, slot = -1, state = Tentative;
  int age = 0, death = 0, birth = 0, det_idx = -1;
};

class UcmcGpu final : public Staged {
 public:
  UcmcGpu(std::shared_ptr<Device> dev, const UcmcParams& p) : core_(std::move(dev), MOT_KF_XYAH), p_(p) {
    // UCMCSingleTrack ctor :185-200: Q = G Q0 G^T, G = [dt^2/2 0; dt 0; 0 dt^2/2; 0 dt], Q0 = diag(wx, wy)
    const double dt = p_.dt;
    const double G[4][2] = {{0.5 * dt * dt, 0.0}, {dt, 0.0}, {0.0, 0.5 * dt * dt}, {0.0, dt}};
    const double Q0[2][2] = {{p_.wx, 0.0}, {0.0, p_.wy}};
    double GQ[4][2];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 2; ++j) GQ[i][j] = G[i][0] * Q0[0][j] + G[i][1] * Q0[1][j];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) Q_[i * 4 + j] = GQ[i][0] * G[j][0] + GQ[i][1] * G[j][1];
    if (p_.has_camera) camera(p_.Ki, p_.Ko);
  }
  ~UcmcGpu() override {
    if (x_) mot_free(core_.dev().ctx, x_);
    if (P_) mot_free(core_.dev().ctx, P_);
  }
  Core& core() override { return core_; }
  void reset() override {  // UCMCTrack::reset :252-259
    tracks_.clear(); confirmed_.clear(); coasted_.clear(); tentative_.clear();
    free_.clear(); next_slot_ = 0; next_id_ = 0;
  }
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : tracks_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }
  // parity hook: [id, state, death, birth, det_idx, age, x(4), P(16)] per track in list order (synchronises the device)
  bool f64_states(std::vector<double>* rows) override {
    Device& dv = core_.dev();
    std::lock_guard<std::mutex> lk(dv.frame_mu);
    rows->clear();
    std::vector<double> hx(static_cast<size_t>(next_slot_) * 4), hP(static_cast<size_t>(next_slot_) * 16);
    if (next_slot_ > 0) {
      dv.check(mot_memcpy_d2h(dv.ctx, hx.data(), x_, hx.size() * sizeof(double)), "ucmc state download");
      dv.check(mot_memcpy_d2h(dv.ctx, hP.data(), P_, hP.size() * sizeof(double)), "ucmc state download");
      dv.check(mot_ctx_sync(dv.ctx), "sync");
    }
    for (const Trk& t : tracks_) {
      rows->push_back(t.id); rows->push_back(t.state); rows->push_back(t.death); rows->push_back(t.birth); rows->push_back(t.det_idx); rows->push_back(t.age);
      for (int k = 0; k < 4; ++k) rows->push_back(hx[static_cast<size_t>(t.slot) * 4 + k]);
      for (int k = 0; k < 16; ++k) rows->push_back(hP[static_cast<size_t>(t.slot) * 16 + k]);
    }
    return true;
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    stage_ = 0;
    Device& dv = core_.dev();
    // the detections at or above det_thresh, in input order (:273-296)
    keep_.clear();
    for (int i = 0; i < in.n; ++i)
      if (!(in.dets[static_cast<size_t>(4) * in.ld + i] < p_.det_thresh)) keep_.push_back(i);
    nd_ = static_cast<int>(keep_.size());
    box_.resize(static_cast<size_t>(nd_) * 4); conf_.resize(nd_); cls_.resize(nd_);
    high_.clear(); low_.clear();
    y_ = nullptr; R_ = nullptr;
    if (nd_ > 0) {
      Span<float> raw = dv.up->alloc<float>(static_cast<size_t>(6) * nd_);
      for (int j = 0; j < nd_; ++j) {
        for (int k = 0; k < 6; ++k) raw.h[static_cast<size_t>(k) * nd_ + j] = in.dets[static_cast<size_t>(k) * in.ld + keep_[j]];
        for (int k = 0; k < 4; ++k) box_[static_cast<size_t>(j) * 4 + k] = raw.h[static_cast<size_t>(k) * nd_ + j];
        conf_[j] = raw.h[static_cast<size_t>(4) * nd_ + j];
        cls_[j] = static_cast<int>(raw.h[static_cast<size_t>(5) * nd_ + j]);
        (conf_[j] >= p_.high_score ? high_ : low_).push_back(j);  // dataAssociation :347-355
      }
      y_ = dv.tmp->alloc<double>(static_cast<size_t>(nd_) * 2).d;
      R_ = dv.tmp->alloc<double>(static_cast<size_t>(nd_) * 4).d;
      mot_ucmc_task t = base_task();
      t.n = nd_; t.dets = raw.d; t.ld = nd_; t.didx = nullptr;
      dv.q().ucmc[MOT_UCMC_MAP].push_back(t);
    }
    ensure_slab(static_cast<int>(high_.size()));  // births of this frame: at most the high-confidence detections
    // predict every track (:358-361)
    const int nt = static_cast<int>(tracks_.size());
    if (nt > 0) {
      std::vector<int> slots(nt);
      for (int i = 0; i < nt; ++i) { slots[i] = tracks_[i].slot; ++tracks_[i].age; tracks_[i].det_idx = -1; }
      mot_ucmc_task t = base_task();
      t.n = nt; t.slots = core_.ints(slots).d;
      dv.q().ucmc[MOT_UCMC_PREDICT].push_back(t);
    }
    // first association (:363-413): high-confidence detections against confirmed + coasted tracks
    ta_ = confirmed_;
    ta_.insert(ta_.end(), coasted_.begin(), coasted_.end());
    lapA_ = Core::Lap();
    if (!high_.empty() && !ta_.empty()) lapA_ = associate(ta_, high_, p_.a1);
    matches_.clear();
  }

  bool advance() override {
    if (stage_ == 0) { after_a(); stage_ = 1; return true; }
    if (stage_ == 1) { after_bc(); stage_ = 2; return true; }
    stage_ = 3;
    return false;
  }

 private:
  mot_ucmc_task base_task() const {
    mot_ucmc_task t{};
    t.x = x_; t.P = P_; t.y = y_; t.R = R_;
    t.dt = p_.dt; t.vmax = p_.vmax; t.mapped = mapped_ ? 1 : 0;
    for (int k = 0; k < 16; ++k) t.Q[k] = Q_[k];
    for (int k = 0; k < 9; ++k) t.invA[k] = invA_[k];
    return t;
  }
  // CameraMapper::CameraMapper :57-83 (Ki 3 x 4, Ko 4 x 4, row-major), Eigen's 3 x 3 inverse (cofactors; det from the first column)
  void camera(const double* Ki, const double* Ko) {
    double KiKo[3][4];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += Ki[i * 4 + k] * Ko[k * 4 + j];
        KiKo[i][j] = s;
      }
    double m[3][3];
    for (int r = 0; r < 3; ++r) { m[r][0] = KiKo[r][0]; m[r][1] = KiKo[r][1]; m[r][2] = KiKo[r][3]; }
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double invdet = 1.0 / ((c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0]);
    invA_[0] = c00 * invdet; invA_[1] = c10 * invdet; invA_[2] = c20 * invdet;
    for (int r = 1; r < 3; ++r)
      for (int c = 0; c < 3; ++c) invA_[r * 3 + c] = cof(c, r) * invdet;
    mapped_ = true;
  }
  void ensure_slab(int births) {
    const int need = next_slot_ + births + 8;
    if (need <= cap_) return;
    Device& dv = core_.dev();
    int ncap = cap_ > 0 ? cap_ : 64;
    while (ncap < need) ncap *= 2;
    void *nx = nullptr, *nP = nullptr;
    dv.check(mot_malloc(dv.ctx, sizeof(double) * 4 * ncap, &nx), "ucmc state slab alloc");
    dv.check(mot_malloc(dv.ctx, sizeof(double) * 16 * ncap, &nP), "ucmc covariance slab alloc");
    if (x_) {
      dv.check(mot_memcpy_d2d(dv.ctx, nx, x_, sizeof(double) * 4 * next_slot_), "ucmc slab copy");
      dv.check(mot_memcpy_d2d(dv.ctx, nP, P_, sizeof(double) * 16 * next_slot_), "ucmc slab copy");
      dv.check(mot_ctx_sync(dv.ctx), "slab sync");
      mot_free(dv.ctx, x_); mot_free(dv.ctx, P_);
    }
    x_ = static_cast<double*>(nx); P_ = static_cast<double*>(nP);
    cap_ = ncap;
  }
  int take_slot() {
    if (!free_.empty()) { const int s = free_.back(); free_.pop_back(); return s; }
    return next_slot_++;
  }
  // cost matrix (cast to float like the reference's cost_matrix.cast<float>()) + assignment of tracks T (indices into tracks_)
  // against the kept detections D
  Core::Lap associate(const std::vector<int>& T, const std::vector<int>& D, double thresh) {
    Device& dv = core_.dev();
    const int n = static_cast<int>(T.size()), m = static_cast<int>(D.size());
    std::vector<int> slots(n);
    for (int i = 0; i < n; ++i) slots[i] = tracks_[T[i]].slot;
    const int ld = round_up(m, 4);
    float* cost = dv.tmp->alloc<float>(static_cast<size_t>(n) * ld).d;
    mot_ucmc_task t = base_task();
    t.n = n; t.m = m; t.slots = core_.ints(slots).d; t.didx = core_.ints(D).d; t.cost = cost; t.ldc = ld;
    dv.q().ucmc[MOT_UCMC_COST].push_back(t);
    return core_.lap(cost, ld, n, m, static_cast<float>(thresh));
  }
  void match(int trk, int det) {  // a matched pair: the filter update is queued for stage 2
    Trk& t = tracks_[trk];
    t.death = 0; t.det_idx = keep_[det];
    matches_.push_back({t.slot, det});
    held_.insert(det);
  }

  void after_a() {
    held_.clear();
    trk_remain_.clear();
    if (lapA_.queued) {
      record(lapA_);
      const int n = static_cast<int>(ta_.size());
      for (int i = 0; i < n; ++i) {
        const int j = lapA_.x.h[i];
        if (j >= 0) { match(ta_[i], high_[j]); tracks_[ta_[i]].state = Confirmed; }
        else trk_remain_.push_back(ta_[i]);
      }
    } else trk_remain_ = ta_;
    // second association (:415-453): the low-confidence detections against the tracks left over
    lapB_ = Core::Lap();
    if (!low_.empty() && !trk_remain_.empty()) lapB_ = associate(trk_remain_, low_, p_.a2);
    // associateTentative (:460-520): the high-confidence detections no track holds (the second association only hands out
    // low-confidence ones), in detection order, against the tentative tracks
    det_remain_.clear();
    for (int j : high_) if (!held_.count(j)) det_remain_.push_back(j);
    lapC_ = Core::Lap();
    if (!det_remain_.empty() && !tentative_.empty()) lapC_ = associate(tentative_, det_remain_, p_.a1);
  }

  void after_bc() {
    Device& dv = core_.dev();
    if (lapB_.queued) {
      record(lapB_);
      const int n = static_cast<int>(trk_remain_.size());
      for (int i = 0; i < n; ++i) {
        const int j = lapB_.x.h[i];
        if (j >= 0) { match(trk_remain_[i], low_[j]); tracks_[trk_remain_[i]].state = Confirmed; }
        else tracks_[trk_remain_[i]].state = Coasted;
      }
    } else {
      for (int i : trk_remain_) tracks_[i].state = Coasted;
    }
    std::vector<int> births;
    if (lapC_.queued) {
      record(lapC_);
      const int n = static_cast<int>(tentative_.size()), m = static_cast<int>(det_remain_.size());
      for (int i = 0; i < n; ++i) {
        const int j = lapC_.x.h[i];
        if (j < 0) continue;
        match(tentative_[i], det_remain_[j]);
        Trk& t = tracks_[tentative_[i]];
        if (++t.birth >= 2) { t.birth = 0; t.state = Confirmed; }
      }
      for (int j = 0; j < m; ++j) if (lapC_.y.h[j] < 0) births.push_back(det_remain_[j]);
    } else births = det_remain_;
    // filter updates of every matched pair (UCMCKalmanFilter::update)
    if (!matches_.empty()) {
      std::vector<int> slots(matches_.size()), dets(matches_.size());
      for (size_t i = 0; i < matches_.size(); ++i) { slots[i] = matches_[i].first; dets[i] = matches_[i].second; }
      mot_ucmc_task t = base_task();
      t.n = static_cast<int>(matches_.size()); t.slots = core_.ints(slots).d; t.didx = core_.ints(dets).d;
      dv.q().ucmc[MOT_UCMC_UPDATE].push_back(t);
    }
    // initTentative (:522-535)
    if (!births.empty()) {
      std::vector<int> slots(births.size());
      for (size_t i = 0; i < births.size(); ++i) {
        Trk t;
        t.id = ++next_id_; t.slot = take_slot(); t.state = Tentative; t.det_idx = keep_[births[i]];
        slots[i] = t.slot;
        tracks_.push_back(t);
      }
      mot_ucmc_task t = base_task();
      t.n = static_cast<int>(births.size()); t.slots = core_.ints(slots).d; t.didx = core_.ints(births).d;
      dv.q().ucmc[MOT_UCMC_INIT].push_back(t);
    }
    // deleteOldTrackers (:537-553)
    size_t w = 0;
    for (size_t i = 0; i < tracks_.size(); ++i) {
      Trk& t = tracks_[i];
      ++t.death;
      const bool del = (t.state == Coasted && t.death >= p_.max_age) || (t.state == Tentative && t.death >= 2);
      if (del) { free_.push_back(t.slot); continue; }
      if (w != i) tracks_[w] = t;
      ++w;
    }
    tracks_.resize(w);
    // updateStatus (:555-572)
    confirmed_.clear(); coasted_.clear(); tentative_.clear();
    for (size_t i = 0; i < tracks_.size(); ++i) {
      if (tracks_[i].state == Confirmed) confirmed_.push_back(static_cast<int>(i));
      else if (tracks_[i].state == Coasted) coasted_.push_back(static_cast<int>(i));
      else if (tracks_[i].state == Tentative) tentative_.push_back(static_cast<int>(i));
    }
    // output (:303-342): confirmed tracks that hold a detection of this frame, with that detection's box
    for (const Trk& t : tracks_) {
      if (t.state != Confirmed || t.det_idx < 0) continue;
      for (int j = 0; j < nd_; ++j)
        if (keep_[j] == t.det_idx) {
          rows_.push_back(box_[static_cast<size_t>(j) * 4 + 0]); rows_.push_back(box_[static_cast<size_t>(j) * 4 + 1]);
          rows_.push_back(box_[static_cast<size_t>(j) * 4 + 2]); rows_.push_back(box_[static_cast<size_t>(j) * 4 + 3]);
          rows_.push_back(static_cast<float>(t.id)); rows_.push_back(conf_[j]);
          rows_.push_back(static_cast<float>(cls_[j])); rows_.push_back(static_cast<float>(keep_[j]));
          break;
        }
    }
  }

  Core core_;
  UcmcParams p_;
  double Q_[16] = {};
  double invA_[9] = {};
  bool mapped_ = false;
  double* x_ = nullptr;
  double* P_ = nullptr;
  int cap_ = 0, next_slot_ = 0, next_id_ = 0;
  std::vector<int> free_;
  std::vector<Trk> tracks_;
  std::vector<int> confirmed_, coasted_, tentative_;
  // per frame
  int stage_ = 0, nd_ = 0;
  std::vector<int> keep_, cls_, high_, low_, ta_, trk_remain_, det_remain_;
  std::vector<float> box_, conf_;
  double* y_ = nullptr;
  double* R_ = nullptr;
  Core::Lap lapA_, lapB_, lapC_;
  std::vector<std::pair<int, int>> matches_;  // (slot, kept detection)
  IdSet held_;
};

}  // namespace

Staged* make_ucmc(std::shared_ptr<Device> dev, const UcmcParams& p) { return new UcmcGpu(std::move(dev), p); }

}  // namespace motcpp::rt
