// BoostTrack on the MI355X hot path: host lifecycle of src/trackers/boosttrack.cpp:465-699 (BoostTrackTracker::update); the ECC camera-motion
// step needs the image and is outside the path. with_reid: the embeddings come with update() (no ReID model here) — raw detection
// embedding x stored track embedding on the fp32 matrix cores (embed_kernel<dot>), the stored embeddings (normalised, EMA with a per-detection
// weight) in a feature slab on the device; without embeddings the frame is motion-only, as in the reference (:539-551).
// The constant-noise Kalman filter, the IoU row maxima of the detection-confidence boost, the association cost (1 - IoU minus the
// weighted Mahalanobis similarity) and the assignment run on the device (mot_boost_task, csrc/boost_kernels.hip); track states are
// 72-float records in the tracker's slab (Core) and never leave HBM.
//
// Stages: 0 predict every track (+ its predicted box), row maxima of IoU(detection, predicted box) and the visual-tracking flag |
//         (host: the boosted confidences — max / pow as the reference calls them —, the detections at or above det_thresh)
//         1 cost matrix detections x tracks, assignment |
//         2 filter updates, births, boxes of the tracks to report.
#include <cmath>
#include <string>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

struct Trk {
  int id = 0, slot = -1, cls = 0, det_ind = -1, tsu = 0, age = 0, hit_streak = 0;
  float conf = 0.f;
  bool has_emb = false;
};

class BoostTrackGpu final : public Staged {
 public:
  BoostTrackGpu(std::shared_ptr<Device> dev, const BoostParams& p) : core_(std::move(dev), MOT_KF_XYAH), p_(p) {}
  ~BoostTrackGpu() override { if (feat_) mot_free(core_.dev().ctx, feat_); }
  Core& core() override { return core_; }
  const float* feature_slab(int* dim, std::vector<char>* has) const override {
    *dim = D_;
    for (const Trk& t : tracks_) has->push_back(t.has_emb ? 1 : 0);
    return feat_;
  }
  void reset() override { tracks_.clear(); frame_count_ = 0; next_id_ = 0; core_.clear_slots(); }  // :272-277
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : tracks_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    stage_ = 0;
    ++frame_count_;
    Device& dv = core_.dev();
    n_ = in.n;
    conf_.resize(n_); cls_.resize(n_);
    raw_ = nullptr;
    core_.reserve(n_ + 8, 8);
    // embeddings of this frame (:539-551): used when with_reid and the caller passed them
    use_emb_ = p_.with_reid && in.embs != nullptr && in.emb_dim > 0 && n_ > 0;
    emb_raw_ = nullptr;
    if (use_emb_) {
      if (D_ == 0) D_ = in.emb_dim;
      if (D_ != in.emb_dim) throw Error("BoostTrack: embedding dimension changed between frames");
      ensure_feat();
      Span<float> e = dv.up->alloc<float>(static_cast<size_t>(n_) * D_);
      for (int i = 0; i < n_; ++i)
        for (int k = 0; k < D_; ++k)
          e.h[static_cast<size_t>(i) * D_ + k] = in.embs_rowmajor ? in.embs[static_cast<size_t>(i) * in.emb_ld + k] : in.embs[static_cast<size_t>(k) * in.emb_ld + i];
      emb_raw_ = e.d;
    }
    if (n_ > 0) {
      Span<float> raw = dv.up->alloc<float>(static_cast<size_t>(6) * n_);
      for (int k = 0; k < 6; ++k)
        for (int i = 0; i < n_; ++i) raw.h[static_cast<size_t>(k) * n_ + i] = in.dets[static_cast<size_t>(k) * in.ld + i];
      for (int i = 0; i < n_; ++i) { conf_[i] = raw.h[static_cast<size_t>(4) * n_ + i]; cls_[i] = static_cast<int>(raw.h[static_cast<size_t>(5) * n_ + i]); }
      raw_ = raw.d;
    }
    // BoostTrack::predict :156-163 for every track
    const int nt = static_cast<int>(tracks_.size());
    pbox_ = nullptr;
    maxs_ = Span<float>(); vt_ = Span<int32_t>();
    if (nt > 0) {
      std::vector<int> slots(nt), tsu(nt);
      for (int j = 0; j < nt; ++j) {
        Trk& t = tracks_[j];
        slots[j] = t.slot;
        ++t.age;
        if (t.tsu > 0) t.hit_streak = 0;
        ++t.tsu;
        tsu[j] = t.tsu;
      }
      slots_d_ = core_.ints(slots).d;
      pbox_ = dv.tmp->alloc<float>(static_cast<size_t>(nt) * 4).d;
      mot_boost_task t{};
      t.n = nt; t.slab = core_.d_mean(); t.slots = slots_d_; t.boxes = pbox_;
      dv.q().boost[MOT_BOOST_PREDICT].push_back(t);
      if (p_.use_dlo && n_ > 0) {  // dlo_confidence_boost :361-426: what it needs from the device
        maxs_ = dv.down->alloc<float>(n_);
        vt_ = dv.down->alloc<int32_t>(n_);
        mot_boost_task d{};
        d.n = n_; d.m = nt; d.dets = raw_; d.ldd = n_; d.boxes = pbox_; d.tsu = core_.ints(tsu).d; d.max_s = maxs_.d; d.vt = vt_.d;
        dv.q().boost[MOT_BOOST_DLO].push_back(d);
      }
    }
  }

  bool advance() override {
    if (stage_ == 0) { after_predict(); stage_ = 1; return true; }
    if (stage_ == 1) { after_assignment(); stage_ = 2; return true; }
    if (stage_ == 2) { emit(); stage_ = 3; }
    return false;
  }

 private:
  void after_predict() {
    Device& dv = core_.dev();
    const int nt = static_cast<int>(tracks_.size());
    if (maxs_.h) {
      for (int i = 0; i < n_; ++i) {
        const float max_s = maxs_.h[i];
        if (!p_.use_sb && !p_.use_vt) conf_[i] = std::max(conf_[i], max_s * p_.dlo_coef);  // :393-400
        else {
          if (p_.use_sb) {  // :402-410
            const float alpha = 0.65f;
            const float bc = alpha * conf_[i] + (1.0f - alpha) * std::pow(max_s, 1.5f);
            conf_[i] = std::max(conf_[i], bc);
          }
          if (p_.use_vt && vt_.h[i]) conf_[i] = std::max(conf_[i], p_.det_thresh + 1e-5f);  // :412-424
        }
      }
    }
    keep_.clear();
    for (int i = 0; i < n_; ++i) if (conf_[i] >= p_.det_thresh) keep_.push_back(i);  // :529-537
    lap_ = Core::Lap();
    const int nd = static_cast<int>(keep_.size());
    if (nd > 0 && nt > 0) {
      const int ld = round_up(nt, 4);
      float* cost = dv.tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
      mot_boost_task t{};
      t.n = nd; t.m = nt; t.slab = core_.d_mean(); t.slots = slots_d_; t.didx = core_.ints(keep_).d; t.dets = raw_; t.ldd = n_;
      t.cost = cost; t.ldc = ld; t.lambda_mhd = p_.lambda_mhd;
      if (use_emb_) {  // :581-594: raw detection embeddings x the tracks' stored ones (a track without one: the slab's all-zero row)
        std::vector<int> frows(nt);
        for (int j = 0; j < nt; ++j) frows[j] = tracks_[j].has_emb ? tracks_[j].slot : zero_row();
        float* emb = dv.tmp->alloc<float>(static_cast<size_t>(nd) * ld).d;
        mot_cos_task c{};
        c.n = nd; c.m = nt; c.d = D_; c.a = emb_raw_; c.lda = D_; c.aidx = t.didx; c.b = feat_; c.ldb = D_; c.bidx = core_.ints(frows).d; c.out = emb; c.ldo = ld;
        dv.q().dot.push_back(c);
        t.emb = emb; t.lde = ld; t.lambda_emb = (1.0f + p_.lambda_iou + p_.lambda_shape + p_.lambda_mhd) * 1.5f;
      }
      dv.q().boost[MOT_BOOST_COST].push_back(t);
      lap_ = core_.lap(cost, ld, nd, nt, p_.iou_threshold);  // rows = detections, columns = tracks (:618-623)
    }
  }

  void after_assignment() {
    Device& dv = core_.dev();
    const int nd = static_cast<int>(keep_.size());
    std::vector<int> us, ud, births;
    std::vector<int> ema_slots, ema_dets, set_slots, set_dets;
    std::vector<float> ema_alpha;
    if (lap_.queued) {
      record(lap_);
      for (int i = 0; i < nd; ++i) {
        const int j = lap_.x.h[i];
        if (j < 0) { births.push_back(keep_[i]); continue; }
        Trk& t = tracks_[j];  // BoostTrack::update :165-181
        t.tsu = 0; ++t.hit_streak;
        t.conf = conf_[keep_[i]]; t.cls = cls_[keep_[i]]; t.det_ind = keep_[i];
        us.push_back(t.slot); ud.push_back(keep_[i]);
        if (use_emb_) {  // update_emb :183-199 with dets_alpha (:637-650)
          const float trust = (conf_[keep_[i]] - p_.det_thresh) / (1.0f - p_.det_thresh);
          const float af = 0.95f;
          if (t.has_emb) { ema_slots.push_back(t.slot); ema_dets.push_back(keep_[i]); ema_alpha.push_back(af + (1.0f - af) * (1.0f - trust)); }
          else { set_slots.push_back(t.slot); set_dets.push_back(keep_[i]); t.has_emb = true; }
        }
      }
    } else births = keep_;
    if (!us.empty()) {
      mot_boost_task t{};
      t.n = static_cast<int>(us.size()); t.slab = core_.d_mean(); t.slots = core_.ints(us).d; t.didx = core_.ints(ud).d; t.dets = raw_; t.ldd = n_;
      dv.q().boost[MOT_BOOST_UPDATE].push_back(t);
    }
    if (!births.empty()) {  // :652-661
      std::vector<int> slots(births.size());
      for (size_t k = 0; k < births.size(); ++k) {
        Trk t;
        t.id = ++next_id_; t.slot = core_.new_slot(); t.conf = conf_[births[k]]; t.cls = cls_[births[k]]; t.det_ind = births[k];
        if (use_emb_) { set_slots.push_back(t.slot); set_dets.push_back(births[k]); t.has_emb = true; }
        slots[k] = t.slot;
        tracks_.push_back(t);
      }
      mot_boost_task t{};
      t.n = static_cast<int>(births.size()); t.slab = core_.d_mean(); t.slots = core_.ints(slots).d; t.didx = core_.ints(births).d; t.dets = raw_; t.ldd = n_;
      dv.q().boost[MOT_BOOST_INIT].push_back(t);
    }
    if (!set_slots.empty()) {  // BoostTrack ctor :147-153 / update_emb's first embedding: emb / |emb| when the norm is positive
      mot_feat_task f{};
      f.n = static_cast<int>(set_slots.size()); f.d = D_; f.feat = feat_; f.ldf = D_; f.slot = core_.ints(set_slots).d;
      f.src = emb_raw_; f.lds = D_; f.sidx = core_.ints(set_dets).d; f.mode = 0; f.alpha = 0.f;
      dv.q().feat_set.push_back(f);
    }
    if (!ema_slots.empty()) {  // update_emb :183-199: normalise the detection's embedding (scratch rows), blend, renormalise
      const int ne = static_cast<int>(ema_slots.size());
      float* tmp = dv.tmp->alloc<float>(static_cast<size_t>(ne) * D_).d;
      mot_feat_task f{};
      f.n = ne; f.d = D_; f.feat = tmp; f.ldf = D_; f.slot = nullptr; f.src = emb_raw_; f.lds = D_; f.sidx = core_.ints(ema_dets).d; f.mode = 0; f.alpha = 0.f;
      dv.q().feat_set.push_back(f);
      mot_feat_task e{};
      e.n = ne; e.d = D_; e.feat = feat_; e.ldf = D_; e.slot = core_.ints(ema_slots).d; e.src = tmp; e.lds = D_; e.sidx = nullptr; e.mode = 1;
      e.alpha = 0.f; e.alpha_i = core_.floats(ema_alpha).d;
      dv.q().feat_ema.push_back(e);
    }
    // the tracks to report (:663-680) need their boxes after the update
    out_.clear();
    for (size_t j = 0; j < tracks_.size(); ++j) {
      const Trk& t = tracks_[j];
      if (t.tsu < 1 && (t.hit_streak >= p_.min_hits || frame_count_ <= p_.min_hits)) out_.push_back(static_cast<int>(j));
    }
    obox_ = Span<float>();
    if (!out_.empty()) {
      std::vector<int> slots(out_.size());
      for (size_t k = 0; k < out_.size(); ++k) slots[k] = tracks_[out_[k]].slot;
      obox_ = dv.down->alloc<float>(out_.size() * 4);
      mot_boost_task t{};
      t.n = static_cast<int>(out_.size()); t.slab = core_.d_mean(); t.slots = core_.ints(slots).d; t.boxes = obox_.d;
      dv.q().boost[MOT_BOOST_BOXES].push_back(t);
    }
  }

  void emit() {
    for (size_t k = 0; k < out_.size(); ++k) {
      const Trk& t = tracks_[out_[k]];
      const float* b = obox_.h + k * 4;
      // filter_outputs :434-463
      const float w = b[2] - b[0], h = b[3] - b[1];
      const float area = w * h, ar = w / (h + 1e-6f);
      if (!(ar <= p_.aspect_ratio_thresh && area > static_cast<float>(p_.min_box_area))) continue;
      rows_.push_back(b[0]); rows_.push_back(b[1]); rows_.push_back(b[2]); rows_.push_back(b[3]);
      rows_.push_back(static_cast<float>(t.id)); rows_.push_back(t.conf);
      rows_.push_back(static_cast<float>(t.cls)); rows_.push_back(static_cast<float>(t.det_ind));
    }
    // :682-687
    size_t w = 0;
    for (size_t j = 0; j < tracks_.size(); ++j) {
      if (tracks_[j].tsu > p_.max_age) { core_.release_slot(tracks_[j].slot); continue; }
      if (w != j) tracks_[w] = tracks_[j];
      ++w;
    }
    tracks_.resize(w);
  }

  // feature slab: one row per Kalman slot, and one all-zero row behind them for tracks that have no embedding yet
  int zero_row() const { return feat_cap_; }
  void ensure_feat() {
    const int need = core_.cap();
    if (feat_cap_ >= need && feat_) return;
    Device& dv = core_.dev();
    void* nf = nullptr;
    dv.check(mot_malloc(dv.ctx, sizeof(float) * static_cast<size_t>(need + 1) * D_, &nf), "feature slab alloc");
    dv.check(mot_memset(dv.ctx, nf, 0, sizeof(float) * static_cast<size_t>(need + 1) * D_), "feature slab clear");
    if (feat_) {
      dv.check(mot_memcpy_d2d(dv.ctx, nf, feat_, sizeof(float) * static_cast<size_t>(feat_cap_) * D_), "feature slab copy");
      dv.check(mot_ctx_sync(dv.ctx), "slab sync");
      mot_free(dv.ctx, feat_);
    }
    feat_ = static_cast<float*>(nf);
    feat_cap_ = need;
  }

  Core core_;
  BoostParams p_;
  std::vector<Trk> tracks_;
  int frame_count_ = 0, next_id_ = 0;
  float* feat_ = nullptr;
  int feat_cap_ = 0, D_ = 0;
  bool use_emb_ = false;
  const float* emb_raw_ = nullptr;
  // per frame
  int stage_ = 0, n_ = 0;
  const float* raw_ = nullptr;
  const int32_t* slots_d_ = nullptr;
  float* pbox_ = nullptr;
  Span<float> maxs_, obox_;
  Span<int32_t> vt_;
  std::vector<float> conf_;
  std::vector<int> cls_, keep_, out_;
  Core::Lap lap_;
};

}  // namespace

Staged* make_boosttrack(std::shared_ptr<Device> dev, const BoostParams& p) { return new BoostTrackGpu(std::move(dev), p); }

}  // namespace motcpp::rt
