// SORT on the MI355X hot path: host lifecycle of src/trackers/sort.cpp:102-255 over device-resident XYSR
// Kalman states. Stage 0: det prepare, in-place predict of every track (+ boxes, downloaded for the NaN
// rule of sort.cpp:132-150), IoU distance, LAP(1 - iou_threshold). Stage 1: Kalman updates / new tracks,
// boxes of the rows to emit. If a predicted box is NaN the association is redone on the survivors.
#include <cmath>

#include "staged.hpp"

namespace motcpp::rt {
namespace {

struct Trk {
  int id = 0, slot = -1, cls = 0, det_ind = -1, hits = 1, tsu = 0, age = 1;
  float conf = 0.f;
};

class SortGpu final : public Staged {
 public:
  SortGpu(std::shared_ptr<Device> dev, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold)
      : core_(std::move(dev), MOT_KF_XYSR), det_thresh_(det_thresh), max_age_(max_age), min_hits_(min_hits), iou_thr_(iou_threshold) {
    (void)max_obs;
  }
  Core& core() override { return core_; }
  void reset() override { trk_.clear(); frame_count_ = 0; core_.clear_slots(); }  // sort.cpp:97-100 (ids keep counting)
  void live_tracks(std::vector<int>* ids, std::vector<int>* slots) const override {
    for (const Trk& t : trk_) { ids->push_back(t.id); slots->push_back(t.slot); }
  }

  void begin(const FrameIn& in) override {
    rows_.clear(); laps_.clear();
    ++frame_count_;
    stage_ = 0;
    valid_.clear(); conf_.clear(); cls_.clear();
    for (int i = 0; i < in.n; ++i)
      if (in.dets[static_cast<size_t>(4) * in.ld + i] >= det_thresh_) valid_.push_back(i);
    conf_.assign(in.n, 0.f); cls_.assign(in.n, 0);
    for (int i = 0; i < in.n; ++i) {
      conf_[i] = in.dets[static_cast<size_t>(4) * in.ld + i];
      cls_[i] = static_cast<int>(in.dets[static_cast<size_t>(5) * in.ld + i]);
    }
    core_.reserve(static_cast<int>(valid_.size()) + 8, 8);
    dets_ = core_.upload_dets(in.dets, in.n, in.ld, MOT_DET_XYSR, in.d_dets, in.d_ld);
    const int nt = static_cast<int>(trk_.size());
    lap_ = Core::Lap();
    pbox_ = Span<float>();
    if (nt > 0) {
      std::vector<int> slots(nt);
      for (int i = 0; i < nt; ++i) { slots[i] = trk_[i].slot; ++trk_[i].age; ++trk_[i].tsu; }  // SortTrack::predict :43-51
      pbox_d_ = core_.predict(slots, nullptr, nullptr, &pbox_);
      if (!valid_.empty()) queue_assoc(pbox_d_, nt, nullptr, nt);
    }
  }

  bool advance() override {
    if (stage_ == 0) {
      // NaN rule (:132-150): drop tracks whose predicted box has a NaN, then associate the survivors
      const int nt = static_cast<int>(trk_.size());
      std::vector<int> keep;
      for (int i = 0; i < nt; ++i) {
        const float s = pbox_.h[i] + pbox_.h[nt + i] + pbox_.h[2 * nt + i] + pbox_.h[3 * nt + i];
        if (!std::isnan(s)) keep.push_back(i);
      }
      if (static_cast<int>(keep.size()) != nt) {
        std::vector<Trk> kept;
        for (int i = 0, k = 0; i < nt; ++i) {
          if (k < static_cast<int>(keep.size()) && keep[k] == i) { kept.push_back(trk_[i]); ++k; }
          else core_.release_slot(trk_[i].slot);
        }
        trk_ = std::move(kept);
        stage_ = 1;
        lap_ = Core::Lap();
        if (!trk_.empty() && !valid_.empty()) {
          keep_d_ = core_.ints(keep);
          queue_assoc(pbox_d_, nt, keep_d_.d, static_cast<int>(keep.size()));
          return true;
        }
      } else {
        stage_ = 1;
      }
    }
    if (stage_ == 1) { apply(); stage_ = 2; return true; }
    if (stage_ == 2) { emit(); stage_ = 3; }
    return false;
  }

 private:
  void queue_assoc(const float* boxes, int ld, const int32_t* aidx, int n) {
    valid_d_ = core_.ints(valid_);
    Core::IouArgs a;
    a.a = boxes; a.lda = ld; a.aidx = aidx; a.n = n;
    a.b = dets_.d_box; a.ldb = dets_.n; a.bidx = valid_d_.d; a.m = static_cast<int>(valid_.size());
    a.mode = MOT_COST_IOU_DIST;
    lap_ = core_.lap_geom(a, 1.0f - iou_thr_);
  }
  void apply() {
    const int nt = static_cast<int>(trk_.size()), nd = static_cast<int>(valid_.size());
    std::vector<int> x(nt, -1), y(nd, -1);
    if (lap_.queued) {
      record(lap_);
      x.assign(lap_.x.h, lap_.x.h + nt);
      y.assign(lap_.y.h, lap_.y.h + nd);
    }
    std::vector<int> us, um, is, im;
    for (int i = 0; i < nt; ++i)
      if (x[i] >= 0) {  // SortTrack::update :53-70
        Trk& t = trk_[i];
        const int det = valid_[x[i]];
        t.conf = conf_[det]; t.cls = cls_[det]; t.det_ind = det;
        ++t.hits; t.tsu = 0;
        us.push_back(t.slot); um.push_back(det);
      }
    for (int j = 0; j < nd; ++j)
      if (y[j] < 0) {  // new tracker :196-204
        Trk t;
        t.id = ++next_id_;
        t.slot = core_.new_slot();
        const int det = valid_[j];
        t.conf = conf_[det]; t.cls = cls_[det]; t.det_ind = det;
        is.push_back(t.slot); im.push_back(det);
        trk_.push_back(t);
      }
    std::vector<Trk> keep;
    for (const Trk& t : trk_) {
      if (t.tsu <= max_age_) keep.push_back(t);
      else core_.release_slot(t.slot);
    }
    trk_ = std::move(keep);
    core_.initiate(is, im, dets_);
    core_.update(us, us, um, dets_);
    out_idx_.clear();
    std::vector<int> slots;
    for (size_t i = 0; i < trk_.size(); ++i) {
      const Trk& t = trk_[i];
      if (t.tsu == 0 && (t.hits >= min_hits_ || frame_count_ <= min_hits_)) { out_idx_.push_back(static_cast<int>(i)); slots.push_back(t.slot); }
    }
    obox_ = Span<float>();
    core_.boxes(slots, &obox_);
  }
  void emit() {
    const int n = static_cast<int>(out_idx_.size());
    for (int k = 0; k < n; ++k) {
      const Trk& t = trk_[out_idx_[k]];
      push_row(obox_.h, n, k, t.id, t.conf, t.cls, t.det_ind);
    }
  }

  Core core_;
  float det_thresh_;
  int max_age_, min_hits_;
  float iou_thr_;
  int frame_count_ = 0, next_id_ = 0, stage_ = 0;
  std::vector<Trk> trk_;
  Core::Dets dets_;
  std::vector<int> valid_, cls_, out_idx_;
  std::vector<float> conf_;
  Span<float> pbox_, obox_;
  float* pbox_d_ = nullptr;
  Span<int32_t> valid_d_, keep_d_;
  Core::Lap lap_;
};

}  // namespace

Staged* make_sort(std::shared_ptr<Device> dev, float det_thresh, int max_age, int max_obs, int min_hits, float iou_threshold) {
  return new SortGpu(std::move(dev), det_thresh, max_age, max_obs, min_hits, iou_threshold);
}

}  // namespace motcpp::rt
