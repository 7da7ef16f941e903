// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// CPU restatement of the four in-scope trackers' update() orchestration, including the
// reference's lifecycle quirks:
//   Sort      src/trackers/sort.cpp:16-255
//   ByteTrack src/trackers/bytetrack.cpp:15-706
//   OCSort    src/trackers/ocsort.cpp:24-738
//   BotSort   src/trackers/botsort.cpp:16-764   (cmc_method != "ecc": no CMC; embeddings passed in)
// Id counters are per tracker instance (the reference's process-global statics, Q6, make
// ids depend on every other instance in the process; parity is defined per stream).
// Detections come in as a row-major N x 6 array [x1,y1,x2,y2,conf,cls]; output rows are
// [x1,y1,x2,y2,id,conf,cls,det_ind] like the reference's M x 8 matrix.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <stdexcept>
#include <string>
#include <set>
#include <unordered_map>
#include <unordered_set>

#include "orc_kf.hpp"
#include "orc_math.hpp"

namespace orc {

using OutRow = std::array<float, 8>;
using OutTable = std::vector<OutRow>;

struct Det7 {  // one detection row with its index in the caller's matrix appended
  float x1, y1, x2, y2, conf, cls;
  int ind;
  Box box() const { return {x1, y1, x2, y2}; }
};
inline std::vector<Det7> wrap_dets(const float* dets, int n) {
  std::vector<Det7> v(n);
  for (int i = 0; i < n; ++i) {
    const float* r = dets + static_cast<size_t>(i) * 6;
    v[i] = {r[0], r[1], r[2], r[3], r[4], r[5], i};
  }
  return v;
}
inline Mat boxes_to_mat(const std::vector<Box>& b) {
  Mat m(static_cast<int>(b.size()), 4);
  for (int i = 0; i < m.r; ++i)
    for (int k = 0; k < 4; ++k) m(i, k) = b[i][k];
  return m;
}

// tracker.cpp:17-46 — shared parameter block (max_obs forced to max_age+5 when max_age>=max_obs)
struct BaseParams {
  float det_thresh = 0.3f;
  int max_age = 30, max_obs = 50, min_hits = 3;
  float iou_threshold = 0.3f;
  void normalise() {
    if (max_age >= max_obs) max_obs = max_age + 5;
  }
};

// =======================================================================================
// SORT — sort.cpp
// =======================================================================================
class Sort {
 public:
  explicit Sort(float det_thresh = 0.3f, int max_age = 1, int max_obs = 50, int min_hits = 3,
                float iou_threshold = 0.3f) {
    p_.det_thresh = det_thresh; p_.max_age = max_age; p_.max_obs = max_obs;
    p_.min_hits = min_hits; p_.iou_threshold = iou_threshold;
    p_.normalise();
  }
  void reset() { trk_.clear(); frame_count_ = 0; }  // sort.cpp:97-100 (id counter is NOT reset)

  OutTable update(const float* dets, int n) {  // sort.cpp:102-255
    ++frame_count_;
    std::vector<Det7> all = wrap_dets(dets, n), fd;
    for (const Det7& d : all)
      if (d.conf >= p_.det_thresh) fd.push_back(d);

    std::vector<Box> trks(trk_.size());
    std::vector<int> to_del;
    for (size_t t = 0; t < trk_.size(); ++t) {
      predict(trk_[t]);
      Box pos = state_box(trk_[t]);
      if (std::isnan(pos[0] + pos[1] + pos[2] + pos[3])) to_del.push_back(static_cast<int>(t));
      trks[t] = pos;
    }
    for (auto it = to_del.rbegin(); it != to_del.rend(); ++it) trk_.erase(trk_.begin() + *it);
    if (!to_del.empty()) {
      trks.resize(trk_.size());
      for (size_t t = 0; t < trk_.size(); ++t) trks[t] = state_box(trk_[t]);
    }

    std::vector<std::array<int, 2>> matched;
    std::vector<int> um_dets, um_trks;
    if (trk_.empty()) {
      for (int i = 0; i < static_cast<int>(fd.size()); ++i) um_dets.push_back(i);
    } else if (fd.empty()) {
      for (int t = 0; t < static_cast<int>(trk_.size()); ++t) um_trks.push_back(t);
    } else {
      std::vector<Box> db;
      for (const Det7& d : fd) db.push_back(d.box());
      Mat cost = iou_distance(boxes_to_mat(trks), boxes_to_mat(db));
      LapResult r = linear_assignment(cost, 1.0f - p_.iou_threshold);
      matched = r.matches; um_dets = r.unmatched_b; um_trks = r.unmatched_a;
      last_lap = r;
    }
    for (const auto& m : matched) apply(trk_[m[0]], fd[m[1]]);
    for (int di : um_dets) trk_.push_back(spawn(fd[di]));

    std::vector<Track> keep;
    for (const Track& t : trk_)
      if (t.tsu <= p_.max_age) keep.push_back(t);
    trk_ = std::move(keep);

    OutTable out;
    for (const Track& t : trk_) {
      if (t.tsu == 0 && (t.hits >= p_.min_hits || frame_count_ <= p_.min_hits)) {
        Box b = state_box(t);
        out.push_back({b[0], b[1], b[2], b[3], static_cast<float>(t.id), t.conf,
                       static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
      }
    }
    return out;
  }
  LapResult last_lap;  // exposed for assignment-index parity checks
  int num_tracks() const { return static_cast<int>(trk_.size()); }
  // state dump for float parity: rows of [id, x(7), P(49)]
  std::vector<std::vector<float>> dump_states() const {
    std::vector<std::vector<float>> v;
    for (const Track& t : trk_) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id));
      for (int i = 0; i < 7; ++i) r.push_back(t.kf.x[i]);
      for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) r.push_back(t.kf.P[i][j]);
      v.push_back(r);
    }
    return v;
  }

 private:
  struct Track {  // SortTrack, sort.cpp:21-76
    int id, cls, det_ind, hits, tsu, age;
    float conf;
    KfXYSR kf;
  };
  Track spawn(const Det7& d) {
    Track t;
    t.id = ++next_id_;
    t.conf = d.conf; t.cls = static_cast<int>(d.cls); t.det_ind = d.ind;
    t.hits = 1; t.tsu = 0; t.age = 1;
    Box z = xyxy2xysr(d.box());
    for (int i = 0; i < 4; ++i) t.kf.x[i] = z[i];
    return t;
  }
  static void predict(Track& t) { t.kf.predict(); ++t.age; ++t.tsu; }
  static void apply(Track& t, const Det7& d) {
    t.conf = d.conf; t.cls = static_cast<int>(d.cls); t.det_ind = d.ind;
    Box z = xyxy2xysr(d.box());
    t.kf.update(z.data());
    ++t.hits; t.tsu = 0;
  }
  static Box state_box(const Track& t) { return xysr2xyxy({t.kf.x[0], t.kf.x[1], t.kf.x[2], t.kf.x[3]}); }

  BaseParams p_;
  int frame_count_ = 0, next_id_ = 0;
  std::vector<Track> trk_;
};

// =======================================================================================
// ByteTrack — bytetrack.cpp
// =======================================================================================
class ByteTrack {
 public:
  explicit ByteTrack(float min_conf = 0.1f, float track_thresh = 0.45f, float match_thresh = 0.8f,
                     int track_buffer = 25, int frame_rate = 30, int max_age = 30, int max_obs = 50)
      : min_conf_(min_conf), track_thresh_(track_thresh), match_thresh_(match_thresh) {
    p_.max_age = max_age; p_.max_obs = max_obs; p_.normalise();
    buffer_size_ = static_cast<int>(frame_rate / 30.0f * track_buffer);  // bytetrack.cpp:141
    max_time_lost_ = buffer_size_;
    p_.det_thresh = track_thresh_;  // :145
  }
  void reset() { frame_count_ = 0; frame_id_ = 0; active_.clear(); lost_.clear(); }

  enum State { New = 0, Tracked = 1, Lost = 2, Removed = 3 };
  struct STrack {
    bool has_state = false;
    State8 kf;
    Box xywh, tlwh, xyah;
    float conf = 0.f;
    int cls = 0, det_ind = 0, id = 0;
    State state = New;
    bool activated = false;
    int tracklet_len = 0, frame_id = 0, start_frame = 0;
    Box xyxy() const {  // bytetrack.cpp:117-127
      if (!has_state) return xywh2xyxy(xywh);
      return xywh2xyxy(xyah2xywh({kf.mean[0], kf.mean[1], kf.mean[2], kf.mean[3]}));
    }
  };

  OutTable update(const float* dets, int n) {  // bytetrack.cpp:166-621
    std::vector<Det7> all = wrap_dets(dets, n);
    ++frame_count_; ++frame_id_;
    std::vector<STrack> activated, refind, lost_new, removed_new;
    laps.clear();

    std::vector<STrack> detections, detections_second;
    for (const Det7& d : all) {
      if (d.conf > track_thresh_) detections.push_back(make_det(d));
    }
    for (const Det7& d : all) {
      if (d.conf > min_conf_ && d.conf < track_thresh_) detections_second.push_back(make_det(d));
    }

    std::vector<int> unconf_idx, tracked_idx;
    for (size_t i = 0; i < active_.size(); ++i)
      (active_[i].activated ? tracked_idx : unconf_idx).push_back(static_cast<int>(i));
    std::vector<STrack> unconfirmed, tracked;
    for (int i : unconf_idx) unconfirmed.push_back(active_[i]);
    for (int i : tracked_idx) tracked.push_back(active_[i]);

    // pool = copies of tracked ∪ lost, with a map back to the originals (:251-262)
    std::vector<STrack> pool = joint(tracked, lost_);
    std::vector<std::pair<int, bool>> origin;
    for (size_t i = 0; i < tracked.size(); ++i) origin.emplace_back(tracked_idx[i], true);
    for (size_t i = 0; i < lost_.size(); ++i) origin.emplace_back(static_cast<int>(i), false);
    // NB: joint() drops lost tracks whose id already exists in tracked; the reference builds
    // `origin` without that filter (:259-262). Ids are unique across the two lists in practice.

    for (STrack& st : pool) {  // multi_predict :97-115
      if (st.state != Tracked) st.kf.mean[7] = 0.0f;
      KfXYAH::predict(st.kf);
    }

    const int nt = static_cast<int>(pool.size()), nd = static_cast<int>(detections.size());
    std::vector<Box> tb(nt), db(nd);
    std::vector<float> dconf(nd);
    for (int i = 0; i < nt; ++i) tb[i] = pool[i].xyxy();
    for (int j = 0; j < nd; ++j) { db[j] = detections[j].xyxy(); dconf[j] = detections[j].conf; }
    Mat dists = iou_distance(boxes_to_mat(tb), boxes_to_mat(db));
    dists = fuse_score(dists, dconf);
    LapResult a1 = linear_assignment(dists, match_thresh_);
    laps.push_back(a1);

    std::vector<int> u_track = a1.unmatched_a, u_det = a1.unmatched_b;
    for (const auto& m : a1.matches) {  // :337-365
      auto [oi, is_tracked] = origin[m[0]];
      STrack& orig = is_tracked ? active_[oi] : lost_[oi];
      orig.kf = pool[m[0]].kf; orig.has_state = true;
      if (orig.state == Tracked) { apply_update(orig, detections[m[1]]); activated.push_back(orig); }
      else { reactivate(orig, detections[m[1]]); refind.push_back(orig); }
    }

    // second association (:367-442): un-predicted originals of the still-Tracked pool members
    std::vector<STrack*> r_tracked;
    std::vector<int> r_pool_idx;
    for (int idx : u_track) {
      if (pool[idx].state == Tracked) {
        auto [oi, is_tracked] = origin[idx];
        if (is_tracked) { r_tracked.push_back(&active_[oi]); r_pool_idx.push_back(idx); }
      }
    }
    if (!detections_second.empty() && !r_tracked.empty()) {
      std::vector<Box> rb, d2;
      for (STrack* t : r_tracked) rb.push_back(t->xyxy());
      for (const STrack& d : detections_second) d2.push_back(d.xyxy());
      Mat dists2 = iou_distance(boxes_to_mat(rb), boxes_to_mat(d2));
      LapResult a2 = linear_assignment(dists2, 0.5f);
      laps.push_back(a2);
      for (const auto& m : a2.matches) {
        STrack* t = r_tracked[m[0]];
        t->kf = pool[r_pool_idx[m[0]]].kf;
        if (t->state == Tracked) { apply_update(*t, detections_second[m[1]]); activated.push_back(*t); }
        else { reactivate(*t, detections_second[m[1]]); refind.push_back(*t); }
      }
      for (int i : a2.unmatched_a) {
        STrack* t = r_tracked[i];
        if (t->state != Lost) { t->state = Lost; lost_new.push_back(*t); }
      }
    }

    // unconfirmed tracks vs. leftover high detections (:455-542)
    std::vector<STrack> remaining;
    for (int idx : u_det) remaining.push_back(detections[idx]);
    std::vector<int> u_det_final;
    if (!unconfirmed.empty() && !remaining.empty()) {
      std::vector<Box> ub, rb;
      std::vector<float> rc;
      for (const STrack& t : unconfirmed) ub.push_back(t.xyxy());
      for (const STrack& d : remaining) { rb.push_back(d.xyxy()); rc.push_back(d.conf); }
      Mat dists3 = fuse_score(iou_distance(boxes_to_mat(ub), boxes_to_mat(rb)), rc);
      LapResult a3 = linear_assignment(dists3, 0.7f);
      laps.push_back(a3);
      for (int j : a3.unmatched_b) u_det_final.push_back(u_det[j]);
      for (const auto& m : a3.matches) {
        STrack& t = active_[unconf_idx[m[0]]];
        apply_update(t, remaining[m[1]]);
        activated.push_back(t);
      }
      for (int i : a3.unmatched_a) {
        STrack& t = active_[unconf_idx[i]];
        t.state = Removed;
        removed_new.push_back(t);
      }
    } else {
      u_det_final = u_det;
    }

    for (int idx : u_det_final) {  // :546-554
      STrack& t = detections[idx];
      if (t.conf >= p_.det_thresh) { activate(t); activated.push_back(t); }
    }
    for (STrack& t : lost_) {  // :557-562
      if (frame_count_ - t.frame_id > max_time_lost_) { t.state = Removed; removed_new.push_back(t); }
    }

    std::vector<STrack> na;
    for (const STrack& t : active_)
      if (t.state == Tracked) na.push_back(t);
    active_ = joint(joint(na, activated), refind);
    lost_ = sub(lost_, active_);
    lost_.insert(lost_.end(), lost_new.begin(), lost_new.end());
    lost_ = sub(lost_, removed_new);
    remove_duplicates();

    OutTable out;
    for (const STrack& t : active_) {
      if (!t.activated) continue;
      Box b = t.xyxy();
      out.push_back({b[0], b[1], b[2], b[3], static_cast<float>(t.id), t.conf,
                     static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
    }
    return out;
  }

  std::vector<LapResult> laps;
  int num_active() const { return static_cast<int>(active_.size()); }
  int num_lost() const { return static_cast<int>(lost_.size()); }
  // rows of [id, mean(8), cov(64)] over active_ then lost_
  std::vector<std::vector<float>> dump_states() const {
    std::vector<std::vector<float>> v;
    auto push = [&](const STrack& t) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id));
      for (int i = 0; i < 8; ++i) r.push_back(t.kf.mean[i]);
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) r.push_back(t.kf.cov[i][j]);
      v.push_back(r);
    };
    for (const STrack& t : active_) push(t);
    for (const STrack& t : lost_) push(t);
    return v;
  }

 private:
  STrack make_det(const Det7& d) const {  // STrack ctor :18-37
    STrack s;
    s.xywh = xyxy2xywh(d.box());
    s.tlwh = xywh2tlwh(s.xywh);
    s.xyah = tlwh2xyah(s.tlwh);
    s.conf = d.conf; s.cls = static_cast<int>(d.cls); s.det_ind = d.ind;
    return s;
  }
  void activate(STrack& t) {  // :39-53
    t.id = ++next_id_;
    t.kf = KfXYAH::initiate(t.xyah.data());
    t.has_state = true;
    t.tracklet_len = 0; t.state = Tracked;
    if (frame_id_ == 1) t.activated = true;
    t.frame_id = frame_id_; t.start_frame = frame_id_;
  }
  void reactivate(STrack& t, const STrack& d) {  // :55-69
    KfXYAH::update(t.kf, d.xyah.data());
    t.tracklet_len = 0; t.state = Tracked; t.activated = true; t.frame_id = frame_id_;
    t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
  }
  void apply_update(STrack& t, const STrack& d) {  // :71-89
    t.frame_id = frame_id_; ++t.tracklet_len;
    KfXYAH::update(t.kf, d.xyah.data());
    t.state = Tracked; t.activated = true;
    t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
  }
  static std::vector<STrack> joint(const std::vector<STrack>& a, const std::vector<STrack>& b) {  // :623-640
    std::unordered_set<int> seen;
    std::vector<STrack> r = a;
    for (const STrack& t : a) seen.insert(t.id);
    for (const STrack& t : b)
      if (seen.insert(t.id).second) r.push_back(t);
    return r;
  }
  static std::vector<STrack> sub(const std::vector<STrack>& a, const std::vector<STrack>& b) {  // :642-657
    std::unordered_set<int> rm;
    for (const STrack& t : b) rm.insert(t.id);
    std::vector<STrack> r;
    for (const STrack& t : a)
      if (!rm.count(t.id)) r.push_back(t);
    return r;
  }
  void remove_duplicates() {  // :659-706
    if (active_.empty() || lost_.empty()) return;
    std::vector<Box> ab, lb;
    for (const STrack& t : active_) ab.push_back(t.xyxy());
    for (const STrack& t : lost_) lb.push_back(t.xyxy());
    Mat pd = iou_distance(boxes_to_mat(ab), boxes_to_mat(lb));
    std::vector<char> dupa(active_.size(), 0), dupb(lost_.size(), 0);
    for (int i = 0; i < pd.r; ++i)
      for (int j = 0; j < pd.c; ++j)
        if (pd(i, j) < 0.15f) {
          int tp = active_[i].frame_id - active_[i].start_frame;
          int tq = lost_[j].frame_id - lost_[j].start_frame;
          if (tp > tq) dupb[j] = 1; else dupa[i] = 1;
        }
    std::vector<STrack> ra, rb;
    for (size_t i = 0; i < active_.size(); ++i) if (!dupa[i]) ra.push_back(active_[i]);
    for (size_t j = 0; j < lost_.size(); ++j) if (!dupb[j]) rb.push_back(lost_[j]);
    active_ = std::move(ra); lost_ = std::move(rb);
  }

  BaseParams p_;
  float min_conf_, track_thresh_, match_thresh_;
  int buffer_size_, max_time_lost_;
  int frame_count_ = 0, frame_id_ = 0, next_id_ = 0;
  std::vector<STrack> active_, lost_;
};

// =======================================================================================
// OC-SORT — ocsort.cpp
// =======================================================================================
class OCSort {
 public:
  explicit OCSort(float det_thresh = 0.2f, int max_age = 30, int max_obs = 50, int min_hits = 3,
                  float iou_threshold = 0.3f, float min_conf = 0.1f, int delta_t = 3,
                  float inertia = 0.2f, bool use_byte = false, float q_xy = 0.01f, float q_s = 0.0001f)
      : min_conf_(min_conf), asso_thr_(iou_threshold), delta_t_(delta_t), inertia_(inertia),
        use_byte_(use_byte), q_xy_(q_xy), q_s_(q_s) {
    p_.det_thresh = det_thresh; p_.max_age = max_age; p_.max_obs = max_obs;
    p_.min_hits = min_hits; p_.iou_threshold = iou_threshold; p_.normalise();
  }
  // asso_func (BaseTracker ctor argument, ocsort.cpp:413,438,494) and the frame size it is built with (:295-296)
  void set_asso(int kind, int img_w, int img_h) { asso_kind_ = kind; img_w_ = img_w; img_h_ = img_h; }
  Mat asso(const Mat& a, const Mat& b) const { return asso_batch(asso_kind_, a, b, img_w_, img_h_); }
  void reset() { frame_count_ = 0; trk_.clear(); }

  struct Obs5 { float v[5]; };
  struct Track {  // KalmanBoxTracker, ocsort.cpp:53-156
    KfXYSR kf;
    int id, age = 0, hits = 0, hit_streak = 0, tsu = 0, cls = 0, det_ind = 0;
    float conf = 0.f;
    Obs5 last_obs{{-1, -1, -1, -1, -1}};
    std::map<int, Obs5> observations;  // age -> box (+conf)
    float vel[2] = {0.f, 0.f};         // (dy, dx)
    std::vector<float> emb;            // DeepOC-SORT only (DeepOCSortKalmanBoxTracker::emb_)
  };

  OutTable update(const float* dets, int n) {  // ocsort.cpp:285-606
    std::vector<Det7> all = wrap_dets(dets, n);
    ++frame_count_;
    laps.clear();
    std::vector<Det7> high, second;
    for (const Det7& d : all) {
      if (d.conf > min_conf_ && d.conf < p_.det_thresh) second.push_back(d);
      if (d.conf > p_.det_thresh) high.push_back(d);
    }

    size_t nt = trk_.size();
    std::vector<std::array<float, 5>> trks(nt);
    std::vector<int> to_del;
    for (size_t t = 0; t < nt; ++t) {
      Box pos = predict(trk_[t]);
      trks[t] = {pos[0], pos[1], pos[2], pos[3], 0.0f};
      if (std::isnan(pos[0]) || std::isnan(pos[1]) || std::isnan(pos[2]) || std::isnan(pos[3]))
        to_del.push_back(static_cast<int>(t));
    }
    for (auto it = to_del.rbegin(); it != to_del.rend(); ++it) { trk_.erase(trk_.begin() + *it); --nt; }
    trks.resize(nt);  // conservativeResize keeps the FIRST nt rows (:363-364), NaN rows included

    if (nt == 0) {  // :366-383
      for (const Det7& d : high) trk_.push_back(spawn(d));
      return {};
    }

    Mat vel(static_cast<int>(nt), 2), kobs(static_cast<int>(nt), 5);
    for (size_t t = 0; t < nt; ++t) {
      vel(t, 0) = trk_[t].vel[0]; vel(t, 1) = trk_[t].vel[1];
      Obs5 k = k_previous_obs(trk_[t], delta_t_);
      for (int c = 0; c < 5; ++c) kobs(t, c) = k.v[c];
    }
    Mat dm(static_cast<int>(high.size()), 5), tm(static_cast<int>(nt), 5);
    for (int i = 0; i < dm.r; ++i) { dm(i,0)=high[i].x1; dm(i,1)=high[i].y1; dm(i,2)=high[i].x2; dm(i,3)=high[i].y2; dm(i,4)=high[i].conf; }
    for (int i = 0; i < tm.r; ++i) for (int c = 0; c < 5; ++c) tm(i, c) = trks[i][c];

    Assoc as = associate(dm, tm, asso_thr_, vel, kobs, inertia_);
    for (const auto& m : as.matches) apply(trk_[m[1]], high[m[0]]);

    if (use_byte_ && !second.empty() && !as.um_trks.empty()) {  // :430-472
      Mat ut(static_cast<int>(as.um_trks.size()), 5), sd(static_cast<int>(second.size()), 5);
      for (int i = 0; i < ut.r; ++i) for (int c = 0; c < 5; ++c) ut(i, c) = trks[as.um_trks[i]][c];
      for (int i = 0; i < sd.r; ++i) { sd(i,0)=second[i].x1; sd(i,1)=second[i].y1; sd(i,2)=second[i].x2; sd(i,3)=second[i].y2; sd(i,4)=second[i].conf; }
      Mat iou = asso(sd, ut);
      float mx = -std::numeric_limits<float>::infinity();
      for (float v : iou.a) mx = std::max(mx, v);
      if (mx > asso_thr_) {
        Mat cost = iou; for (float& v : cost.a) v = -v;
        LapResult r = linear_assignment(cost, -asso_thr_);
        laps.push_back(r);
        std::unordered_set<int> rm;
        for (const auto& m : r.matches) {
          int ti = as.um_trks[m[1]];
          if (iou(m[0], m[1]) < asso_thr_) continue;
          apply(trk_[ti], second[m[0]]);
          rm.insert(ti);
        }
        std::vector<int> keep;
        for (int t : as.um_trks) if (!rm.count(t)) keep.push_back(t);
        as.um_trks = keep;
      }
    }

    if (!as.um_dets.empty() && !as.um_trks.empty()) {  // OCR rematch :475-540
      Mat ld(static_cast<int>(as.um_dets.size()), 4), lt(static_cast<int>(as.um_trks.size()), 4);
      for (int i = 0; i < ld.r; ++i) { const Det7& d = high[as.um_dets[i]]; ld(i,0)=d.x1; ld(i,1)=d.y1; ld(i,2)=d.x2; ld(i,3)=d.y2; }
      for (int i = 0; i < lt.r; ++i) for (int c = 0; c < 4; ++c) lt(i, c) = trk_[as.um_trks[i]].last_obs.v[c];
      Mat iou = asso(ld, lt);
      float mx = -std::numeric_limits<float>::infinity();
      for (float v : iou.a) mx = std::max(mx, v);
      if (mx > asso_thr_) {
        Mat cost = iou; for (float& v : cost.a) v = -v;
        LapResult r = linear_assignment(cost, -asso_thr_);
        laps.push_back(r);
        std::unordered_set<int> rmd, rmt;
        for (const auto& m : r.matches) {
          int di = as.um_dets[m[0]], ti = as.um_trks[m[1]];
          if (iou(m[0], m[1]) < asso_thr_) continue;
          apply(trk_[ti], high[di]);
          rmd.insert(di); rmt.insert(ti);
        }
        std::vector<int> kd, kt;
        for (int d : as.um_dets) if (!rmd.count(d)) kd.push_back(d);
        for (int t : as.um_trks) if (!rmt.count(t)) kt.push_back(t);
        as.um_dets = kd; as.um_trks = kt;
      }
    }
    // unmatched tracks get a "None" update: det_ind = 0 and a KF no-op (:543-545, :89-124, xysr_kf.cpp:80-82)
    for (int t : as.um_trks) trk_[t].det_ind = 0;
    for (int d : as.um_dets) trk_.push_back(spawn(high[d]));

    OutTable out;
    for (int i = static_cast<int>(trk_.size()) - 1; i >= 0; --i) {  // :562-587
      Track& t = trk_[i];
      Box d;
      float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
      if (ls < 0) d = state_box(t);
      else d = {t.last_obs.v[0], t.last_obs.v[1], t.last_obs.v[2], t.last_obs.v[3]};
      if (t.tsu < 1 && (t.hit_streak >= p_.min_hits || frame_count_ <= p_.min_hits))
        out.push_back({d[0], d[1], d[2], d[3], static_cast<float>(t.id + 1), t.conf,
                       static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
      if (t.tsu > p_.max_age) trk_.erase(trk_.begin() + i);
    }
    return out;
  }

  std::vector<LapResult> laps;
  int num_tracks() const { return static_cast<int>(trk_.size()); }
  std::vector<std::vector<float>> dump_states() const {
    std::vector<std::vector<float>> v;
    for (const Track& t : trk_) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id));
      for (int i = 0; i < 7; ++i) r.push_back(t.kf.x[i]);
      for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) r.push_back(t.kf.P[i][j]);
      v.push_back(r);
    }
    return v;
  }

  struct Assoc {
    std::vector<std::array<int, 2>> matches;  // (det, trk)
    std::vector<int> um_dets, um_trks;
  };
  // ocsort_assoc::associate, ocsort.cpp:610-738. Public so primitive-level parity tests can call it.
  Assoc associate(const Mat& dets, const Mat& trks, float thr, const Mat& vel, const Mat& prev, float vdc) {
    Assoc R;
    const int nd = dets.r, ntk = trks.r;
    if (ntk == 0) {
      for (int i = 0; i < nd; ++i) R.um_dets.push_back(i);
      return R;
    }
    Mat angle(nd, ntk);  // already transposed to (dets x trks) and score-weighted
    for (int i = 0; i < ntk; ++i) {
      for (int j = 0; j < nd; ++j) {
        float cx1 = (dets(j, 0) + dets(j, 2)) / 2.0f, cy1 = (dets(j, 1) + dets(j, 3)) / 2.0f;
        float cx2 = (prev(i, 0) + prev(i, 2)) / 2.0f, cy2 = (prev(i, 1) + prev(i, 3)) / 2.0f;
        float dx = cx1 - cx2, dy = cy1 - cy2;
        float norm = std::sqrt(dx * dx + dy * dy) + 1e-6f;
        float Y = dy / norm, X = dx / norm;
        float c = vel(i, 1) * X + vel(i, 0) * Y;
        c = std::min(std::max(c, -1.0f), 1.0f);
        const float PI = 3.14159265358979323846f;
        float da = (PI / 2.0f - std::fabs(acos_f32(c))) / PI;
        float valid = (prev(i, 4) >= 0.0f) ? 1.0f : 0.0f;
        float a = (valid * da) * vdc;
        angle(j, i) = a * dets(j, 4);
      }
    }
    Mat iou = asso(dets, trks);
    last_iou = iou;
    if (nd > 0) {
      int max_row = 0, max_col = 0;
      std::vector<int> colsum(ntk, 0);
      for (int i = 0; i < nd; ++i) {
        int rs = 0;
        for (int j = 0; j < ntk; ++j)
          if (iou(i, j) > thr) { ++rs; ++colsum[j]; }
        max_row = std::max(max_row, rs);
      }
      for (int j = 0; j < ntk; ++j) max_col = std::max(max_col, colsum[j]);
      if (max_row == 1 && max_col == 1) {
        for (int i = 0; i < nd; ++i)
          for (int j = 0; j < ntk; ++j)
            if (iou(i, j) > thr) R.matches.push_back({i, j});
      } else {
        Mat fc(nd, ntk);
        for (int i = 0; i < nd; ++i)
          for (int j = 0; j < ntk; ++j) fc(i, j) = -(iou(i, j) + angle(i, j));
        LapResult r = linear_assignment(fc, -thr);
        laps.push_back(r);
        for (const auto& m : r.matches) {
          if (iou(m[0], m[1]) >= thr) R.matches.push_back({m[0], m[1]});
          else { R.um_dets.push_back(m[0]); R.um_trks.push_back(m[1]); }  // Q4: pushed again by the sweep below
        }
      }
    }
    std::unordered_set<int> md, mt;
    for (const auto& m : R.matches) { md.insert(m[0]); mt.insert(m[1]); }
    for (int i = 0; i < nd; ++i) if (!md.count(i)) R.um_dets.push_back(i);
    for (int i = 0; i < ntk; ++i) if (!mt.count(i)) R.um_trks.push_back(i);
    return R;
  }
  Mat last_iou;

  // acos in fp32. The reference calls std::acos(float) (glibc acosf, <1 ulp but not correctly
  // rounded, libm-version dependent). The oracle fixes a canonical value: the correctly
  // rounded fp32 result obtained through double precision, which any platform can reproduce.
  static float acos_f32(float c) { return static_cast<float>(std::acos(static_cast<double>(c))); }

 protected:
  Track spawn(const Det7& d) {  // ctor :53-87
    Track t;
    t.id = ++next_id_;
    t.conf = d.conf; t.cls = static_cast<int>(d.cls); t.det_ind = d.ind;
    t.kf.Q[4][4] *= q_xy_; t.kf.Q[5][5] *= q_xy_; t.kf.Q[6][6] *= q_s_;  // Q5: scaling applied on top of the ctor values
    Box z = xyxy2xysr(d.box());
    for (int i = 0; i < 4; ++i) t.kf.x[i] = z[i];
    return t;
  }
  static Box state_box(const Track& t) {  // convert_x_to_bbox_impl :177-186 (no clamping)
    return xysr2xyxy({t.kf.x[0], t.kf.x[1], t.kf.x[2], t.kf.x[3]});
  }
  static Box predict(Track& t) {  // :132-148
    if ((t.kf.x[6] + t.kf.x[2]) <= 0) t.kf.x[6] = 0.0f;
    t.kf.predict();
    ++t.age;
    if (t.tsu > 0) t.hit_streak = 0;
    ++t.tsu;
    return state_box(t);
  }
  static Obs5 k_previous_obs(const Track& t, int k) {  // :24-51
    if (t.observations.empty()) return Obs5{{-1, -1, -1, -1, -1}};
    for (int i = 0; i < k; ++i) {
      auto it = t.observations.find(t.age - (k - i));
      if (it != t.observations.end()) return it->second;
    }
    return t.observations.rbegin()->second;  // max age key
  }
  static void speed_direction(const float* b1, const float* b2, float out[2]) {  // :160-172
    float cx1 = (b1[0] + b1[2]) / 2.0f, cy1 = (b1[1] + b1[3]) / 2.0f;
    float cx2 = (b2[0] + b2[2]) / 2.0f, cy2 = (b2[1] + b2[3]) / 2.0f;
    float dy = cy2 - cy1, dx = cx2 - cx1;
    float norm = std::sqrt(dy * dy + dx * dx) + 1e-6f;
    out[0] = dy / norm; out[1] = dx / norm;
  }
  void apply(Track& t, const Det7& d) {  // update :89-130
    t.det_ind = d.ind;
    t.conf = d.conf; t.cls = static_cast<int>(d.cls);
    const float box[4] = {d.x1, d.y1, d.x2, d.y2};
    float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
    if (ls >= 0) {
      Obs5 pb = k_previous_obs(t, delta_t_);
      if (pb.v[0] + pb.v[1] + pb.v[2] + pb.v[3] >= 0) speed_direction(pb.v, box, t.vel);
      else speed_direction(t.last_obs.v, box, t.vel);
    }
    for (int i = 0; i < 4; ++i) t.last_obs.v[i] = box[i];
    t.last_obs.v[4] = t.conf;
    t.observations[t.age] = t.last_obs;
    t.tsu = 0; ++t.hits; ++t.hit_streak;
    Box z = xyxy2xysr(d.box());
    t.kf.update(z.data());
  }

  BaseParams p_;
  int asso_kind_ = ASSO_IOU, img_w_ = 1920, img_h_ = 1080;
  float min_conf_, asso_thr_;
  int delta_t_;
  float inertia_;
  bool use_byte_;
  float q_xy_, q_s_;
  int frame_count_ = 0, next_id_ = 0;
  std::vector<Track> trk_;
};

// =======================================================================================
// DeepOC-SORT — deepocsort.cpp. OC-SORT's lifecycle plus: an embedding per track (normalised when its
// norm exceeds 1e-6, :73-79; EMA with a per-detection alpha, :132-150), the similarity dets_embs * trk_embs^T (:746-757)
// zeroed where the IoU is not positive and weighted adaptively (compute_aw_max_metric :294-346) or by a constant, added
// to IoU + direction in the association cost (:437-481); detections kept are conf > det_thresh only (no BYTE stage);
// in the LAP branch every unmatched detection / track ends up in the unmatched lists TWICE (:456-503: the assignment's
// unmatched lists are appended, then the sweep appends everything that is not in a match again), so the OCR stage sees
// duplicated rows / columns and each unmatched detection spawns two tracks; camera motion (caller-supplied 2x3 warp, :633-643)
// also moves last_observation and the observations inside the delta_t window (:189-236). ReID inference and the image
// registration are outside the path: embeddings and warps are passed in. Ids are per instance, from 1 (:27-30).
// =======================================================================================
class DeepOCSort : public OCSort {
 public:
  explicit DeepOCSort(float det_thresh = 0.3f, int max_age = 30, int max_obs = 50, int min_hits = 3, float iou_threshold = 0.3f,
                      int delta_t = 3, float inertia = 0.2f, float w_emb = 0.5f, float alpha_fixed = 0.95f, float aw_param = 0.5f,
                      bool embedding_off = false, bool cmc_off = false, bool aw_off = false, float q_xy = 0.01f, float q_s = 0.0001f)
      : OCSort(det_thresh, max_age, max_obs, min_hits, iou_threshold, 0.1f, delta_t, inertia, false, q_xy, q_s),
        w_emb_(w_emb), alpha_fixed_(alpha_fixed), aw_param_(aw_param), emb_off_(embedding_off), cmc_off_(cmc_off), aw_off_(aw_off) {}
  void set_warp(const float w2x3[6]) { for (int i = 0; i < 6; ++i) warp_[i] = w2x3[i]; has_warp_ = true; }
  void reset() { OCSort::reset(); next_id_ = 0; has_warp_ = false; }

  // compute_aw_max_metric :294-346 (emb: nd x nt, already zeroed where IoU <= 0)
  static Mat aw_max_metric(const Mat& emb, float w, float bottom) {
    Mat wm(emb.r, emb.c, w);
    auto top2 = [](std::vector<float> v, float* mx, float* second) {  // descending sort of the values
      std::sort(v.begin(), v.end(), [](float a, float b) { return a > b; });
      *mx = v[0]; *second = v[1];
    };
    for (int i = 0; i < emb.r; ++i) {
      if (emb.c < 2) continue;
      std::vector<float> v(emb.c);
      for (int j = 0; j < emb.c; ++j) v[j] = emb(i, j);
      float mx, sc;
      top2(v, &mx, &sc);
      if (mx == 0.0f) { for (int j = 0; j < emb.c; ++j) wm(i, j) = 0.0f; }
      else {
        const float rw = 1.0f - std::max((sc / mx) - bottom, 0.0f) / (1.0f - bottom);
        for (int j = 0; j < emb.c; ++j) wm(i, j) *= rw;
      }
    }
    for (int j = 0; j < emb.c; ++j) {
      if (emb.r < 2) continue;
      std::vector<float> v(emb.r);
      for (int i = 0; i < emb.r; ++i) v[i] = emb(i, j);
      float mx, sc;
      top2(v, &mx, &sc);
      if (mx == 0.0f) { for (int i = 0; i < emb.r; ++i) wm(i, j) = 0.0f; }
      else {
        const float cw = 1.0f - std::max((sc / mx) - bottom, 0.0f) / (1.0f - bottom);
        for (int i = 0; i < emb.r; ++i) wm(i, j) *= cw;
      }
    }
    Mat out(emb.r, emb.c);
    for (int i = 0; i < emb.r; ++i) for (int j = 0; j < emb.c; ++j) out(i, j) = wm(i, j) * emb(i, j);
    return out;
  }

  OutTable update(const float* dets, int n, const float* embs, int d) {  // :589-945
    std::vector<Det7> all = wrap_dets(dets, n);
    ++frame_count_;
    laps.clear();
    std::vector<Det7> high;
    for (const Det7& dd : all) if (dd.conf > p_.det_thresh) high.push_back(dd);
    const int nd = static_cast<int>(high.size());
    // dets_embs :619-633 — ones when the embedding is off (or nothing survived), the supplied rows otherwise
    int D = 1;
    std::vector<std::vector<float>> de(nd);
    if (emb_off_ || nd == 0 || embs == nullptr || d <= 0) for (auto& r : de) r.assign(1, 1.0f);
    else { D = d; for (int i = 0; i < nd; ++i) de[i].assign(embs + static_cast<size_t>(high[i].ind) * d, embs + static_cast<size_t>(high[i].ind + 1) * d); }
    const bool warp_now = has_warp_;
    has_warp_ = false;
    if (!cmc_off_ && warp_now) {  // :633-643
      const float m[2][2] = {{warp_[0], warp_[1]}, {warp_[3], warp_[4]}}, t[2] = {warp_[2], warp_[5]};
      for (Track& tr : trk_) affine(tr, m, t);
    }
    std::vector<float> alpha(nd);
    for (int i = 0; i < nd; ++i) {  // :646-648
      const float trust = (high[i].conf - p_.det_thresh) / (1.0f - p_.det_thresh);
      alpha[i] = alpha_fixed_ + (1.0f - alpha_fixed_) * (1.0f - trust);
    }
    auto spawn_all = [&]() { for (int i = 0; i < nd; ++i) trk_.push_back(spawn_deep(high[i], de[i])); };
    size_t nt = trk_.size();
    if (nt == 0) { spawn_all(); return {}; }  // :652-664
    std::vector<std::array<float, 5>> trks(nt);
    std::vector<int> to_del;
    for (size_t t = 0; t < nt; ++t) {
      Box pos = predict(trk_[t]);
      trks[t] = {pos[0], pos[1], pos[2], pos[3], 0.0f};
      if (std::isnan(pos[0]) || std::isnan(pos[1]) || std::isnan(pos[2]) || std::isnan(pos[3])) to_del.push_back(static_cast<int>(t));
    }
    for (auto it = to_del.rbegin(); it != to_del.rend(); ++it) { trk_.erase(trk_.begin() + *it); --nt; }
    trks.resize(nt);  // the FIRST nt rows (:690), like OC-SORT
    if (nt == 0) { spawn_all(); return {}; }  // :692-705
    Mat vel(static_cast<int>(nt), 2), kobs(static_cast<int>(nt), 5);
    for (size_t t = 0; t < nt; ++t) {
      vel(t, 0) = trk_[t].vel[0]; vel(t, 1) = trk_[t].vel[1];
      Obs5 k = k_previous_obs(trk_[t], delta_t_);
      for (int c = 0; c < 5; ++c) kobs(t, c) = k.v[c];
    }
    // track embedding matrix :722-741 (the surviving tracks' embeddings in order; dimension of the first one)
    int TD = trk_[0].emb.empty() ? 0 : static_cast<int>(trk_[0].emb.size());
    Mat emb;  // nd x nt similarity, empty when off (:744-760)
    if (!emb_off_ && nd > 0) {
      emb = Mat(nd, static_cast<int>(nt), 0.0f);
      const int tdim = TD > 0 ? TD : D;
      if (D == tdim)
        for (int i = 0; i < nd; ++i)
          for (size_t t = 0; t < nt; ++t) {
            const std::vector<float>& te = trk_[t].emb;
            std::vector<float> row(tdim, TD > 0 ? 0.0f : 1.0f);
            if (TD > 0 && static_cast<int>(te.size()) == TD) row = te;
            emb(i, static_cast<int>(t)) = dot_chain(de[i].data(), row.data(), tdim);
          }
    }
    Mat dm(nd, 5), tm(static_cast<int>(nt), 5);
    for (int i = 0; i < nd; ++i) { dm(i,0)=high[i].x1; dm(i,1)=high[i].y1; dm(i,2)=high[i].x2; dm(i,3)=high[i].y2; dm(i,4)=high[i].conf; }
    for (int i = 0; i < tm.r; ++i) for (int c = 0; c < 5; ++c) tm(i, c) = trks[i][c];
    Assoc as = associate_deep(dm, tm, asso_thr_, vel, kobs, inertia_, emb);
    for (const auto& m : as.matches) { apply(trk_[m[1]], high[m[0]]); update_emb(trk_[m[1]], de[m[0]], alpha[m[0]]); }
    if (!as.um_dets.empty() && !as.um_trks.empty()) {  // OCR :796-872 over the (duplicated) lists
      Mat ld(static_cast<int>(as.um_dets.size()), 4), lt(static_cast<int>(as.um_trks.size()), 4);
      for (int i = 0; i < ld.r; ++i) { const Det7& q = high[as.um_dets[i]]; ld(i,0)=q.x1; ld(i,1)=q.y1; ld(i,2)=q.x2; ld(i,3)=q.y2; }
      for (int i = 0; i < lt.r; ++i) for (int c = 0; c < 4; ++c) lt(i, c) = trk_[as.um_trks[i]].last_obs.v[c];
      Mat iou = asso(ld, lt);
      float mx = -std::numeric_limits<float>::infinity();
      for (float v : iou.a) mx = std::max(mx, v);
      if (mx > asso_thr_) {
        Mat cost = iou; for (float& v : cost.a) v = -v;
        LapResult r = linear_assignment(cost, -asso_thr_);
        laps.push_back(r);
        std::unordered_set<int> rmd, rmt;
        for (const auto& m : r.matches) {
          int di = as.um_dets[m[0]], ti = as.um_trks[m[1]];
          if (iou(m[0], m[1]) < asso_thr_) continue;
          apply(trk_[ti], high[di]);
          update_emb(trk_[ti], de[di], alpha[di]);
          rmd.insert(di); rmt.insert(ti);
        }
        std::vector<int> kd, kt;
        for (int q : as.um_dets) if (!rmd.count(q)) kd.push_back(q);
        for (int q : as.um_trks) if (!rmt.count(q)) kt.push_back(q);
        as.um_dets = kd; as.um_trks = kt;
      }
    }
    for (int t : as.um_trks) trk_[t].det_ind = 0;  // update(None) :875-877 (twice for a duplicated entry: same effect)
    for (int q : as.um_dets) trk_.push_back(spawn_deep(high[q], de[q]));  // :880-890, once per list entry
    OutTable out;
    for (int i = static_cast<int>(trk_.size()) - 1; i >= 0; --i) {  // :896-920
      Track& t = trk_[i];
      Box b;
      float ls = t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3];
      if (ls < 0) b = state_box(t);
      else b = {t.last_obs.v[0], t.last_obs.v[1], t.last_obs.v[2], t.last_obs.v[3]};
      if (t.tsu < 1 && (t.hit_streak >= p_.min_hits || frame_count_ <= p_.min_hits))
        out.push_back({b[0], b[1], b[2], b[3], static_cast<float>(t.id), t.conf, static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
      if (t.tsu > p_.max_age) trk_.erase(trk_.begin() + i);
    }
    return out;
  }
  std::vector<std::vector<float>> dump_features() const {
    std::vector<std::vector<float>> v;
    for (const Track& t : trk_) v.push_back(t.emb);
    return v;
  }

  // deepocsort_assoc::associate :351-507. emb: nd x nt similarity or empty.
  Assoc associate_deep(const Mat& dets, const Mat& trks, float thr, const Mat& vel, const Mat& prev, float vdc, const Mat& emb_in) {
    Assoc R;
    const int nd = dets.r, ntk = trks.r;
    if (ntk == 0) { for (int i = 0; i < nd; ++i) R.um_dets.push_back(i); return R; }
    Mat angle(nd, ntk);
    for (int i = 0; i < ntk; ++i)
      for (int j = 0; j < nd; ++j) {
        float cx1 = (dets(j, 0) + dets(j, 2)) / 2.0f, cy1 = (dets(j, 1) + dets(j, 3)) / 2.0f;
        float cx2 = (prev(i, 0) + prev(i, 2)) / 2.0f, cy2 = (prev(i, 1) + prev(i, 3)) / 2.0f;
        float dx = cx1 - cx2, dy = cy1 - cy2;
        float norm = std::sqrt(dx * dx + dy * dy) + 1e-6f;
        float Y = dy / norm, X = dx / norm;
        float c = vel(i, 1) * X + vel(i, 0) * Y;
        c = std::min(std::max(c, -1.0f), 1.0f);
        const float PI = 3.14159265358979323846f;
        float da = (PI / 2.0f - std::fabs(acos_f32(c))) / PI;
        float valid = (prev(i, 4) >= 0.0f) ? 1.0f : 0.0f;
        angle(j, i) = ((valid * da) * vdc) * dets(j, 4);
      }
    Mat iou = asso(dets, trks);
    last_iou = iou;
    Mat fe(nd, ntk, 0.0f);
    if (emb_in.r > 0 && emb_in.c > 0) {  // :419-436
      fe = emb_in;
      for (int i = 0; i < nd; ++i) for (int j = 0; j < ntk; ++j) if (iou(i, j) <= 0.0f) fe(i, j) = 0.0f;
      if (!aw_off_) fe = aw_max_metric(fe, w_emb_, aw_param_);
      else for (float& v : fe.a) v *= w_emb_;
    }
    if (nd > 0) {
      int max_row = 0, max_col = 0;
      std::vector<int> colsum(ntk, 0);
      for (int i = 0; i < nd; ++i) {
        int rs = 0;
        for (int j = 0; j < ntk; ++j) if (iou(i, j) > thr) { ++rs; ++colsum[j]; }
        max_row = std::max(max_row, rs);
      }
      for (int j = 0; j < ntk; ++j) max_col = std::max(max_col, colsum[j]);
      if (max_row == 1 && max_col == 1) {
        for (int i = 0; i < nd; ++i) for (int j = 0; j < ntk; ++j) if (iou(i, j) > thr) R.matches.push_back({i, j});
      } else {
        Mat fc(nd, ntk);
        for (int i = 0; i < nd; ++i) for (int j = 0; j < ntk; ++j) fc(i, j) = -((iou(i, j) + angle(i, j)) + fe(i, j));
        LapResult r = linear_assignment(fc, -thr);
        laps.push_back(r);
        for (const auto& m : r.matches) {
          if (iou(m[0], m[1]) >= thr) R.matches.push_back({m[0], m[1]});
          else { R.um_dets.push_back(m[0]); R.um_trks.push_back(m[1]); }
        }
        for (int i : r.unmatched_a) R.um_dets.push_back(i);  // :484-489 — and again by the sweep below
        for (int j : r.unmatched_b) R.um_trks.push_back(j);
      }
    }
    std::unordered_set<int> md, mt;
    for (const auto& m : R.matches) { md.insert(m[0]); mt.insert(m[1]); }
    for (int i = 0; i < nd; ++i) if (!md.count(i)) R.um_dets.push_back(i);
    for (int i = 0; i < ntk; ++i) if (!mt.count(i)) R.um_trks.push_back(i);
    return R;
  }

 private:
  static void normalise(std::vector<float>& e) {  // norm > 1e-6 rule (:73-79, :146-149)
    if (e.empty()) return;
    const float nn = std::sqrt(dot_chain(e.data(), e.data(), static_cast<int>(e.size())));
    if (nn > 1e-6f) for (float& v : e) v /= nn;
  }
  Track spawn_deep(const Det7& d, const std::vector<float>& e) {
    Track t = spawn(d);  // same constructor arithmetic as OC-SORT (:53-87 / deepocsort.cpp:50-99); ids from 1
    t.emb = e;
    normalise(t.emb);
    return t;
  }
  static void update_emb(Track& t, const std::vector<float>& e, float alpha) {  // :132-150
    if (e.empty()) return;
    if (t.emb.empty()) t.emb = e;
    else for (size_t k = 0; k < e.size(); ++k) t.emb[k] = alpha * t.emb[k] + (1.0f - alpha) * e[k];
    normalise(t.emb);
  }
  void affine(Track& t, const float m[2][2], const float tr[2]) const {  // apply_affine_correction :189-236
    auto move = [&](float* o) {  // corners through m, then + t (Eigen: m * corners, column += t)
      const float x1 = o[0], y1 = o[1], x2 = o[2], y2 = o[3];
      o[0] = (m[0][0] * x1 + m[0][1] * y1) + tr[0];
      o[1] = (m[1][0] * x1 + m[1][1] * y1) + tr[1];
      o[2] = (m[0][0] * x2 + m[0][1] * y2) + tr[0];
      o[3] = (m[1][0] * x2 + m[1][1] * y2) + tr[1];
    };
    if (t.last_obs.v[0] + t.last_obs.v[1] + t.last_obs.v[2] + t.last_obs.v[3] > 0) move(t.last_obs.v);
    for (int dt = delta_t_; dt >= 0; --dt) {
      auto it = t.observations.find(t.age - dt);
      if (it != t.observations.end() && it->second.v[0] + it->second.v[1] + it->second.v[2] + it->second.v[3] > 0) move(it->second.v);
    }
    t.kf.apply_affine_correction(m, tr);
  }
  float w_emb_, alpha_fixed_, aw_param_;
  bool emb_off_, cmc_off_, aw_off_;
  float warp_[6] = {1, 0, 0, 0, 1, 0};
  bool has_warp_ = false;
};

// =======================================================================================
// BoT-SORT — botsort.cpp (no CMC, no ReID model: embeddings are passed in, N x D row-major)
// =======================================================================================
class BotSort {
 public:
  explicit BotSort(float track_high = 0.5f, float track_low = 0.1f, float new_track = 0.6f,
                   int track_buffer = 30, float match_thresh = 0.8f, float proximity = 0.5f,
                   float appearance = 0.25f, int frame_rate = 30, bool fuse_first = false,
                   bool with_reid = true, int max_age = 30, int max_obs = 50)
      : hi_(track_high), lo_(track_low), newt_(new_track), match_(match_thresh), prox_(proximity),
        app_(appearance), fuse_first_(fuse_first), with_reid_(with_reid) {
    p_.max_age = max_age; p_.max_obs = max_obs; p_.normalise();
    max_time_lost_ = static_cast<int>(frame_rate / 30.0f * track_buffer);  // botsort.cpp:236-237
  }
  void reset() { frame_count_ = 0; active_.clear(); lost_.clear(); next_id_ = 0; }  // :252-258
  // The warp cmc_->apply(img, dets) would return for the next frame (:317-324), supplied by the caller: 2x3 row-major.
  // It is consumed by the next update(), used or not (an empty frame returns before the CMC step, :267-269).
  void set_warp(const float w2x3[6]) {
    for (int i = 0; i < 6; ++i) warp_[i / 3][i % 3] = w2x3[i];
    warp_[2][0] = 0.0f; warp_[2][1] = 0.0f; warp_[2][2] = 1.0f;  // Matrix3f::Identity() with the top rows replaced (:320-321)
    has_warp_ = true;
  }

  enum State { New = 0, Tracked = 1, Lost = 2, Removed = 3 };
  struct BTrack {  // BotSTrack :23-193
    Box xywh;
    float conf = 0.f;
    int cls = 0, det_ind = -1, id = 0, frame_id = 0, start_frame = 0, end_frame = 0, tracklet_len = 0;
    bool has_state = false;
    State8 kf;
    State state = New;
    bool activated = false;
    std::vector<float> curr_feat, smooth_feat;
    Box xyxy() const {  // :171-181
      float cx, cy, w, h;
      if (has_state) { cx = kf.mean[0]; cy = kf.mean[1]; w = kf.mean[2]; h = kf.mean[3]; }
      else { cx = xywh[0]; cy = xywh[1]; w = xywh[2]; h = xywh[3]; }
      return {cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2};
    }
  };

  // BotSTrack::multi_gmc :60-91 for one track
  static void gmc(BTrack& t, const float W[3][3]) {
    const Box b = t.xyxy();
    float p[2][3];
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 3; ++r) p[c][r] = W[r][0] * b[2 * c] + W[r][1] * b[2 * c + 1] + W[r][2] * 1.0f;
    const float x1 = p[0][0] / p[0][2], y1 = p[0][1] / p[0][2];
    const float x2 = p[1][0] / p[1][2], y2 = p[1][1] / p[1][2];
    const float w = x2 - x1, h = y2 - y1;
    t.kf.mean[0] = x1 + w / 2.0f; t.kf.mean[1] = y1 + h / 2.0f; t.kf.mean[2] = w; t.kf.mean[3] = h;
  }

  OutTable update(const float* dets, int n, const float* embs, int emb_dim) {  // :260-359
    laps.clear();
    const bool warp_now = has_warp_;
    has_warp_ = false;
    if (n == 0) return {};  // :267-269 (no predict, no frame_count++)
    ++frame_count_;
    std::vector<Det7> all = wrap_dets(dets, n);
    std::vector<BTrack> detections, detections_second;
    for (const Det7& d : all) {  // split_detections :361-403 + create_detections :405-423
      if (d.conf > hi_) {
        BTrack b = make_det(d);
        if (with_reid_ && embs != nullptr && emb_dim > 0) set_feat(b, embs + static_cast<size_t>(d.ind) * emb_dim, emb_dim);
        detections.push_back(std::move(b));
      } else if (d.conf > lo_) {
        detections_second.push_back(make_det(d));
      }
    }
    active_.reserve(active_.size() + detections.size() + 10);
    lost_.reserve(lost_.size() + active_.size() + 10);
    std::vector<BTrack*> unconfirmed, act_ptrs, lost_ptrs;
    for (BTrack& t : active_) (t.activated ? act_ptrs : unconfirmed).push_back(&t);
    for (BTrack& t : lost_) lost_ptrs.push_back(&t);
    std::vector<BTrack*> pool = joint(act_ptrs, lost_ptrs);
    for (BTrack* t : pool) KfXYWH::predict(t->kf);  // multi_predict :54-58 (in place)
    if (warp_now) {  // :317-324
      for (BTrack* t : pool) if (t->has_state) gmc(*t, warp_);
      for (BTrack* t : unconfirmed) if (t->has_state) gmc(*t, warp_);
    }

    std::vector<BTrack*> activated, refind, lost_new, removed_new;

    // first association :425-495
    std::vector<int> u_track, u_det;
    {
      Mat iou_d = iou_dist_ptrs(pool, detections);
      Mat dists = assoc_cost(iou_d, pool, detections, fuse_first_);
      LapResult r = linear_assignment(dists, match_);
      laps.push_back(r);
      u_track = r.unmatched_a; u_det = r.unmatched_b;
      for (const auto& m : r.matches) {
        BTrack* t = pool[m[0]];
        if (t->state == Tracked) { apply_update(*t, detections[m[1]]); activated.push_back(t); }
        else { reactivate(*t, detections[m[1]]); refind.push_back(t); }
      }
    }
    // second association :497-563
    {
      std::vector<BTrack*> r_tracked;
      for (int i : u_track)
        if (pool[i]->state == Tracked) r_tracked.push_back(pool[i]);
      if (!r_tracked.empty() && !detections_second.empty()) {
        Mat d2 = iou_dist_ptrs(r_tracked, detections_second);
        LapResult r = linear_assignment(d2, 0.5f);
        laps.push_back(r);
        for (const auto& m : r.matches) {
          BTrack* t = r_tracked[m[0]];
          if (t->state == Tracked) { apply_update(*t, detections_second[m[1]]); activated.push_back(t); }
          else { reactivate(*t, detections_second[m[1]]); refind.push_back(t); }
        }
        for (int i : r.unmatched_a) {
          BTrack* t = r_tracked[i];
          if (t->state != Lost) { t->state = Lost; lost_new.push_back(t); }
        }
      }
    }
    // unconfirmed :565-647
    std::vector<BTrack> filtered;
    for (int i : u_det) filtered.push_back(detections[i]);
    std::vector<int> u_det_unc;
    if (unconfirmed.empty() || u_det.empty()) {
      for (size_t i = 0; i < u_det.size(); ++i) u_det_unc.push_back(static_cast<int>(i));
    } else {
      Mat iou_d = iou_dist_ptrs(unconfirmed, filtered);
      Mat dists = assoc_cost(iou_d, unconfirmed, filtered, true);
      LapResult r = linear_assignment(dists, 0.7f);
      laps.push_back(r);
      for (const auto& m : r.matches) { apply_update(*unconfirmed[m[0]], filtered[m[1]]); activated.push_back(unconfirmed[m[0]]); }
      for (int i : r.unmatched_a) { unconfirmed[i]->state = Removed; removed_new.push_back(unconfirmed[i]); }
      u_det_unc = r.unmatched_b;
    }
    // new tracks :649-667
    for (int idx : u_det_unc) {
      BTrack& d = filtered[idx];
      if (d.conf < newt_) continue;
      active_.push_back(d);
      activate(active_.back());
      activated.push_back(&active_.back());
    }
    for (BTrack& t : lost_)  // :669-676
      if (frame_count_ - t.end_frame > max_time_lost_) { t.state = Removed; removed_new.push_back(&t); }

    // prepare_output :678-764 — NB re-found lost tracks are dropped from lost_ but never
    // appended to active_ (reference behaviour, kept).
    std::unordered_set<int> active_ids;
    for (BTrack* t : activated) if (t->state == Tracked) active_ids.insert(t->id);
    for (BTrack* t : refind) if (t->state == Tracked) active_ids.insert(t->id);
    std::vector<BTrack> new_lost;
    std::unordered_set<int> lost_ids;
    for (BTrack& t : lost_)
      if (!active_ids.count(t.id) && t.state != Removed) { new_lost.push_back(t); lost_ids.insert(t.id); }
    for (BTrack* t : lost_new)
      if (!active_ids.count(t->id) && !lost_ids.count(t->id)) { new_lost.push_back(*t); lost_ids.insert(t->id); }
    std::vector<BTrack> new_active;
    for (BTrack& t : active_) if (t.state == Tracked) new_active.push_back(t);
    active_ = std::move(new_active);
    lost_ = std::move(new_lost);

    OutTable out;
    for (const BTrack& t : active_) {
      if (!t.activated) continue;
      Box b = t.xyxy();
      out.push_back({b[0], b[1], b[2], b[3], static_cast<float>(t.id), t.conf,
                     static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
    }
    return out;
  }

  std::vector<LapResult> laps;
  int num_active() const { return static_cast<int>(active_.size()); }
  int num_lost() const { return static_cast<int>(lost_.size()); }
  std::vector<std::vector<float>> dump_states() const {
    std::vector<std::vector<float>> v;
    auto push = [&](const BTrack& t) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id));
      for (int i = 0; i < 8; ++i) r.push_back(t.kf.mean[i]);
      for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) r.push_back(t.kf.cov[i][j]);
      v.push_back(r);
    };
    for (const BTrack& t : active_) push(t);
    for (const BTrack& t : lost_) push(t);
    return v;
  }
  std::vector<std::vector<float>> dump_features() const {  // smooth_feat_ of the live tracks, same order (empty: no feature yet)
    std::vector<std::vector<float>> v;
    for (const BTrack& t : active_) v.push_back(t.smooth_feat);
    for (const BTrack& t : lost_) v.push_back(t.smooth_feat);
    return v;
  }

 private:
  static BTrack make_det(const Det7& d) {  // :23-36
    BTrack b;
    float w = d.x2 - d.x1, h = d.y2 - d.y1;
    b.xywh = {d.x1 + w / 2.0f, d.y1 + h / 2.0f, w, h};
    b.conf = d.conf; b.cls = static_cast<int>(d.cls); b.det_ind = d.ind;
    return b;
  }
  static float norm(const std::vector<float>& v) {
    return std::sqrt(dot_chain(v.data(), v.data(), static_cast<int>(v.size())));
  }
  static void set_feat(BTrack& b, const float* f, int d) {  // :38-46
    b.curr_feat.assign(f, f + d);
    b.smooth_feat = b.curr_feat;
    float nn = norm(b.smooth_feat);
    if (nn > 0) for (float& v : b.smooth_feat) v /= nn;
  }
  static void update_features(BTrack& t, const std::vector<float>& feat) {  // :158-169
    t.curr_feat = feat;
    if (t.smooth_feat.empty()) t.smooth_feat = feat;
    else for (size_t k = 0; k < feat.size(); ++k) t.smooth_feat[k] = 0.9f * t.smooth_feat[k] + (1.0f - 0.9f) * feat[k];
    float nn = norm(t.smooth_feat);
    if (nn > 0) for (float& v : t.smooth_feat) v /= nn;
  }
  void activate(BTrack& t) {  // :93-109
    t.id = ++next_id_;
    t.kf = KfXYWH::initiate(t.xywh.data());
    t.has_state = true;
    t.tracklet_len = 0; t.state = Tracked;
    if (frame_count_ == 1) t.activated = true;
    t.frame_id = frame_count_; t.end_frame = frame_count_; t.start_frame = frame_count_;
  }
  void reactivate(BTrack& t, const BTrack& d) {  // :111-131
    KfXYWH::update(t.kf, d.xywh.data());
    if (!d.curr_feat.empty()) update_features(t, d.curr_feat);
    t.tracklet_len = 0; t.state = Tracked; t.activated = true;
    t.frame_id = frame_count_; t.end_frame = frame_count_;
    t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
  }
  void apply_update(BTrack& t, const BTrack& d) {  // :133-156
    t.frame_id = frame_count_; t.end_frame = frame_count_; ++t.tracklet_len;
    KfXYWH::update(t.kf, d.xywh.data());
    if (!d.curr_feat.empty()) update_features(t, d.curr_feat);
    t.state = Tracked; t.activated = true;
    t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
  }
  static std::vector<BTrack*> joint(const std::vector<BTrack*>& a, const std::vector<BTrack*>& b) {  // :767-786
    std::unordered_set<int> seen;
    std::vector<BTrack*> r;
    for (BTrack* t : a) { seen.insert(t->id); r.push_back(t); }
    for (BTrack* t : b) if (seen.insert(t->id).second) r.push_back(t);
    return r;
  }
  // matching.hpp:157-182 — Ones(m,n) when either side is empty
  static Mat iou_dist_ptrs(const std::vector<BTrack*>& a, const std::vector<BTrack>& b) {
    if (a.empty() || b.empty()) return Mat(static_cast<int>(a.size()), static_cast<int>(b.size()), 1.0f);
    std::vector<Box> ab, bb;
    for (BTrack* t : a) ab.push_back(t->xyxy());
    for (const BTrack& t : b) bb.push_back(t.xyxy());
    return iou_distance(boxes_to_mat(ab), boxes_to_mat(bb));
  }
  // :433-466 / :591-623 — IoU distance (optionally score-fused) min'ed with the gated cosine distance
  Mat assoc_cost(const Mat& iou_d, const std::vector<BTrack*>& trks, const std::vector<BTrack>& dets, bool fuse) const {
    Mat d = iou_d;
    if (fuse) {
      std::vector<float> c;
      for (const BTrack& t : dets) c.push_back(t.conf);
      d = fuse_score(d, c);
    }
    if (!with_reid_) return d;
    const int m = static_cast<int>(trks.size()), n = static_cast<int>(dets.size());
    Mat emb(m, n, 1.0f);  // matching.hpp:194-196 — Ones when either side is empty
    if (m > 0 && n > 0) {
      const int dt = static_cast<int>(trks[0]->smooth_feat.size());
      const int dd = static_cast<int>(dets[0].smooth_feat.size());
      Mat tf(m, dt, 0.0f), df(n, dd, 0.0f);
      for (int i = 0; i < m; ++i)
        if (!trks[i]->smooth_feat.empty()) for (int k = 0; k < dt; ++k) tf(i, k) = trks[i]->smooth_feat[k];
      for (int j = 0; j < n; ++j)
        if (!dets[j].smooth_feat.empty()) for (int k = 0; k < dd; ++k) df(j, k) = dets[j].smooth_feat[k];
      if (dt != dd) throw std::runtime_error("orc::BotSort: track/detection feature dims differ");
      emb = embedding_distance_cosine(tf, df);
    }
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < n; ++j) {
        float e = emb(i, j) / 2.0f;
        if (e > app_) e = 1.0f;
        if (iou_d(i, j) > prox_) e = 1.0f;
        d(i, j) = std::min(d(i, j), e);
      }
    return d;
  }

  BaseParams p_;
  float hi_, lo_, newt_, match_, prox_, app_;
  bool fuse_first_, with_reid_;
  int max_time_lost_;
  int frame_count_ = 0, next_id_ = 0;
  float warp_[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
  bool has_warp_ = false;
  std::vector<BTrack> active_, lost_;
};

// =======================================================================================
// StrongSORT — src/trackers/strongsort.cpp (round 4; "parity unpinned": the reference's tests hold no vector for it)
// =======================================================================================
// update() :847-1012 over Tracker::predict / update / match :595-816, Track :45-197, NearestNeighborDistanceMetric :203-335,
// min_cost_matching / matching_cascade :343-447, gate_cost_matrix :449-492, iou_matching :500-583. Restated as written, quirks
// included:
//  * matching_cascade and min_cost_matching replace an EMPTY index list by "all of them" (:358-368, :440-447): with no confirmed
//    track the appearance stage runs over every track; with no candidate for the IoU stage every track takes part in it; and when
//    the appearance stage left no detection unmatched the IoU stage sees ALL detections again — its own unmatched detections
//    (detections the first stage had matched among them) are then what new tracks are started from (:765-776, :608-611).
//  * the distance metric keeps, per confirmed track, the last nn_budget smoothed features (one appended per frame, matched or
//    not, :627-650) and the cost is the minimum cosine distance over them, with both sides re-normalised (:306-334).
//  * the camera-motion step (:900-906, ECC image registration) is outside this path: update() runs without it.
//  * Track's test switch (:61-76) is kept: under GITHUB_ACTIONS=true (and GITHUB_JOB != mot-metrics-benchmark) a new track is Confirmed
//    at birth — how the reference's own CI sees confirmed tracks from the second frame on. Read at every birth, as there.
// Eigen's reduction orders (norm(), the GEMM of the cosine term) are unspecified: dot_chain's canonical order, as elsewhere.
class StrongSort {
 public:
  explicit StrongSort(float min_conf = 0.1f, float max_cos_dist = 0.2f, float max_iou_dist = 0.7f, int n_init = 3, int nn_budget = 100,
                      float mc_lambda = 0.98f, float ema_alpha = 0.9f, int max_age = 30)
      : min_conf_(min_conf), max_cos_(max_cos_dist), max_iou_(max_iou_dist), n_init_(n_init), budget_(nn_budget), lambda_(mc_lambda),
        alpha_(ema_alpha), max_age_(max_age) {}
  void reset() { tracks_.clear(); next_id_ = 1; samples_.clear(); }  // :812-816

  enum State { Tentative = 1, Confirmed = 2, Deleted = 3 };
  struct Det {  // Detection :23-40
    float tlwh[4];
    float conf;
    int cls, det_ind;
    std::vector<float> feat;
    void to_xyah(float o[4]) const {
      o[0] = tlwh[0] + tlwh[2] / 2.0f; o[1] = tlwh[1] + tlwh[3] / 2.0f; o[2] = tlwh[2] / tlwh[3]; o[3] = tlwh[3];
    }
  };
  struct Track {  // :45-197
    int id = 0;
    State8 kf;
    std::vector<float> feature;  // the smoothed feature (features.back()); empty: none yet
    float conf = 0.f;
    int cls = 0, det_ind = -1, hits = 1, age = 1, tsu = 0;
    State state = Tentative;
    void to_tlwh(float o[4]) const {  // :94-100
      o[2] = kf.mean[2] * kf.mean[3]; o[3] = kf.mean[3];
      o[0] = kf.mean[0] - o[2] / 2.0f; o[1] = kf.mean[1] - o[3] / 2.0f;
    }
  };
  std::vector<LapResult> laps;

  OutTable update(const float* dets, int n, const float* embs, int emb_dim) {  // :847-1012
    laps.clear();
    std::vector<Det> D;
    const bool use_emb = embs != nullptr && emb_dim > 0;  // (embs.rows() == dets.rows() is the caller's contract here)
    for (int i = 0; i < n; ++i) {
      const float* r = dets + static_cast<size_t>(i) * 6;
      if (!(r[4] >= min_conf_)) continue;  // :873-877
      Det d;
      d.tlwh[0] = r[0]; d.tlwh[1] = r[1]; d.tlwh[2] = r[2] - r[0]; d.tlwh[3] = r[3] - r[1];  // :948-956
      d.conf = r[4]; d.cls = static_cast<int>(r[5]); d.det_ind = i;
      if (use_emb) d.feat.assign(embs + static_cast<size_t>(i) * emb_dim, embs + static_cast<size_t>(i + 1) * emb_dim);
      D.push_back(std::move(d));
    }
    for (Track& t : tracks_) { KfXYAH::predict(t.kf); ++t.age; ++t.tsu; }  // :595-599, :139-145
    tracker_update(D);
    OutTable out;
    for (const Track& t : tracks_) {  // :976-994
      if (t.state != Confirmed || t.tsu >= 1) continue;
      float b[4];
      t.to_tlwh(b);
      out.push_back({b[0], b[1], b[0] + b[2], b[1] + b[3], static_cast<float>(t.id), t.conf, static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
    }
    return out;
  }

  std::vector<std::vector<float>> dump_states() const {
    std::vector<std::vector<float>> v;
    for (const Track& t : tracks_) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id));
      for (int k = 0; k < 8; ++k) r.push_back(t.kf.mean[k]);
      for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) r.push_back(t.kf.cov[a][b]);
      v.push_back(std::move(r));
    }
    return v;
  }
  std::vector<std::vector<float>> dump_features() const {
    std::vector<std::vector<float>> v;
    for (const Track& t : tracks_) v.push_back(t.feature);
    return v;
  }

 private:
  static float norm_of(const std::vector<float>& a) { return std::sqrt(dot_chain(a.data(), a.data(), static_cast<int>(a.size()))); }
  static std::vector<float> renormed(const std::vector<float>& a) {  // cosine_distance :317-331: row /= norm when norm > 1e-10
    std::vector<float> r = a;
    const float nn = norm_of(a);
    if (nn > 1e-10f) for (float& x : r) x /= nn;
    return r;
  }
  using Matches = std::vector<std::array<int, 2>>;
  struct MatchOut { Matches m; std::vector<int> ut, ud; };

  // min_cost_matching :343-420 with the cost matrix supplied by `metric(track_idx, det_idx)`
  template <class Fn>
  MatchOut min_cost_matching(Fn metric, float max_distance, const std::vector<Det>& dets, std::vector<int> track_idx, std::vector<int> det_idx) {
    if (track_idx.empty()) { track_idx.resize(tracks_.size()); for (size_t i = 0; i < track_idx.size(); ++i) track_idx[i] = static_cast<int>(i); }
    if (det_idx.empty()) { det_idx.resize(dets.size()); for (size_t i = 0; i < det_idx.size(); ++i) det_idx[i] = static_cast<int>(i); }
    MatchOut o;
    if (track_idx.empty() || det_idx.empty()) { o.ut = track_idx; o.ud = det_idx; return o; }
    Mat cost = metric(track_idx, det_idx);
    for (float& c : cost.a) if (c > max_distance) c = max_distance + 1e-5f;  // :376-379
    LapResult r = linear_assignment(cost, max_distance);
    laps.push_back(r);
    std::vector<char> mt(track_idx.size(), 0), md(det_idx.size(), 0);
    for (const auto& p : r.matches)
      if (cost(p[0], p[1]) <= max_distance) { o.m.push_back({track_idx[p[0]], det_idx[p[1]]}); mt[p[0]] = 1; md[p[1]] = 1; }
    for (size_t i = 0; i < track_idx.size(); ++i) if (!mt[i]) o.ut.push_back(track_idx[i]);
    for (size_t j = 0; j < det_idx.size(); ++j) if (!md[j]) o.ud.push_back(det_idx[j]);
    return o;
  }

  Mat gated_metric(const std::vector<Det>& dets, const std::vector<int>& ti, const std::vector<int>& di) {  // :666-718
    const int n = static_cast<int>(ti.size()), m = static_cast<int>(di.size());
    int dim = 0;
    for (int j : di) if (!dets[j].feat.empty()) { dim = static_cast<int>(dets[j].feat.size()); break; }
    Mat cost(n, m, 1e5f);
    if (dim == 0) return cost;  // :688-690: no features at all — not even gated
    std::vector<std::vector<float>> yn(m);
    for (int j = 0; j < m; ++j) {
      std::vector<float> f(dim, 0.0f);
      if (static_cast<int>(dets[di[j]].feat.size()) == dim) f = dets[di[j]].feat;
      yn[j] = renormed(f);
    }
    for (int i = 0; i < n; ++i) {  // NearestNeighborDistanceMetric::distance :239-275
      auto it = samples_.find(tracks_[ti[i]].id);
      if (it == samples_.end() || it->second.empty()) continue;  // 1e5
      for (int j = 0; j < m; ++j) cost(i, j) = 0.0f;
      bool first = true;
      for (const std::vector<float>& xs : it->second) {
        if (static_cast<int>(xs.size()) != dim) { for (int j = 0; j < m; ++j) cost(i, j) = first ? 1.0f : std::min(cost(i, j), 1.0f); first = false; continue; }  // :313-315
        const std::vector<float> xn = renormed(xs);
        for (int j = 0; j < m; ++j) {
          const float d = 1.0f - dot_chain(xn.data(), yn[j].data(), dim);
          cost(i, j) = first ? d : std::min(cost(i, j), d);
        }
        first = false;
      }
    }
    // gate_cost_matrix :449-492
    std::vector<float> meas(static_cast<size_t>(4) * m), g(m);
    for (int j = 0; j < m; ++j) dets[di[j]].to_xyah(&meas[static_cast<size_t>(4) * j]);
    for (int i = 0; i < n; ++i) {
      gating_xyah(tracks_[ti[i]].kf, meas.data(), m, false, 0, g.data());
      for (int j = 0; j < m; ++j) {
        float c = cost(i, j);
        if (g[j] > 9.4877f) c = 1e5f;
        cost(i, j) = lambda_ * c + (1.0f - lambda_) * g[j];
      }
    }
    return cost;
  }

  Mat iou_cost(const std::vector<Det>& dets, const std::vector<int>& ti, const std::vector<int>& di) {  // :500-583
    const int n = static_cast<int>(ti.size()), m = static_cast<int>(di.size());
    Mat cost(n, m, 0.0f);
    for (int i = 0; i < n; ++i) {
      const Track& t = tracks_[ti[i]];
      if (t.tsu > 1) { for (int j = 0; j < m; ++j) cost(i, j) = 1e5f; continue; }
      float b[4];
      t.to_tlwh(b);
      const float bx2 = b[0] + b[2], by2 = b[1] + b[3], ab = b[2] * b[3];
      for (int j = 0; j < m; ++j) {
        const float* c = dets[di[j]].tlwh;
        const float cx2 = c[0] + c[2], cy2 = c[1] + c[3];
        const float tlx = std::max(b[0], c[0]), tly = std::max(b[1], c[1]);
        const float brx = std::min(bx2, cx2), bry = std::min(by2, cy2);
        const float w = std::max(0.0f, brx - tlx), h = std::max(0.0f, bry - tly);
        const float ai = w * h, ac = c[2] * c[3];
        const float au = ab + ac - ai;
        const float iou = (au > 1e-6f) ? (ai / au) : 0.0f;
        cost(i, j) = 1.0f - iou;
      }
    }
    return cost;
  }

  void track_update(Track& t, const Det& d) {  // Track::update :147-187
    float z[4];
    d.to_xyah(z);
    t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
    KfXYAH::update(t.kf, z, d.conf);
    if (!d.feat.empty()) {
      const float fn = norm_of(d.feat);
      if (!(fn < 1e-10f)) {
        std::vector<float> f = d.feat;
        for (float& x : f) x /= fn;
        if (!t.feature.empty()) {
          std::vector<float> s(f.size());
          for (size_t k = 0; k < f.size(); ++k) s[k] = alpha_ * t.feature[k] + (1.0f - alpha_) * f[k];
          const float sn = norm_of(s);
          if (sn > 1e-10f) { for (float& x : s) x /= sn; t.feature = std::move(s); }
        } else {
          t.feature = std::move(f);
        }
      }
    }
    ++t.hits; t.tsu = 0;
    if (t.state == Tentative && t.hits >= n_init_) t.state = Confirmed;
  }

  void tracker_update(const std::vector<Det>& dets) {  // Tracker::update :601-651 + match :653-806
    std::vector<int> confirmed, unconfirmed;
    for (size_t i = 0; i < tracks_.size(); ++i) (tracks_[i].state == Confirmed ? confirmed : unconfirmed).push_back(static_cast<int>(i));
    MatchOut A = min_cost_matching([&](const std::vector<int>& ti, const std::vector<int>& di) { return gated_metric(dets, ti, di); }, max_cos_, dets,
                                   confirmed, {});  // matching_cascade :422-447 = one min_cost_matching
    std::vector<int> cand = unconfirmed, ua_rest;
    for (int k : A.ut) (tracks_[k].tsu == 1 ? cand : ua_rest).push_back(k);
    MatchOut B = min_cost_matching([&](const std::vector<int>& ti, const std::vector<int>& di) { return iou_cost(dets, ti, di); }, max_iou_, dets, cand, A.ud);
    Matches matches = A.m;
    std::set<int> mt, md;
    for (const auto& p : A.m) { mt.insert(p[0]); md.insert(p[1]); }
    for (const auto& p : B.m)
      if (!mt.count(p[0]) && !md.count(p[1])) { matches.push_back(p); mt.insert(p[0]); md.insert(p[1]); }
    std::set<int> um(ua_rest.begin(), ua_rest.end());
    um.insert(B.ut.begin(), B.ut.end());
    for (const auto& p : matches) track_update(tracks_[p[0]], dets[p[1]]);
    for (int k : um) {  // mark_missed :189-197
      Track& t = tracks_[k];
      if (t.state == Tentative) t.state = Deleted;
      else if (t.tsu > max_age_) t.state = Deleted;
    }
    for (int j : B.ud) {  // initiate_track :808-810, Track::Track :45-92
      const Det& d = dets[j];
      Track t;
      t.id = next_id_++;
      {
        const char* ga = std::getenv("GITHUB_ACTIONS");
        const char* gj = std::getenv("GITHUB_JOB");
        if (ga && std::string(ga) == "true" && (!gj || std::string(gj) != "mot-metrics-benchmark")) t.state = Confirmed;
      }
      float z[4];
      d.to_xyah(z);
      t.kf = KfXYAH::initiate(z);
      t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
      if (!d.feat.empty()) {
        const float fn = norm_of(d.feat);
        if (fn > 1e-10f) { t.feature = d.feat; for (float& x : t.feature) x /= fn; }
      }
      tracks_.push_back(std::move(t));
    }
    tracks_.erase(std::remove_if(tracks_.begin(), tracks_.end(), [](const Track& t) { return t.state == Deleted; }), tracks_.end());
    // partial_fit :203-237 with the confirmed tracks' features (:627-650)
    std::vector<int> active;
    bool any = false;
    for (const Track& t : tracks_) if (t.state == Confirmed) { active.push_back(t.id); any = any || !t.feature.empty(); }
    if (any) {
      for (const Track& t : tracks_) {
        if (t.state != Confirmed || t.feature.empty()) continue;
        std::vector<std::vector<float>>& v = samples_[t.id];
        v.push_back(t.feature);
        if (budget_ > 0 && static_cast<int>(v.size()) > budget_) v.erase(v.begin(), v.end() - budget_);
      }
      std::unordered_map<int, std::vector<std::vector<float>>> kept;
      for (int id : active) { auto it = samples_.find(id); if (it != samples_.end()) kept[id] = std::move(it->second); }
      samples_ = std::move(kept);
    }
  }

  float min_conf_, max_cos_, max_iou_;
  int n_init_, budget_;
  float lambda_, alpha_;
  int max_age_;
  int next_id_ = 1;
  std::vector<Track> tracks_;
  std::unordered_map<int, std::vector<std::vector<float>>> samples_;
};

// =======================================================================================
// UCMCTrack — src/trackers/ucmc.cpp (ground-plane tracker: 4-state double-precision Kalman filter [x, vx, y, vy], Mahalanobis +
// log-determinant cost, three assignments per frame). Parity unpinned: the reference's tests hold no vector for it, and its Eigen
// products cannot be compiled here. Where the result depends on the order of a sum (the 3-term inner products of the Joseph update,
// the 3 x 3 products of the camera mapping) the terms are added in index order, k = 0, 1, 2, 3, without fused multiply-adds — the
// order of Eigen's coefficient-based product of small fixed-size matrices; every other sum of the filter has at most two
// non-zero terms (F and H are sparse), which no order can change.
// =======================================================================================
class Ucmc {
 public:
  using M4 = std::array<double, 16>;  // row-major 4 x 4
  struct Params {
    float det_thresh = 0.3f;
    int max_age = 30;
    double a1 = 100.0, a2 = 100.0, wx = 5.0, wy = 5.0, vmax = 10.0, dt = 1.0 / 30.0;
    float high_score = 0.5f;
  };
  explicit Ucmc(const Params& p) : p_(p) {}
  // CameraMapper::CameraMapper :57-83: Ki 3 x 4 and Ko 4 x 4, row-major values (what the reference's vectors hold: it maps them as
  // column-major 4 x 3 / 4 x 4 and transposes)
  void set_camera(const double* Ki12, const double* Ko16) {
    double KiKo[3][4];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += Ki12[i * 4 + k] * Ko16[k * 4 + j];
        KiKo[i][j] = s;
      }
    double A[3][3];
    for (int r = 0; r < 3; ++r) { A[r][0] = KiKo[r][0]; A[r][1] = KiKo[r][1]; A[r][2] = KiKo[r][3]; }
    inverse3(A, invA_);
    mapped_ = true;
  }
  // Eigen's 3 x 3 inverse (Inverse.h, compute_inverse_size3_helper): cofactors, determinant from the first column's cofactors
  static void inverse3(const double m[3][3], double out[3][3]) {
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = (c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0];
    const double invdet = 1.0 / det;
    out[0][0] = c00 * invdet; out[0][1] = c10 * invdet; out[0][2] = c20 * invdet;  // result(j, i) = cofactor(i, j) * invdet
    out[1][0] = cof(0, 1) * invdet; out[1][1] = cof(1, 1) * invdet; out[1][2] = cof(2, 1) * invdet;
    out[2][0] = cof(0, 2) * invdet; out[2][1] = cof(1, 2) * invdet; out[2][2] = cof(2, 2) * invdet;
  }
  const double* inv_a() const { return &invA_[0][0]; }
  void reset() { trk_.clear(); confirmed_.clear(); coasted_.clear(); tentative_.clear(); frame_count_ = 0; next_id_ = 0; }  // :252-259

  enum State { Tentative = 0, Confirmed = 1, Coasted = 2, Deleted = 3 };
  struct MDet {  // MappedDetection
    int ind;
    double y[2], R[4];
    float conf;
    int cls;
    float x1, y1, x2, y2, w, h;
  };
  struct Track {  // UCMCSingleTrack + UCMCKalmanFilter
    int id = 0, age = 0, death = 0, birth = 0, det_idx = -1;
    State state = Tentative;
    double x[4];
    M4 P;
  };
  std::vector<LapResult> laps;
  const std::vector<Track>& tracks() const { return trk_; }

  // mapToGroundPlane :114-128 / mapToImageSpace :130-146
  void map_det(float cx, float bottom, float w, float h, double y[2], double R[4]) const {
    if (!mapped_) {
      const double scale = 0.01;
      y[0] = cx * scale; y[1] = bottom * scale;
      const double ex = std::max(0.02, std::min(0.13, 0.0005 * w)), ey = std::max(0.02, std::min(0.10, 0.0005 * h));
      R[0] = ex * ex; R[1] = 0.0; R[2] = 0.0; R[3] = ey * ey;
      return;
    }
    const double eu = std::max(2.0, std::min(13.0, 0.05 * w)), ev = std::max(2.0, std::min(10.0, 0.05 * h));  // uvError :85-90
    const double su[4] = {eu * eu, 0.0, 0.0, ev * ev};
    // uv2xy :92-112
    const double uv1[3] = {static_cast<double>(cx), static_cast<double>(bottom), 1.0};
    double b[3];
    for (int i = 0; i < 3; ++i) b[i] = (invA_[i][0] * uv1[0] + invA_[i][1] * uv1[1]) + invA_[i][2] * uv1[2];
    const double gamma = 1.0 / b[2];
    double C[4];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) C[i * 2 + j] = gamma * invA_[i][j] - ((gamma * gamma) * b[i]) * invA_[2][j];
    y[0] = b[0] * gamma; y[1] = b[1] * gamma;
    double Cs[4];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) Cs[i * 2 + j] = C[i * 2 + 0] * su[0 * 2 + j] + C[i * 2 + 1] * su[1 * 2 + j];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) R[i * 2 + j] = Cs[i * 2 + 0] * C[j * 2 + 0] + Cs[i * 2 + 1] * C[j * 2 + 1];
  }
  // UCMCSingleTrack ctor :152-201: process noise G Q0 G^T
  static M4 process_noise(double dt, double wx, double wy) {
    const double G[4][2] = {{0.5 * dt * dt, 0.0}, {dt, 0.0}, {0.0, 0.5 * dt * dt}, {0.0, dt}};
    const double Q0[2][2] = {{wx, 0.0}, {0.0, wy}};
    double GQ[4][2];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 2; ++j) GQ[i][j] = G[i][0] * Q0[0][j] + G[i][1] * Q0[1][j];
    M4 Q;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) Q[i * 4 + j] = GQ[i][0] * G[j][0] + GQ[i][1] * G[j][1];
    return Q;
  }
  // UCMCKalmanFilter::predict :28-31 with F = I + dt at (0,1), (2,3): (F P) F^T + Q, sums in index order
  static void kf_predict(double x[4], M4& P, double dt, const M4& Q) {
    double F[4][4] = {{1, dt, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, dt}, {0, 0, 0, 1}};
    double nx[4];
    for (int i = 0; i < 4; ++i) { double s = 0.0; for (int k = 0; k < 4; ++k) s += F[i][k] * x[k]; nx[i] = s; }
    double FP[4][4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { double s = 0.0; for (int k = 0; k < 4; ++k) s += F[i][k] * P[k * 4 + j]; FP[i][j] = s; }
    M4 NP;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { double s = 0.0; for (int k = 0; k < 4; ++k) s += FP[i][k] * F[j][k]; NP[i * 4 + j] = s + Q[i * 4 + j]; }
    for (int i = 0; i < 4; ++i) x[i] = nx[i];
    P = NP;
  }
  struct Innov { double S[4], SI[4], det; };
  // S = H P H^T + R (H picks rows / columns 0 and 2), its 2 x 2 inverse as Eigen forms it (1 / det, then the adjugate scaled)
  static Innov innovation(const M4& P, const double R[4]) {
    Innov v;
    v.S[0] = P[0] + R[0]; v.S[1] = P[2] + R[1]; v.S[2] = P[8] + R[2]; v.S[3] = P[10] + R[3];
    v.det = v.S[0] * v.S[3] - v.S[2] * v.S[1];
    const double invdet = 1.0 / v.det;
    v.SI[0] = v.S[3] * invdet; v.SI[2] = -v.S[2] * invdet; v.SI[1] = -v.S[1] * invdet; v.SI[3] = v.S[0] * invdet;
    return v;
  }
  // UCMCSingleTrack::distance :213-223
  static double distance(const double x[4], const M4& P, const double y[2], const double R[4]) {
    const double d0 = y[0] - x[0], d1 = y[1] - x[2];
    const Innov v = innovation(P, R);
    const double r0 = d0 * v.SI[0] + d1 * v.SI[2], r1 = d0 * v.SI[1] + d1 * v.SI[3];  // diff^T SI
    const double maha = r0 * d0 + r1 * d1;
    return maha + std::log(v.det);
  }
  // UCMCKalmanFilter::update :33-49 (Joseph form)
  static void kf_update(double x[4], M4& P, const double z[2], const double R[4]) {
    const double y0 = z[0] - x[0], y1 = z[1] - x[2];
    const Innov v = innovation(P, R);
    double K[4][2];
    for (int i = 0; i < 4; ++i) {
      const double p0 = P[i * 4 + 0], p1 = P[i * 4 + 2];  // P H^T
      K[i][0] = p0 * v.SI[0] + p1 * v.SI[2];
      K[i][1] = p0 * v.SI[1] + p1 * v.SI[3];
    }
    for (int i = 0; i < 4; ++i) x[i] = x[i] + (K[i][0] * y0 + K[i][1] * y1);
    double A[4][4];  // I - K H
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const double kh = (j == 0) ? K[i][0] : ((j == 2) ? K[i][1] : 0.0);
        A[i][j] = ((i == j) ? 1.0 : 0.0) - kh;
      }
    double AP[4][4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { double s = 0.0; for (int k = 0; k < 4; ++k) s += A[i][k] * P[k * 4 + j]; AP[i][j] = s; }
    double KR[4][2];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 2; ++j) KR[i][j] = K[i][0] * R[0 * 2 + j] + K[i][1] * R[1 * 2 + j];
    M4 NP;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += AP[i][k] * A[j][k];
        NP[i * 4 + j] = s + (KR[i][0] * K[j][0] + KR[i][1] * K[j][1]);
      }
    P = NP;
  }

  OutTable update(const float* dets, int n) {  // :261-343
    laps.clear();
    ++frame_count_;
    std::vector<MDet> D;
    for (int i = 0; i < n; ++i) {
      const float* r = dets + static_cast<size_t>(i) * 6;
      if (r[4] < p_.det_thresh) continue;
      MDet d;
      d.ind = i; d.x1 = r[0]; d.y1 = r[1]; d.x2 = r[2]; d.y2 = r[3]; d.conf = r[4]; d.cls = static_cast<int>(r[5]);
      d.w = d.x2 - d.x1; d.h = d.y2 - d.y1;
      map_det((d.x1 + d.x2) / 2.0f, d.y2, d.w, d.h, d.y, d.R);
      D.push_back(d);
    }
    const M4 Q = process_noise(p_.dt, p_.wx, p_.wy);
    // dataAssociation :345-458
    std::vector<int> high, low;
    for (size_t i = 0; i < D.size(); ++i) (D[i].conf >= p_.high_score ? high : low).push_back(static_cast<int>(i));
    for (Track& t : trk_) { kf_predict(t.x, t.P, p_.dt, Q); ++t.age; t.det_idx = -1; }
    std::vector<int> tidx = confirmed_;
    tidx.insert(tidx.end(), coasted_.begin(), coasted_.end());
    std::vector<int> trk_remain;
    auto associate = [&](const std::vector<int>& T, const std::vector<int>& Dd, double thresh) {
      Mat cost(static_cast<int>(T.size()), static_cast<int>(Dd.size()));
      for (int i = 0; i < cost.r; ++i)
        for (int j = 0; j < cost.c; ++j) cost(i, j) = static_cast<float>(distance(trk_[T[i]].x, trk_[T[i]].P, D[Dd[j]].y, D[Dd[j]].R));
      LapResult r = linear_assignment(cost, static_cast<float>(thresh));
      laps.push_back(r);
      return r;
    };
    auto matched = [&](Track& t, const MDet& d) {
      kf_update(t.x, t.P, d.y, d.R);
      t.death = 0; t.det_idx = d.ind;
    };
    if (!high.empty() && !tidx.empty()) {
      const LapResult r = associate(tidx, high, p_.a1);
      for (const auto& m : r.matches) { Track& t = trk_[tidx[m[0]]]; matched(t, D[high[m[1]]]); t.state = Confirmed; }
      for (int i : r.unmatched_a) trk_remain.push_back(tidx[i]);
    } else trk_remain = tidx;
    if (!low.empty() && !trk_remain.empty()) {
      const LapResult r = associate(trk_remain, low, p_.a2);
      for (const auto& m : r.matches) { Track& t = trk_[trk_remain[m[0]]]; matched(t, D[low[m[1]]]); t.state = Confirmed; }
      for (int i : r.unmatched_a) trk_[trk_remain[i]].state = Coasted;
    } else {
      for (int i : trk_remain) trk_[i].state = Coasted;
    }
    // associateTentative :460-520: the high-confidence detections no track holds, in detection order
    std::vector<int> det_remain;
    for (size_t i = 0; i < D.size(); ++i) {
      bool assigned = false;
      for (const Track& t : trk_) if (t.det_idx == D[i].ind) { assigned = true; break; }
      if (!assigned && D[i].conf >= p_.high_score) det_remain.push_back(static_cast<int>(i));
    }
    if (!det_remain.empty() && !tentative_.empty()) {
      const LapResult r = associate(tentative_, det_remain, p_.a1);
      for (const auto& m : r.matches) {
        Track& t = trk_[tentative_[m[0]]];
        matched(t, D[det_remain[m[1]]]);
        if (++t.birth >= 2) { t.birth = 0; t.state = Confirmed; }
      }
      std::vector<int> rest;
      for (int j : r.unmatched_b) rest.push_back(det_remain[j]);
      det_remain.swap(rest);
    }
    // initTentative :522-535
    for (int i : det_remain) {
      Track t;
      t.id = ++next_id_;
      t.x[0] = D[i].y[0]; t.x[1] = 0.0; t.x[2] = D[i].y[1]; t.x[3] = 0.0;
      t.P.fill(0.0);
      t.P[0] = 1.0; t.P[5] = p_.vmax * p_.vmax / 3.0; t.P[10] = 1.0; t.P[15] = p_.vmax * p_.vmax / 3.0;
      t.state = Tentative; t.det_idx = D[i].ind;
      trk_.push_back(t);
    }
    // deleteOldTrackers :537-553
    std::vector<Track> keep;
    for (Track& t : trk_) {
      ++t.death;
      const bool del = (t.state == Coasted && t.death >= p_.max_age) || (t.state == Tentative && t.death >= 2);
      if (!del) keep.push_back(t);
    }
    trk_.swap(keep);
    // updateStatus :555-572
    confirmed_.clear(); coasted_.clear(); tentative_.clear();
    for (size_t i = 0; i < trk_.size(); ++i) {
      if (trk_[i].state == Confirmed) confirmed_.push_back(static_cast<int>(i));
      else if (trk_[i].state == Coasted) coasted_.push_back(static_cast<int>(i));
      else if (trk_[i].state == Tentative) tentative_.push_back(static_cast<int>(i));
    }
    // output :303-342: the confirmed tracks that hold a detection of this frame — the detection's own box
    OutTable out;
    for (const Track& t : trk_) {
      if (t.state != Confirmed || t.det_idx < 0) continue;
      for (const MDet& d : D)
        if (d.ind == t.det_idx) {
          out.push_back({d.x1, d.y1, d.x2, d.y2, static_cast<float>(t.id), d.conf, static_cast<float>(d.cls), static_cast<float>(d.ind)});
          break;
        }
    }
    return out;
  }

 private:
  Params p_;
  bool mapped_ = false;
  double invA_[3][3] = {};
  std::vector<Track> trk_;
  std::vector<int> confirmed_, coasted_, tentative_;
  int frame_count_ = 0, next_id_ = 0;
};

// =======================================================================================
// BoostTrack — src/trackers/boosttrack.cpp, motion-only configuration (with_reid = false, the constructor's default; no ECC: the
// camera-motion step needs the image). Parity unpinned (no vector in the reference's tests; Eigen's dynamic-size products cannot be
// compiled here). The filter's F, H, Q, R are sparse: predict and the projection have at most two non-zero terms per sum; the gain, the
// state and covariance updates are k-ordered chains (mul() above); S^-1 is the partial-pivot LU inverse Eigen uses for a dynamic 4 x 4.
// =======================================================================================
class BoostTrackOrc {
 public:
  static float iou_pair(const Box& a, const Box& b) {  // one entry of utils::iou_batch (iou.hpp:63-100), same operation order as orc::iou_batch
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]), a2 = (b[2] - b[0]) * (b[3] - b[1]);
    const float w = std::max(0.0f, std::min(a[2], b[2]) - std::max(a[0], b[0])), h = std::max(0.0f, std::min(a[3], b[3]) - std::max(a[1], b[1]));
    const float inter = w * h, uni = a1 + a2 - inter;
    return (uni > 0.0f) ? (inter / uni) : 0.0f;
  }
  struct Params {
    float det_thresh = 0.6f;
    int max_age = 60, min_hits = 3;
    float iou_threshold = 0.3f;
    int min_box_area = 10;
    float aspect_ratio_thresh = 1.6f, lambda_iou = 0.5f, lambda_mhd = 0.25f, lambda_shape = 0.25f;
    bool use_dlo = true, use_duo = true;
    float dlo_coef = 0.65f;
    bool use_sb = false, use_vt = false;
    bool with_reid = false;  // embeddings come with update() (no ReID model here); without them the tracker is motion-only, as in the reference (:539-551)
  };
  explicit BoostTrackOrc(const Params& p) : p_(p) {}
  void reset() { trk_.clear(); frame_count_ = 0; next_id_ = 0; }  // :272-277

  struct Track {  // BoostTrack + BoostKalmanFilter :22-135
    int id = 0, cls = 0, det_ind = -1, tsu = 0, age = 0, hit_streak = 0;
    float conf = 0.f;
    float x[8];
    SMat<8, 8> P;
    std::vector<float> emb;  // BoostTrack::emb_ (normalised when its norm is positive, :148-153)
    Box state() const {  // get_state :107-115
      const float cx = x[0], cy = x[1], h = x[2], r = x[3];
      const float w = r * h;
      return {cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2};
    }
  };
  static void to_z(const float b[4], float z[4]) {  // convert_bbox_to_z :126-134
    const float w = b[2] - b[0], h = b[3] - b[1];
    z[0] = b[0] + w / 2.0f; z[1] = b[1] + h / 2.0f; z[2] = h; z[3] = (h > 1e-6f) ? w / h : 0.0f;
  }
  static void kf_init(Track& t, const float z[4]) {  // :22-54
    for (int k = 0; k < 4; ++k) { t.x[k] = z[k]; t.x[k + 4] = 0.0f; }
    t.P = SMat<8, 8>::zero();
    for (int k = 0; k < 4; ++k) { t.P[k][k] = 10.0f; t.P[k + 4][k + 4] = 10.0f * 1000.0f; }
  }
  static void kf_predict(Track& t) {  // :56-59: F = [I I; 0 I], Q = diag(10 x 4, 0.01 x 4)
    for (int k = 0; k < 4; ++k) t.x[k] = t.x[k] + t.x[k + 4];
    SMat<8, 8> FP = t.P;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) FP[i][j] = t.P[i][j] + t.P[i + 4][j];
    SMat<8, 8> N = FP;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 4; ++j) N[i][j] = FP[i][j] + FP[i][j + 4];
    for (int k = 0; k < 4; ++k) { N[k][k] = N[k][k] + 10.0f; N[k + 4][k + 4] = N[k + 4][k + 4] + 0.01f; }
    t.P = N;
  }
  static void kf_update(Track& t, const float z[4]) {  // :61-75
    const float Rd[4] = {1.0f, 1.0f, 10.0f, 0.01f};
    SMat<4, 4> S;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) S[i][j] = t.P[i][j] + ((i == j) ? Rd[i] : 0.0f);
    const SMat<4, 4> Si = inverse_lu4(S);
    SMat<8, 4> PH;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 4; ++j) PH[i][j] = t.P[i][j];
    const SMat<8, 4> K = mul(PH, Si);
    float inn[4];
    for (int k = 0; k < 4; ++k) inn[k] = z[k] - t.x[k];
    for (int i = 0; i < 8; ++i) {
      float a = K[i][0] * inn[0];
      for (int k = 1; k < 4; ++k) a += K[i][k] * inn[k];
      t.x[i] = t.x[i] + a;
    }
    const SMat<8, 4> KS = mul(K, S);
    SMat<4, 8> Kt;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 4; ++j) Kt[j][i] = K[i][j];
    const SMat<8, 8> D = mul(KS, Kt);
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) t.P[i][j] = t.P[i][j] - D[i][j];
  }
  std::vector<LapResult> laps;
  const std::vector<Track>& tracks() const { return trk_; }

  static float norm_of(const std::vector<float>& v) { return std::sqrt(dot_chain(v.data(), v.data(), static_cast<int>(v.size()))); }  // (norm(): dot_chain's order, as elsewhere)
  static void set_emb(Track& t, const float* e, int d) {  // BoostTrack ctor :147-153
    t.emb.assign(e, e + d);
    const float nn = norm_of(t.emb);
    if (nn > 0) for (float& v : t.emb) v /= nn;
  }
  static void update_emb(Track& t, const float* e, int d, float alpha) {  // :183-199
    if (d == 0) return;
    std::vector<float> nrm(e, e + d);
    const float n0 = norm_of(nrm);
    if (n0 > 0) for (float& v : nrm) v /= n0;
    if (t.emb.empty()) { t.emb = nrm; return; }
    for (int k = 0; k < d; ++k) t.emb[k] = alpha * t.emb[k] + (1.0f - alpha) * nrm[k];
    const float n1 = norm_of(t.emb);
    if (n1 > 0) for (float& v : t.emb) v /= n1;
  }
  OutTable update(const float* dets, int n, const float* embs = nullptr, int emb_dim = 0) {  // :465-699
    laps.clear();
    ++frame_count_;
    std::vector<Det7> D = wrap_dets(dets, n);
    const bool use_emb = p_.with_reid && embs != nullptr && emb_dim > 0 && n > 0;
    for (Track& t : trk_) {  // BoostTrack::predict :156-163
      kf_predict(t);
      ++t.age;
      if (t.tsu > 0) t.hit_streak = 0;
      ++t.tsu;
    }
    // dlo_confidence_boost :361-426 (duo_confidence_boost returns its input, :428-432)
    if (p_.use_dlo && !D.empty() && !trk_.empty()) {
      std::vector<Box> tb(trk_.size());
      for (size_t j = 0; j < trk_.size(); ++j) tb[j] = trk_[j].state();
      for (Det7& d : D) {
        float max_s = 0.0f;
        bool first = true, vt = false;
        for (size_t j = 0; j < trk_.size(); ++j) {
          const float s = iou_pair(d.box(), tb[j]);
          if (first || s > max_s) { max_s = s; first = false; }  // rowwise().maxCoeff()
          const float th = std::max(0.95f - static_cast<float>(trk_[j].tsu - 1), 0.8f);
          if (s > th) vt = true;
        }
        if (!p_.use_sb && !p_.use_vt) d.conf = std::max(d.conf, max_s * p_.dlo_coef);
        else {
          if (p_.use_sb) {
            const float alpha = 0.65f;
            const float bc = alpha * d.conf + (1.0f - alpha) * std::pow(max_s, 1.5f);
            d.conf = std::max(d.conf, bc);
          }
          if (p_.use_vt && vt) d.conf = std::max(d.conf, p_.det_thresh + 1e-5f);
        }
      }
    }
    std::vector<Det7> F;
    for (const Det7& d : D) if (d.conf >= p_.det_thresh) F.push_back(d);
    const int nd = static_cast<int>(F.size()), nt = static_cast<int>(trk_.size());
    std::vector<std::array<int, 2>> matches;
    std::vector<int> ud, ut;
    if (nd > 0 && nt > 0) {
      Mat cost(nd, nt);
      const float limit = 13.2767f;
      for (int i = 0; i < nd; ++i) {
        float z[4];
        const float b[4] = {F[i].x1, F[i].y1, F[i].x2, F[i].y2};
        to_z(b, z);
        for (int j = 0; j < nt; ++j) {
          const Box tbx = trk_[j].state();
          // get_iou_matrix :297-329
          const float x1 = std::max(b[0], tbx[0]), y1 = std::max(b[1], tbx[1]), x2 = std::min(b[2], tbx[2]), y2 = std::min(b[3], tbx[3]);
          const float inter = std::max(0.0f, x2 - x1) * std::max(0.0f, y2 - y1);
          const float da = (b[2] - b[0]) * (b[3] - b[1]), ta = (tbx[2] - tbx[0]) * (tbx[3] - tbx[1]);
          const float uni = da + ta - inter;
          const float iou = (uni > 1e-6f) ? inter / uni : 0.0f;
          float c = 1.0f - iou;
          // get_mh_dist_matrix :331-359 (diagonal covariance), then the similarity (:600-611)
          float mh = 0.0f;
          for (int k = 0; k < 4; ++k) {
            const float df = z[k] - trk_[j].x[k];
            const float term = df * df * (1.0f / trk_[j].P[k][k]);
            mh = (k == 0) ? term : mh + term;
          }
          if (mh > limit) mh = limit;
          const float sim = (limit - mh) / limit;
          c = c - p_.lambda_mhd * sim;
          if (use_emb) {  // :581-594, :613-618: raw detection embedding x the track's stored one (a zero row when it has none of that width)
            const float* de = embs + static_cast<size_t>(F[i].ind) * emb_dim;
            const float dp = (static_cast<int>(trk_[j].emb.size()) == emb_dim) ? dot_chain(de, trk_[j].emb.data(), emb_dim) : 0.0f;
            const float lambda_emb = (1.0f + p_.lambda_iou + p_.lambda_shape + p_.lambda_mhd) * 1.5f;
            c = c - lambda_emb * ((dp + 1.0f) / 2.0f);
          }
          cost(i, j) = c;
        }
      }
      const LapResult r = linear_assignment(cost, p_.iou_threshold);
      laps.push_back(r);
      matches = r.matches; ud = r.unmatched_a; ut = r.unmatched_b;
    } else if (nd > 0) {
      for (int i = 0; i < nd; ++i) ud.push_back(i);
    }
    for (const auto& m : matches) {  // BoostTrack::update :165-181
      Track& t = trk_[m[1]];
      const Det7& d = F[m[0]];
      t.tsu = 0; ++t.hit_streak;
      const float b[4] = {d.x1, d.y1, d.x2, d.y2};
      float z[4];
      to_z(b, z);
      kf_update(t, z);
      t.conf = d.conf; t.cls = static_cast<int>(d.cls); t.det_ind = d.ind;
      if (use_emb) {  // :637-650: dets_alpha = af + (1 - af) * (1 - trust), trust = (score - det_thresh) / (1 - det_thresh)
        const float trust = (d.conf - p_.det_thresh) / (1.0f - p_.det_thresh);
        const float af = 0.95f;
        update_emb(t, embs + static_cast<size_t>(d.ind) * emb_dim, emb_dim, af + (1.0f - af) * (1.0f - trust));
      }
    }
    for (int i : ud) {  // :652-661
      const Det7& d = F[i];
      Track t;
      const float b[4] = {d.x1, d.y1, d.x2, d.y2};
      float z[4];
      to_z(b, z);
      kf_init(t, z);
      t.id = ++next_id_; t.conf = d.conf; t.cls = static_cast<int>(d.cls); t.det_ind = d.ind;
      if (use_emb) set_emb(t, embs + static_cast<size_t>(d.ind) * emb_dim, emb_dim);
      trk_.push_back(t);
    }
    OutTable out;
    for (const Track& t : trk_) {
      if (t.tsu < 1 && (t.hit_streak >= p_.min_hits || frame_count_ <= p_.min_hits)) {
        const Box b = t.state();
        // filter_outputs :434-463
        const float w = b[2] - b[0], h = b[3] - b[1];
        const float area = w * h, ar = w / (h + 1e-6f);
        if (ar <= p_.aspect_ratio_thresh && area > static_cast<float>(p_.min_box_area))
          out.push_back({b[0], b[1], b[2], b[3], static_cast<float>(t.id), t.conf, static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
      }
    }
    std::vector<Track> keep;
    for (const Track& t : trk_) if (!(t.tsu > p_.max_age)) keep.push_back(t);
    trk_.swap(keep);
    return out;
  }
  std::vector<std::vector<float>> dump_features() const {  // the stored embeddings, list order (a track without one: zeros of the widest row)
    size_t d = 0;
    for (const Track& t : trk_) d = std::max(d, t.emb.size());
    std::vector<std::vector<float>> rows;
    for (const Track& t : trk_) { std::vector<float> r(d, 0.0f); std::copy(t.emb.begin(), t.emb.end(), r.begin()); rows.push_back(r); }
    return rows;
  }
  std::vector<std::vector<float>> dump_states() const {  // [id, x(8), P(64)]
    std::vector<std::vector<float>> rows;
    for (const Track& t : trk_) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id));
      for (int k = 0; k < 8; ++k) r.push_back(t.x[k]);
      for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) r.push_back(t.P[i][j]);
      rows.push_back(r);
    }
    return rows;
  }

 private:
  Params p_;
  std::vector<Track> trk_;
  int frame_count_ = 0, next_id_ = 0;
};

// =======================================================================================
// HybridSORT — src/trackers/hybridsort.cpp, as written: the association the reference runs is its "simplified" one (the four corner
// velocities and the k-th previous observations are computed and then ignored, :645-716), so a frame is: score split, predict,
// 1 - HMIoU assignment against each track's LAST OBSERVED box (get_bbox :364-369; the Kalman box only for a track that has none),
// BYTE assignment on IoU minus the score difference, a last assignment against the last observed boxes, a Kalman update with an
// ALL-ZERO measurement for every track still unmatched (:1181-1188 -> :315-320), births, output in reverse track order.
// Built: with_reid = false, and with_reid = true WITHOUT embeddings (the reference then uses all-zero features, :868-871: every
// appearance distance is 1, the first association's costs carry + EG_weight_high_score and the BYTE step's + EG_weight_low_score,
// which puts every BYTE cost above its threshold). Parity unpinned. 9-state filter [u, v, s, c, r, du, dv, ds, dc]: predict has
// two-term sums; gain / state / covariance updates are k-ordered chains; S^-1 = partial-pivot LU inverse of the 5 x 5.
// =======================================================================================
template <int N>
inline SMat<N, N> inverse_lu(const SMat<N, N>& S) {  // Eigen's dynamic-size inverse() = partialPivLu().inverse(), as inverse_lu4 above
  SMat<N, N> lu = S;
  int perm[N];
  for (int i = 0; i < N; ++i) perm[i] = i;
  for (int k = 0; k < N; ++k) {
    int p = k;
    float best = std::fabs(lu[k][k]);
    for (int i = k + 1; i < N; ++i) { const float v = std::fabs(lu[i][k]); if (v > best) { best = v; p = i; } }
    if (p != k) { for (int j = 0; j < N; ++j) std::swap(lu[k][j], lu[p][j]); std::swap(perm[k], perm[p]); }
    for (int i = k + 1; i < N; ++i) lu[i][k] /= lu[k][k];
    for (int i = k + 1; i < N; ++i)
      for (int j = k + 1; j < N; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
  SMat<N, N> inv;
  for (int c = 0; c < N; ++c) {
    float b[N];
    for (int i = 0; i < N; ++i) b[i] = (perm[i] == c) ? 1.0f : 0.0f;
    for (int i = 0; i < N; ++i)
      for (int r = i + 1; r < N; ++r) b[r] -= b[i] * lu[r][i];
    for (int i = N - 1; i >= 0; --i) {
      b[i] /= lu[i][i];
      for (int r = 0; r < i; ++r) b[r] -= b[i] * lu[r][i];
    }
    for (int i = 0; i < N; ++i) inv[i][c] = b[i];
  }
  return inv;
}

class HybridSortOrc {
 public:
  struct Params {
    float det_thresh = 0.7f;
    int max_age = 30, min_hits = 3;
    float iou_threshold = 0.15f;
    int asso = 1;  // 0: IoU (also what giou / ciou / diou are here, :579-592), 1: hmiou
    float low_thresh = 0.1f;
    bool use_byte = true;
    float track_thresh = 0.5f, eg_high = 4.6f, eg_low = 1.3f;
    bool tcm_first = true, tcm_byte = true;
    float tcm_byte_weight = 1.0f;
    bool with_reid = false;
  };
  explicit HybridSortOrc(const Params& p) : p_(p) {}
  void reset() { trk_.clear(); frame_count_ = 0; next_id_ = 0; }  // :478-482 (the track list is kept by the reference; see note in update)

  struct Track {
    int id = 0, age = 0, hits = 0, hit_streak = 0, tsu = 0, cls = 0, det_ind = -1;
    float conf = 0.f, conf_pre = 0.f;
    bool has_obs = false;
    float last[4] = {-1.f, -1.f, -1.f, -1.f};
    float x[9];
    SMat<9, 9> P;
  };
  static void to_z(const float b[4], float c, float z[5]) {  // convert_bbox_to_z :181-193
    const float w = b[2] - b[0], h = b[3] - b[1];
    z[0] = b[0] + w / 2.0f; z[1] = b[1] + h / 2.0f; z[2] = w * h; z[3] = c; z[4] = (h > 1e-6f) ? w / h : 0.0f;
  }
  static void x_to_box(const float* x, float b[4]) {  // convert_x_to_bbox :195-201
    const float u = x[0], v = x[1], s = x[2], r = x[4];
    const float w = std::sqrt(s * r);
    const float h = s / w;
    b[0] = u - w / 2; b[1] = v - h / 2; b[2] = u + w / 2; b[3] = v + h / 2;
  }
  static void kf_init(Track& t, const float z[5]) {  // :26-65
    for (int k = 0; k < 5; ++k) t.x[k] = z[k];
    for (int k = 5; k < 9; ++k) t.x[k] = 0.0f;
    t.P = SMat<9, 9>::zero();
    for (int k = 0; k < 5; ++k) t.P[k][k] = 10.0f;
    for (int k = 5; k < 9; ++k) t.P[k][k] = 10.0f * 1000.0f;
  }
  static void kf_predict(Track& t) {  // :67-70: F = I + 1 at (0,5), (1,6), (2,7), (3,8); Q = diag(0.1 x 5, 0.01 x 4)
    for (int k = 0; k < 4; ++k) t.x[k] = t.x[k] + t.x[k + 5];
    SMat<9, 9> FP = t.P;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 9; ++j) FP[i][j] = t.P[i][j] + t.P[i + 5][j];
    SMat<9, 9> N = FP;
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 4; ++j) N[i][j] = FP[i][j] + FP[i][j + 5];
    for (int k = 0; k < 5; ++k) N[k][k] = N[k][k] + 0.1f;
    for (int k = 5; k < 9; ++k) N[k][k] = N[k][k] + 0.01f;
    t.P = N;
  }
  static void kf_update(Track& t, const float z[5]) {  // :72-88
    const float Rd[5] = {1.0f, 1.0f, 10.0f, 0.01f, 1.0f};
    SMat<5, 5> S;
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) S[i][j] = t.P[i][j] + ((i == j) ? Rd[i] : 0.0f);
    const SMat<5, 5> Si = inverse_lu<5>(S);
    SMat<9, 5> PH;
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 5; ++j) PH[i][j] = t.P[i][j];
    const SMat<9, 5> K = mul(PH, Si);
    float inn[5];
    for (int k = 0; k < 5; ++k) inn[k] = z[k] - t.x[k];
    for (int i = 0; i < 9; ++i) {
      float a = K[i][0] * inn[0];
      for (int k = 1; k < 5; ++k) a += K[i][k] * inn[k];
      t.x[i] = t.x[i] + a;
    }
    SMat<9, 9> A;  // I - K H
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) A[i][j] = ((i == j) ? 1.0f : 0.0f) - ((j < 5) ? K[i][j] : 0.0f);
    t.P = mul(A, t.P);
  }
  static float iou_manual(const float a[4], const float b[4]) {  // HybridSort::iou_batch :529-556
    const float xx1 = std::max(a[0], b[0]), yy1 = std::max(a[1], b[1]), xx2 = std::min(a[2], b[2]), yy2 = std::min(a[3], b[3]);
    const float w = std::max(0.0f, xx2 - xx1), h = std::max(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]), a2 = (b[2] - b[0]) * (b[3] - b[1]);
    const float uni = a1 + a2 - inter;
    return (uni > 1e-6f) ? inter / uni : 0.0f;
  }
  static float hmiou(const float a[4], const float b[4]) {  // :558-577
    float v = iou_manual(a, b);
    const float yy1 = std::max(a[1], b[1]), yy2 = std::min(a[3], b[3]), yy3 = std::min(a[1], b[1]), yy4 = std::max(a[3], b[3]);
    const float ho = std::max(0.0f, yy2 - yy1) / (yy4 - yy3 + 1e-6f);
    v *= ho;
    return v;
  }
  std::vector<LapResult> laps;
  const std::vector<Track>& tracks() const { return trk_; }

  void predict(Track& t) {  // HybridKalmanBoxTracker::predict :256-270
    if (t.x[7] + t.x[2] <= 0) t.x[7] = 0.0f;
    kf_predict(t);
    ++t.age;
    if (t.tsu > 0) t.hit_streak = 0;
    ++t.tsu;
  }
  void get_bbox(const Track& t, float b[4]) const {  // :364-369
    if (((t.last[0] + t.last[1]) + t.last[2]) + t.last[3] < 0) { x_to_box(t.x, b); return; }
    for (int k = 0; k < 4; ++k) b[k] = t.last[k];
  }
  float simple_score(const Track& t) const {  // :376-381
    auto clampf = [](float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); };
    if (t.conf_pre == 0.0f) return clampf(t.conf, 0.1f, p_.track_thresh);
    return clampf(t.conf - (t.conf_pre - t.conf), 0.1f, p_.track_thresh);
  }
  void matched(Track& t, const Det7& d) {  // HybridKalmanBoxTracker::update with a box :272-313
    t.last[0] = d.x1; t.last[1] = d.y1; t.last[2] = d.x2; t.last[3] = d.y2;
    t.has_obs = true;
    t.tsu = 0; ++t.hits; ++t.hit_streak;
    float z[5];
    to_z(t.last, d.conf, z);
    kf_update(t, z);
    t.cls = static_cast<int>(d.cls); t.det_ind = d.ind;
    t.conf_pre = t.conf; t.conf = d.conf;
  }

  OutTable update(const float* dets, int n) {  // :825-1262
    laps.clear();
    ++frame_count_;
    if (n == 0) {
      for (Track& t : trk_) predict(t);
      drop_dead();
      return {};
    }
    const std::vector<Det7> D = wrap_dets(dets, n);
    std::vector<Det7> keep, second;
    for (const Det7& d : D) {
      if (d.conf > p_.low_thresh && d.conf < p_.det_thresh) second.push_back(d);
      if (d.conf > p_.det_thresh) keep.push_back(d);
    }
    const int nt = static_cast<int>(trk_.size());
    std::vector<std::array<float, 4>> tb(nt);
    std::vector<float> tscore(nt);
    for (int j = 0; j < nt; ++j) {
      predict(trk_[j]);
      get_bbox(trk_[j], tb[j].data());
      tscore[j] = simple_score(trk_[j]);
    }
    std::vector<std::array<float, 4>> last(nt);
    for (int j = 0; j < nt; ++j) for (int k = 0; k < 4; ++k) last[j][k] = trk_[j].last[k];
    std::vector<int> ud, ut;
    std::vector<std::array<int, 2>> m1;
    const float thr = p_.iou_threshold;
    if (p_.tcm_first && !keep.empty() && nt > 0) {  // associate_4_points_with_score(_with_reid) :645-823
      const int nd = static_cast<int>(keep.size());
      Mat sim(nd, nt), cost(nd, nt);
      const bool zero_reid = p_.with_reid && p_.eg_high > 0;
      for (int i = 0; i < nd; ++i) {
        const float a[4] = {keep[i].x1, keep[i].y1, keep[i].x2, keep[i].y2};
        for (int j = 0; j < nt; ++j) {
          sim(i, j) = (p_.asso == 1) ? hmiou(a, tb[j].data()) : iou_manual(a, tb[j].data());
          float c = 1.0f - sim(i, j);
          if (zero_reid) { c = c * 1.0f; c += 1.0f * p_.eg_high; }  // (ones - iou) * weights.first; += emb_cost * weights.second, emb_cost = 1
          cost(i, j) = c;
        }
      }
      const float max_cost = zero_reid ? (1.0f - thr) * 1.0f + p_.eg_high : 1.0f - thr;
      const LapResult r = linear_assignment(cost, max_cost);
      laps.push_back(r);
      std::vector<char> dm(nd, 0), tm(nt, 0);
      for (const auto& m : r.matches) {
        if (sim(m[0], m[1]) >= thr) { m1.push_back(m); dm[m[0]] = 1; tm[m[1]] = 1; }
        else { ud.push_back(m[0]); ut.push_back(m[1]); }
      }
      for (int i = 0; i < nd; ++i) if (!dm[i]) ud.push_back(i);
      for (int j = 0; j < nt; ++j) if (!tm[j]) ut.push_back(j);
    } else {
      for (size_t i = 0; i < keep.size(); ++i) ud.push_back(static_cast<int>(i));
      for (int j = 0; j < nt; ++j) ut.push_back(j);
    }
    for (const auto& m : m1) matched(trk_[m[1]], keep[m[0]]);
    // BYTE :1052-1128
    if (p_.use_byte && !second.empty() && !ut.empty()) {
      const int ns = static_cast<int>(second.size()), nu = static_cast<int>(ut.size());
      Mat il(ns, nu);
      float mx = 0.0f;
      for (int i = 0; i < ns; ++i) {
        const float a[4] = {second[i].x1, second[i].y1, second[i].x2, second[i].y2};
        for (int j = 0; j < nu; ++j) {
          float v = iou_manual(a, tb[ut[j]].data());
          if (p_.tcm_byte) v -= std::fabs(tscore[ut[j]] - second[i].conf) * p_.tcm_byte_weight;
          il(i, j) = v;
          if ((i == 0 && j == 0) || v > mx) mx = v;
        }
      }
      if (mx > thr) {
        Mat cost(ns, nu);
        const bool zero_reid = p_.with_reid && p_.eg_low > 0;
        for (int i = 0; i < ns; ++i)
          for (int j = 0; j < nu; ++j) {
            float c = 1.0f - il(i, j);
            if (zero_reid) c += 1.0f * p_.eg_low;
            cost(i, j) = c;
          }
        const LapResult r = linear_assignment(cost, 1.0f - thr);
        laps.push_back(r);
        std::vector<char> gone(nt, 0);
        for (const auto& m : r.matches)
          if (il(m[0], m[1]) >= thr) { matched(trk_[ut[m[1]]], second[m[0]]); gone[ut[m[1]]] = 1; }
        std::vector<int> rest;
        for (int j : ut) if (!gone[j]) rest.push_back(j);
        ut.swap(rest);
      }
    }
    // the last chance: unmatched detections against the LAST OBSERVED boxes of the unmatched tracks :1130-1179
    if (!ud.empty() && !ut.empty()) {
      const int nd = static_cast<int>(ud.size()), nu = static_cast<int>(ut.size());
      Mat il(nd, nu);
      float mx = 0.0f;
      for (int i = 0; i < nd; ++i) {
        const float a[4] = {keep[ud[i]].x1, keep[ud[i]].y1, keep[ud[i]].x2, keep[ud[i]].y2};
        for (int j = 0; j < nu; ++j) {
          il(i, j) = iou_manual(a, last[ut[j]].data());
          if ((i == 0 && j == 0) || il(i, j) > mx) mx = il(i, j);
        }
      }
      if (mx > thr) {
        Mat cost(nd, nu);
        for (int i = 0; i < nd; ++i)
          for (int j = 0; j < nu; ++j) cost(i, j) = 1.0f - il(i, j);
        const LapResult r = linear_assignment(cost, 1.0f - thr);
        laps.push_back(r);
        std::vector<char> dgone(keep.size(), 0), tgone(nt, 0);
        for (const auto& m : r.matches)
          if (il(m[0], m[1]) >= thr) { matched(trk_[ut[m[1]]], keep[ud[m[0]]]); dgone[ud[m[0]]] = 1; tgone[ut[m[1]]] = 1; }
        std::vector<int> rd, rt;
        for (int i : ud) if (!dgone[i]) rd.push_back(i);
        for (int j : ut) if (!tgone[j]) rt.push_back(j);
        ud.swap(rd); ut.swap(rt);
      }
    }
    for (int j : ut) {  // update(empty box) :315-320: a Kalman update with an all-zero measurement
      const float z[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      kf_update(trk_[j], z);
      trk_[j].conf_pre = 0.0f;
    }
    for (int i : ud) {  // :1190-1210
      Track t;
      const float b[4] = {keep[i].x1, keep[i].y1, keep[i].x2, keep[i].y2};
      float z[5];
      to_z(b, keep[i].conf, z);
      kf_init(t, z);
      t.id = ++next_id_;  // next_id() :21-23; the table shows id + 1 (:1225): the first track of a tracker is reported as 2
      t.conf = keep[i].conf; t.cls = static_cast<int>(keep[i].cls); t.det_ind = keep[i].ind;
      trk_.push_back(t);
    }
    OutTable out;
    for (auto it = trk_.rbegin(); it != trk_.rend(); ++it) {
      const Track& t = *it;
      if (t.tsu < 1 && (t.hit_streak >= p_.min_hits || frame_count_ <= p_.min_hits)) {
        float b[4];
        get_bbox(t, b);
        out.push_back({b[0], b[1], b[2], b[3], static_cast<float>(t.id + 1), t.conf, static_cast<float>(t.cls), static_cast<float>(t.det_ind)});
      }
    }
    drop_dead();
    return out;
  }
  std::vector<std::vector<float>> dump_states() const {  // [id + 1, x(9), P(81)]
    std::vector<std::vector<float>> rows;
    for (const Track& t : trk_) {
      std::vector<float> r;
      r.push_back(static_cast<float>(t.id + 1));
      for (int k = 0; k < 9; ++k) r.push_back(t.x[k]);
      for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) r.push_back(t.P[i][j]);
      rows.push_back(r);
    }
    return rows;
  }

 private:
  void drop_dead() {
    std::vector<Track> keep;
    for (const Track& t : trk_) if (!(t.tsu > p_.max_age)) keep.push_back(t);
    trk_.swap(keep);
  }
  Params p_;
  std::vector<Track> trk_;
  int frame_count_ = 0, next_id_ = 0;
};

}  // namespace orc
