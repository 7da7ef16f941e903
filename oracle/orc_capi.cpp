// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
// Flat C entry points over the CPU restatement so tests/ and bench.py's cpu_baseline leg can
// drive it through ctypes. Nothing in the product links or loads this library.
#include <cstring>
#include <limits>
#include <memory>
#include <variant>

#include "orc_trackers.hpp"

using namespace orc;

namespace {
struct Handle {
  int kind;
  std::unique_ptr<Sort> sort;
  std::unique_ptr<ByteTrack> byte;
  std::unique_ptr<OCSort> oc;
  std::unique_ptr<BotSort> bot;
  std::unique_ptr<DeepOCSort> deep;
  std::unique_ptr<StrongSort> strong;
  std::unique_ptr<Ucmc> ucmc;
  std::unique_ptr<BoostTrackOrc> boost;
  std::unique_ptr<HybridSortOrc> hybrid;
  const std::vector<LapResult>* laps() const {
    switch (kind) {
      case 1: return &byte->laps;
      case 2: return &oc->laps;
      case 3: return &bot->laps;
      case 4: return &deep->laps;
      case 5: return &strong->laps;
      case 6: return &ucmc->laps;
      case 7: return &boost->laps;
      case 8: return &hybrid->laps;
      default: return nullptr;
    }
  }
  std::vector<LapResult> sort_laps;
};
float P(const float* p, int n, int i, float dflt) { return (p && i < n) ? p[i] : dflt; }
Mat as_mat(const float* a, int r, int c) {
  Mat m(r, c);
  if (r * c > 0) std::memcpy(m.a.data(), a, sizeof(float) * static_cast<size_t>(r) * c);
  return m;
}
}  // namespace

extern "C" {

// kind: 0 SORT, 1 ByteTrack, 2 OC-SORT, 3 BoT-SORT. Parameter vectors (missing tail = defaults):
//  SORT     [det_thresh, max_age, max_obs, min_hits, iou_threshold]
//  ByteTrack[min_conf, track_thresh, match_thresh, track_buffer, frame_rate, max_age, max_obs]
//  OC-SORT  [det_thresh, max_age, max_obs, min_hits, iou_threshold, min_conf, delta_t, inertia, use_byte, Q_xy, Q_s]
//  BoT-SORT [track_high, track_low, new_track, track_buffer, match_thresh, proximity, appearance,
//            frame_rate, fuse_first_associate, with_reid, max_age, max_obs]
void* orc_tracker_create(int kind, const float* p, int np) {
  auto* h = new Handle();
  h->kind = kind;
  switch (kind) {
    case 0:
      h->sort = std::make_unique<Sort>(P(p, np, 0, 0.3f), (int)P(p, np, 1, 1), (int)P(p, np, 2, 50),
                                       (int)P(p, np, 3, 3), P(p, np, 4, 0.3f));
      break;
    case 1:
      h->byte = std::make_unique<ByteTrack>(P(p, np, 0, 0.1f), P(p, np, 1, 0.45f), P(p, np, 2, 0.8f),
                                            (int)P(p, np, 3, 25), (int)P(p, np, 4, 30),
                                            (int)P(p, np, 5, 30), (int)P(p, np, 6, 50));
      break;
    case 2:
      h->oc = std::make_unique<OCSort>(P(p, np, 0, 0.2f), (int)P(p, np, 1, 30), (int)P(p, np, 2, 50),
                                       (int)P(p, np, 3, 3), P(p, np, 4, 0.3f), P(p, np, 5, 0.1f),
                                       (int)P(p, np, 6, 3), P(p, np, 7, 0.2f), P(p, np, 8, 0.f) != 0.f,
                                       P(p, np, 9, 0.01f), P(p, np, 10, 0.0001f));
      h->oc->set_asso((int)P(p, np, 11, 0.f), (int)P(p, np, 12, 1920.f), (int)P(p, np, 13, 1080.f));
      break;
    case 3:
      h->bot = std::make_unique<BotSort>(P(p, np, 0, 0.5f), P(p, np, 1, 0.1f), P(p, np, 2, 0.6f),
                                         (int)P(p, np, 3, 30), P(p, np, 4, 0.8f), P(p, np, 5, 0.5f),
                                         P(p, np, 6, 0.25f), (int)P(p, np, 7, 30), P(p, np, 8, 0.f) != 0.f,
                                         P(p, np, 9, 1.f) != 0.f, (int)P(p, np, 10, 30), (int)P(p, np, 11, 50));
      break;
    case 4:  // det_thresh, max_age, max_obs, min_hits, iou_thr, delta_t, inertia, w_emb, alpha_fixed, aw_param, emb_off, cmc_off, aw_off, q_xy, q_s, asso, w, h
      h->deep = std::make_unique<DeepOCSort>(P(p, np, 0, 0.3f), (int)P(p, np, 1, 30), (int)P(p, np, 2, 50), (int)P(p, np, 3, 3),
                                             P(p, np, 4, 0.3f), (int)P(p, np, 5, 3), P(p, np, 6, 0.2f), P(p, np, 7, 0.5f),
                                             P(p, np, 8, 0.95f), P(p, np, 9, 0.5f), P(p, np, 10, 0.f) != 0.f, P(p, np, 11, 0.f) != 0.f,
                                             P(p, np, 12, 0.f) != 0.f, P(p, np, 13, 0.01f), P(p, np, 14, 0.0001f));
      h->deep->set_asso((int)P(p, np, 15, 0.f), (int)P(p, np, 16, 1920.f), (int)P(p, np, 17, 1080.f));
      break;
    case 5:  // min_conf, max_cos_dist, max_iou_dist, n_init, nn_budget, mc_lambda, ema_alpha, max_age
      h->strong = std::make_unique<StrongSort>(P(p, np, 0, 0.1f), P(p, np, 1, 0.2f), P(p, np, 2, 0.7f), (int)P(p, np, 3, 3), (int)P(p, np, 4, 100),
                                               P(p, np, 5, 0.98f), P(p, np, 6, 0.9f), (int)P(p, np, 7, 30));
      break;
    case 7: {  // BoostTrack, motion only: det_thresh, max_age, min_hits, iou_threshold, min_box_area, aspect_ratio_thresh, lambda_iou, lambda_mhd,
               // lambda_shape, use_dlo_boost, use_duo_boost, dlo_boost_coef, use_sb, use_vt
      BoostTrackOrc::Params q;
      q.det_thresh = P(p, np, 0, 0.6f); q.max_age = (int)P(p, np, 1, 60); q.min_hits = (int)P(p, np, 2, 3); q.iou_threshold = P(p, np, 3, 0.3f);
      q.min_box_area = (int)P(p, np, 4, 10); q.aspect_ratio_thresh = P(p, np, 5, 1.6f); q.lambda_iou = P(p, np, 6, 0.5f); q.lambda_mhd = P(p, np, 7, 0.25f);
      q.lambda_shape = P(p, np, 8, 0.25f); q.use_dlo = P(p, np, 9, 1.f) != 0.f; q.use_duo = P(p, np, 10, 1.f) != 0.f; q.dlo_coef = P(p, np, 11, 0.65f);
      q.use_sb = P(p, np, 12, 0.f) != 0.f; q.use_vt = P(p, np, 13, 0.f) != 0.f; q.with_reid = P(p, np, 14, 0.f) != 0.f;
      h->boost = std::make_unique<BoostTrackOrc>(q);
      break;
    }
    case 8: {  // HybridSORT: det_thresh, max_age, min_hits, iou_threshold, asso (0 iou, 1 hmiou), low_thresh, use_byte, track_thresh,
               // EG_weight_high_score, EG_weight_low_score, TCM_first_step, TCM_byte_step, TCM_byte_step_weight, with_reid (no embeddings)
      HybridSortOrc::Params q;
      q.det_thresh = P(p, np, 0, 0.7f); q.max_age = (int)P(p, np, 1, 30); q.min_hits = (int)P(p, np, 2, 3); q.iou_threshold = P(p, np, 3, 0.15f);
      q.asso = (int)P(p, np, 4, 1.f); q.low_thresh = P(p, np, 5, 0.1f); q.use_byte = P(p, np, 6, 1.f) != 0.f; q.track_thresh = P(p, np, 7, 0.5f);
      q.eg_high = P(p, np, 8, 4.6f); q.eg_low = P(p, np, 9, 1.3f); q.tcm_first = P(p, np, 10, 1.f) != 0.f; q.tcm_byte = P(p, np, 11, 1.f) != 0.f;
      q.tcm_byte_weight = P(p, np, 12, 1.0f); q.with_reid = P(p, np, 13, 0.f) != 0.f;
      h->hybrid = std::make_unique<HybridSortOrc>(q);
      break;
    }
    default:
      delete h;
      return nullptr;
  }
  return h;
}
// UCMCTrack (kind 6): p = [det_thresh, max_age, a1, a2, wx, wy, vmax, dt, high_score] in double precision (dt = 1.0 / fps is not a
// float), Ki (3 x 4) / Ko (4 x 4) row-major or null (image-space fallback)
void* orc_ucmc_create(const double* p, int np, const double* Ki12, const double* Ko16) {
  auto D = [&](int i, double dflt) { return (p && i < np) ? p[i] : dflt; };
  Ucmc::Params q;
  q.det_thresh = static_cast<float>(D(0, 0.3)); q.max_age = static_cast<int>(D(1, 30)); q.a1 = D(2, 100.0); q.a2 = D(3, 100.0);
  q.wx = D(4, 5.0); q.wy = D(5, 5.0); q.vmax = D(6, 10.0); q.dt = D(7, 1.0 / 30.0); q.high_score = static_cast<float>(D(8, 0.5));
  auto* h = new Handle();
  h->kind = 6;
  h->ucmc = std::make_unique<Ucmc>(q);
  if (Ki12 && Ko16) h->ucmc->set_camera(Ki12, Ko16);
  return h;
}
// rows of [id, state, death, birth, det_idx, age, x(4), P(16)] in list order; returns the number of tracks (or -needed)
int orc_ucmc_dump(void* hv, double* out, int cap_rows) {
  auto* h = static_cast<Handle*>(hv);
  if (!h->ucmc) return 0;
  const auto& T = h->ucmc->tracks();
  if (static_cast<int>(T.size()) > cap_rows) return -static_cast<int>(T.size());
  for (size_t i = 0; i < T.size(); ++i) {
    double* r = out + i * 26;
    r[0] = T[i].id; r[1] = T[i].state; r[2] = T[i].death; r[3] = T[i].birth; r[4] = T[i].det_idx; r[5] = T[i].age;
    for (int k = 0; k < 4; ++k) r[6 + k] = T[i].x[k];
    for (int k = 0; k < 16; ++k) r[10 + k] = T[i].P[k];
  }
  return static_cast<int>(T.size());
}
// the primitives on their own: distance of n tracks to m mapped detections (row-major n x m floats, as the tracker casts them), the mapping
void orc_ucmc_distance(int n, const double* x, const double* P, int m, const double* y, const double* R, float* out) {
  for (int i = 0; i < n; ++i) {
    Ucmc::M4 Pi;
    for (int k = 0; k < 16; ++k) Pi[k] = P[static_cast<size_t>(i) * 16 + k];
    for (int j = 0; j < m; ++j) out[static_cast<size_t>(i) * m + j] = static_cast<float>(Ucmc::distance(x + static_cast<size_t>(i) * 4, Pi, y + static_cast<size_t>(j) * 2, R + static_cast<size_t>(j) * 4));
  }
}
void orc_tracker_destroy(void* hv) { delete static_cast<Handle*>(hv); }
void orc_tracker_reset(void* hv) {
  auto* h = static_cast<Handle*>(hv);
  if (h->sort) h->sort->reset();
  if (h->byte) h->byte->reset();
  if (h->oc) h->oc->reset();
  if (h->bot) h->bot->reset();
  if (h->deep) h->deep->reset();
  if (h->strong) h->strong->reset();
  if (h->ucmc) h->ucmc->reset();
  if (h->boost) h->boost->reset();
  if (h->hybrid) h->hybrid->reset();
}

// BoT-SORT only: the 2x3 camera-motion warp of the next update() (returns 0, or -1 for the other trackers)
int orc_tracker_set_warp(void* hv, const float* w2x3) {
  auto* h = static_cast<Handle*>(hv);
  if (h->deep) { h->deep->set_warp(w2x3); return 0; }
  if (!h->bot) return -1;
  h->bot->set_warp(w2x3);
  return 0;
}

// dets: row-major n x 6. embs: row-major n x d or null. out: row-major cap x 8. Returns rows (or -needed).
int orc_tracker_update(void* hv, const float* dets, int n, const float* embs, int d, float* out, int cap) {
  auto* h = static_cast<Handle*>(hv);
  OutTable t;
  switch (h->kind) {
    case 0:
      t = h->sort->update(dets, n);
      h->sort_laps.clear();
      if (!h->sort->last_lap.x.empty() || !h->sort->last_lap.y.empty()) h->sort_laps.push_back(h->sort->last_lap);
      h->sort->last_lap = LapResult();
      break;
    case 1: t = h->byte->update(dets, n); break;
    case 2: t = h->oc->update(dets, n); break;
    case 3: t = h->bot->update(dets, n, embs, d); break;
    case 4: t = h->deep->update(dets, n, embs, d); break;
    case 5: t = h->strong->update(dets, n, embs, d); break;
    case 6: t = h->ucmc->update(dets, n); break;
    case 7: t = h->boost->update(dets, n, embs, d); break;
    case 8: t = h->hybrid->update(dets, n); break;
  }
  const int rows = static_cast<int>(t.size());
  if (rows > cap) return -rows;
  for (int i = 0; i < rows; ++i) std::memcpy(out + static_cast<size_t>(i) * 8, t[i].data(), 8 * sizeof(float));
  return rows;
}
int orc_tracker_lap_count(void* hv) {
  auto* h = static_cast<Handle*>(hv);
  if (h->kind == 0) return static_cast<int>(h->sort_laps.size());
  return static_cast<int>(h->laps()->size());
}
// copies x (n) and y (m) of the k-th assignment solved during the last update
int orc_tracker_lap_get(void* hv, int k, int* n, int* m, int* x, int* y, int cap) {
  auto* h = static_cast<Handle*>(hv);
  const std::vector<LapResult>& v = (h->kind == 0) ? h->sort_laps : *h->laps();
  if (k < 0 || k >= static_cast<int>(v.size())) return -1;
  *n = static_cast<int>(v[k].x.size());
  *m = static_cast<int>(v[k].y.size());
  if (*n > cap || *m > cap) return -2;
  std::memcpy(x, v[k].x.data(), sizeof(int) * *n);
  std::memcpy(y, v[k].y.data(), sizeof(int) * *m);
  return 0;
}
// state dump: rows of [id, mean(d), cov(d*d)] — returns number of rows, row width in *w
int orc_tracker_dump_states(void* hv, float* out, int cap_floats, int* w) {
  auto* h = static_cast<Handle*>(hv);
  std::vector<std::vector<float>> s;
  switch (h->kind) {
    case 0: s = h->sort->dump_states(); break;
    case 1: s = h->byte->dump_states(); break;
    case 2: s = h->oc->dump_states(); break;
    case 3: s = h->bot->dump_states(); break;
    case 4: s = h->deep->dump_states(); break;
    case 5: s = h->strong->dump_states(); break;
    case 7: s = h->boost->dump_states(); break;
    case 8: s = h->hybrid->dump_states(); break;
  }
  *w = s.empty() ? 0 : static_cast<int>(s[0].size());
  size_t need = s.size() * static_cast<size_t>(*w);
  if (need > static_cast<size_t>(cap_floats)) return -static_cast<int>(s.size());
  for (size_t i = 0; i < s.size(); ++i) std::memcpy(out + i * *w, s[i].data(), sizeof(float) * *w);
  return static_cast<int>(s.size());
}

// ---- primitives (row-major everywhere) -------------------------------------------------
void orc_iou_batch(const float* a, int n, int ca, const float* b, int m, int cb, float* out) {
  Mat r = iou_batch(as_mat(a, n, ca), as_mat(b, m, cb));
  if (r.size()) std::memcpy(out, r.a.data(), sizeof(float) * r.size());
}
void orc_iou_distance(const float* a, int n, const float* b, int m, float* out) {
  Mat r = iou_distance(as_mat(a, n, 4), as_mat(b, m, 4));
  if (r.size()) std::memcpy(out, r.a.data(), sizeof(float) * r.size());
}
void orc_fuse_score(const float* cost, int n, int m, const float* conf, float* out) {
  Mat r = fuse_score(as_mat(cost, n, m), std::vector<float>(conf, conf + m));
  if (r.size()) std::memcpy(out, r.a.data(), sizeof(float) * r.size());
}
void orc_cosine_distance(const float* t, int n, const float* dd, int m, int d, float* out) {
  Mat r = embedding_distance_cosine(as_mat(t, n, d), as_mat(dd, m, d));
  if (r.size()) std::memcpy(out, r.a.data(), sizeof(float) * r.size());
}
// x: n ints, y: m ints (-1 = unmatched)
// matching.cpp:93-101 — euclidean metric: |t_i - d_j|, restated as the k-ordered chain of squared differences (Eigen's
// reduction order is unspecified; same convention as dot_chain); metric 1: the raw inner product t_i . d_j (deepocsort.cpp:404)
void orc_embedding_distance(int metric, const float* t, int n, const float* dd, int m, int d, float* out) {
  if (metric == 0) { orc_cosine_distance(t, n, dd, m, d, out); return; }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const float* a = t + static_cast<size_t>(i) * d;
      const float* b = dd + static_cast<size_t>(j) * d;
      float s = 0.0f;
      if (metric == 1) s = dot_chain(a, b, d);
      else { for (int k = 0; k < d; ++k) { const float df = a[k] - b[k]; s = std::fmaf(df, df, s); } s = std::sqrt(s); }
      out[static_cast<size_t>(i) * m + j] = s;
    }
}
// gating distances and the two blends built on them (StrongSORT's gate; matching.hpp:60-94, strongsort.cpp:449-492).
// kind 1 = XYAH (BaseKalmanFilter), 2 = XYWH. mean [n][8], cov [n][64], meas [m][4]; mode 0: distances, 1: fuse_motion,
// 2: gate_cost_matrix. out n x m row-major.
void orc_gate_cost(int kind, int mode, int n, int m, const float* mean, const float* cov, const float* meas, const float* cost,
                   int only_position, int metric, float lambda, float gated_cost, float* out) {
  static const float chi2inv95[] = {3.8415f, 5.9915f, 7.8147f, 9.4877f};  // matching.hpp:16-26
  std::vector<float> g(static_cast<size_t>(m > 0 ? m : 1));
  for (int i = 0; i < n; ++i) {
    State8 s;
    for (int k = 0; k < 8; ++k) s.mean[k] = mean[8 * i + k];
    for (int a = 0; a < 8; ++a)
      for (int b = 0; b < 8; ++b) s.cov[a][b] = cov[64 * static_cast<size_t>(i) + 8 * a + b];
    if (kind == 1) gating_xyah(s, meas, m, only_position != 0, metric, g.data());
    else gating_xywh(s, meas, m, only_position != 0, g.data());
    for (int j = 0; j < m; ++j) {
      const size_t o = static_cast<size_t>(i) * m + j;
      if (mode == 0) out[o] = g[j];
      else if (mode == 1) {
        const float thr = chi2inv95[(only_position ? 2 : 4) - 1];
        out[o] = (g[j] > thr) ? std::numeric_limits<float>::infinity() : lambda * cost[o] + (1.0f - lambda) * g[j];
      } else {
        float c = cost[o];
        if (g[j] > 9.4877f) c = gated_cost;
        out[o] = lambda * c + (1.0f - lambda) * g[j];
      }
    }
  }
}
// fuse_iou, matching.cpp:109-128
void orc_fuse_iou(const float* reid, int n, int m, const float* a, const float* b, float* out) {
  Mat d = iou_distance(as_mat(a, n, 4), as_mat(b, m, 4));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const size_t o = static_cast<size_t>(i) * m + j;
      const float reid_sim = 1.0f - reid[o];
      const float iou_sim = 1.0f - d.a[o];
      const float fuse_sim = reid_sim * ((1.0f + iou_sim) / 2.0f);
      out[o] = 1.0f - fuse_sim;
    }
}
void orc_linear_assignment(const float* cost, int n, int m, float thresh, int* x, int* y) {
  LapResult r = linear_assignment(as_mat(cost, n, m), thresh);
  if (n) std::memcpy(x, r.x.data(), sizeof(int) * n);
  if (m) std::memcpy(y, r.y.data(), sizeof(int) * m);
}
// association measures (iou.hpp:122-414): kind 0 iou, 1 hmiou, 2 giou, 3 ciou, 4 diou, 5 centroid; a n x 4, b m x 4 -> out n x m
void orc_asso_batch(int kind, const float* a, int n, const float* b, int m, int frame_w, int frame_h, float* out) {
  Mat r = asso_batch(kind, as_mat(a, n, 4), as_mat(b, m, 4), frame_w, frame_h);
  std::memcpy(out, r.a.data(), sizeof(float) * r.a.size());
}
// work counters of the last assignment solved on this thread (instrumentation)
// Appearance post-processing. mode 0: BotSTrack ctor botsort.cpp:38-46 (set, normalise if norm > 0);
// 1: update_features botsort.cpp:158-169 (EMA with alpha, normalise if norm > 0); 2: ReIDBackend::normalize_features
// reid_backend.cpp:72-88 (normalise if norm > 1e-6). norm = sqrt of the k-ordered inner product (orc_math.hpp dot_chain).
// modes 0-2: botsort.cpp:38-46 / :158-169 / reid_backend.cpp:72-88; mode 3: DeepOC-SORT's update_emb (deepocsort.cpp:132-150): EMA with
// the detection's own weight, normalised where the norm exceeds 1e-6. alpha_i: optional per-row weights.
void orc_feat_update_alpha(int mode, float alpha, const float* alpha_i, int n, int d, float* feat, const float* src) {
  for (int i = 0; i < n; ++i) {
    float* f = feat + static_cast<size_t>(i) * d;
    const float* s = src + static_cast<size_t>(i) * d;
    const float a = alpha_i ? alpha_i[i] : alpha;
    for (int k = 0; k < d; ++k) f[k] = (mode == 1 || mode == 3) ? a * f[k] + (1.0f - a) * s[k] : s[k];
    const float nn = std::sqrt(dot_chain(f, f, d));
    const bool go = (mode >= 2) ? (nn > 1e-6f) : (nn > 0.0f);
    if (go) for (int k = 0; k < d; ++k) f[k] /= nn;
  }
}
void orc_feat_update(int mode, float alpha, int n, int d, float* feat, const float* src) { orc_feat_update_alpha(mode, alpha, nullptr, n, d, feat, src); }
// smooth features of the live BoT-SORT tracks, dump_states order: returns rows, *d = feature length (0: none yet)
int orc_tracker_dump_features(void* hv, float* out, int cap_floats, int* d) {
  auto* h = static_cast<Handle*>(hv);
  *d = 0;
  if (h->kind != 3 && h->kind != 4 && h->kind != 5 && h->kind != 7) return 0;
  const std::vector<std::vector<float>> f = (h->kind == 3) ? h->bot->dump_features() : ((h->kind == 4) ? h->deep->dump_features() : ((h->kind == 5) ? h->strong->dump_features() : h->boost->dump_features()));
  for (const auto& r : f) if (!r.empty()) *d = static_cast<int>(r.size());
  if (*d == 0) return static_cast<int>(f.size());
  if (f.size() * static_cast<size_t>(*d) > static_cast<size_t>(cap_floats)) return -static_cast<int>(f.size());
  for (size_t i = 0; i < f.size(); ++i)
    for (int k = 0; k < *d; ++k) out[i * *d + k] = (k < static_cast<int>(f[i].size())) ? f[i][k] : 0.0f;
  return static_cast<int>(f.size());
}
// arithmetic mode of everything Eigen-dependent (orc_kf.hpp): 0 = the canonical order, 1 = the alternative orders
void orc_set_arith_mode(int m) { arith_mode() = m; }
int orc_get_arith_mode() { return arith_mode(); }
void orc_lap_stats(long* out) {
  const LapStats& s = lap_stats();
  long v[9] = {s.n, s.free_after_colred, s.unique_rows, s.carr_iters, s.paths, s.finds, s.find_records, s.scan_rows, s.scan_ties};
  for (int i = 0; i < 9; ++i) out[i] = v[i];
}
// OC-SORT first-stage association (ocsort.cpp:610-738). dets nd x 5, trks nt x 5, vel nt x 2, prev nt x 5.
// Writes matches as (det,trk) pairs; um lists may contain duplicates (Q4). Returns match count.
int orc_ocsort_associate(const float* dets, int nd, const float* trks, int nt, const float* vel,
                         const float* prev, float thr, float vdc, int* matches, int* um_d, int* n_umd,
                         int* um_t, int* n_umt, int* used_lap) {
  OCSort o;
  OCSort::Assoc a = o.associate(as_mat(dets, nd, 5), as_mat(trks, nt, 5), thr, as_mat(vel, nt, 2),
                                as_mat(prev, nt, 5), vdc);
  for (size_t i = 0; i < a.matches.size(); ++i) { matches[2 * i] = a.matches[i][0]; matches[2 * i + 1] = a.matches[i][1]; }
  *n_umd = static_cast<int>(a.um_dets.size());
  *n_umt = static_cast<int>(a.um_trks.size());
  std::memcpy(um_d, a.um_dets.data(), sizeof(int) * a.um_dets.size());
  std::memcpy(um_t, a.um_trks.data(), sizeof(int) * a.um_trks.size());
  *used_lap = o.laps.empty() ? 0 : 1;
  return static_cast<int>(a.matches.size());
}
// OC-SORT final cost matrix -(iou + angle) and the iou matrix, both nd x nt (for kernel parity)
void orc_ocsort_cost(const float* dets, int nd, const float* trks, int nt, const float* vel,
                     const float* prev, float vdc, float* cost, float* iou) {
  Mat D = as_mat(dets, nd, 5), T = as_mat(trks, nt, 5), V = as_mat(vel, nt, 2), Pv = as_mat(prev, nt, 5);
  Mat I = iou_batch(D, T);
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nd; ++j) {
      float cx1 = (D(j, 0) + D(j, 2)) / 2.0f, cy1 = (D(j, 1) + D(j, 3)) / 2.0f;
      float cx2 = (Pv(i, 0) + Pv(i, 2)) / 2.0f, cy2 = (Pv(i, 1) + Pv(i, 3)) / 2.0f;
      float dx = cx1 - cx2, dy = cy1 - cy2;
      float norm = std::sqrt(dx * dx + dy * dy) + 1e-6f;
      float Y = dy / norm, X = dx / norm;
      float c = V(i, 1) * X + V(i, 0) * Y;
      c = std::min(std::max(c, -1.0f), 1.0f);
      const float PI = 3.14159265358979323846f;
      float da = (PI / 2.0f - std::fabs(OCSort::acos_f32(c))) / PI;
      float valid = (Pv(i, 4) >= 0.0f) ? 1.0f : 0.0f;
      float a = ((valid * da) * vdc) * D(j, 4);
      cost[static_cast<size_t>(j) * nt + i] = -(I(j, i) + a);
      iou[static_cast<size_t>(j) * nt + i] = I(j, i);
    }
}

// ---- Kalman primitives on arrays of states ---------------------------------------------
// kind 0 XYSR (d=7), 1 XYAH (d=8), 2 XYWH (d=8). mean: n x d, cov: n x d x d, row-major.
// XYSR extra: q = [Q44, Q55, Q66] process-noise diagonal entries (others 1), may be null.
static void load_xysr(KfXYSR& k, const float* mean, const float* cov, const float* q) {
  for (int i = 0; i < 7; ++i) k.x[i] = mean[i];
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) k.P[i][j] = cov[i * 7 + j];
  if (q) { k.Q[4][4] = q[0]; k.Q[5][5] = q[1]; k.Q[6][6] = q[2]; }
}
static void store_xysr(const KfXYSR& k, float* mean, float* cov) {
  for (int i = 0; i < 7; ++i) mean[i] = k.x[i];
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) cov[i * 7 + j] = k.P[i][j];
}
static void load8(State8& s, const float* mean, const float* cov) {
  for (int i = 0; i < 8; ++i) s.mean[i] = mean[i];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s.cov[i][j] = cov[i * 8 + j];
}
static void store8(const State8& s, float* mean, float* cov) {
  for (int i = 0; i < 8; ++i) mean[i] = s.mean[i];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) cov[i * 8 + j] = s.cov[i][j];
}
void orc_kf_initiate(int kind, int n, const float* meas, const float* q, float* mean, float* cov) {
  (void)q;
  for (int t = 0; t < n; ++t) {
    const float* z = meas + 4 * t;
    if (kind == 0) {
      KfXYSR k;
      for (int i = 0; i < 4; ++i) k.x[i] = z[i];
      store_xysr(k, mean + 7 * t, cov + 49 * t);
    } else {
      State8 s = (kind == 1) ? KfXYAH::initiate(z) : KfXYWH::initiate(z);
      store8(s, mean + 8 * t, cov + 64 * t);
    }
  }
}
void orc_kf_predict(int kind, int n, const float* q, float* mean, float* cov) {
  for (int t = 0; t < n; ++t) {
    if (kind == 0) {
      KfXYSR k; load_xysr(k, mean + 7 * t, cov + 49 * t, q);
      k.predict(); store_xysr(k, mean + 7 * t, cov + 49 * t);
    } else {
      State8 s; load8(s, mean + 8 * t, cov + 64 * t);
      if (kind == 1) KfXYAH::predict(s); else KfXYWH::predict(s);
      store8(s, mean + 8 * t, cov + 64 * t);
    }
  }
}
void orc_kf_update(int kind, int n, const float* meas, const float* q, float* mean, float* cov) {
  for (int t = 0; t < n; ++t) {
    const float* z = meas + 4 * t;
    if (kind == 0) {
      KfXYSR k; load_xysr(k, mean + 7 * t, cov + 49 * t, q);
      k.update(z); store_xysr(k, mean + 7 * t, cov + 49 * t);
    } else {
      State8 s; load8(s, mean + 8 * t, cov + 64 * t);
      if (kind == 1) KfXYAH::update(s, z); else KfXYWH::update(s, z);
      store8(s, mean + 8 * t, cov + 64 * t);
    }
  }
}
// XYAH update with per-measurement confidences (NSA Kalman, kalman_filter.cpp:60-75; StrongSORT's Track::update, strongsort.cpp:153)
void orc_kf_update_conf(int n, const float* meas, const float* conf, float* mean, float* cov) {
  for (int t = 0; t < n; ++t) {
    State8 s; load8(s, mean + 8 * t, cov + 64 * t);
    KfXYAH::update(s, meas + 4 * t, conf ? conf[t] : 0.0f);
    store8(s, mean + 8 * t, cov + 64 * t);
  }
}
// camera-motion warp of stored states, warp9 = 3x3 row-major: kind 0 KalmanFilterXYSR::apply_affine_correction
// (m = W[0:2,0:2], t = W[0:2,2]); kind 2 BotSTrack::multi_gmc on the XYWH state. Returns -1 for XYAH (no such step).
int orc_kf_warp(int kind, int n, const float* warp9, float* mean, float* cov) {
  if (kind == 1) return -1;
  for (int t = 0; t < n; ++t) {
    if (kind == 0) {
      KfXYSR k; load_xysr(k, mean + 7 * t, cov + 49 * t, nullptr);
      const float m[2][2] = {{warp9[0], warp9[1]}, {warp9[3], warp9[4]}};
      const float tr[2] = {warp9[2], warp9[5]};
      k.apply_affine_correction(m, tr);
      store_xysr(k, mean + 7 * t, cov + 49 * t);
    } else {
      BotSort::BTrack b;
      b.has_state = true;
      load8(b.kf, mean + 8 * t, cov + 64 * t);
      float W[3][3];
      for (int i = 0; i < 9; ++i) W[i / 3][i % 3] = warp9[i];
      BotSort::gmc(b, W);
      store8(b.kf, mean + 8 * t, cov + 64 * t);
    }
  }
  return 0;
}
// box conversions exposed for known-answer tests: op 0 xyxy2xysr, 1 xysr2xyxy, 2 xyxy2xywh, 3 xywh2xyxy,
// 4 xywh2tlwh, 5 tlwh2xyah, 6 xyah2xywh
void orc_box_convert(int op, int n, const float* in, float* out) {
  for (int i = 0; i < n; ++i) {
    Box b{in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]}, r{};
    switch (op) {
      case 0: r = xyxy2xysr(b); break;
      case 1: r = xysr2xyxy(b); break;
      case 2: r = xyxy2xywh(b); break;
      case 3: r = xywh2xyxy(b); break;
      case 4: r = xywh2tlwh(b); break;
      case 5: r = tlwh2xyah(b); break;
      case 6: r = xyah2xywh(b); break;
    }
    for (int k = 0; k < 4; ++k) out[4 * i + k] = r[k];
  }
}

}  // extern "C"
