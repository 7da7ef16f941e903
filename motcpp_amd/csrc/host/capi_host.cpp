// C handles over the stage machines (include/motcpp_c.h).
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <atomic>
#include <functional>
#include "motcpp_c.h"
#include "pool.hpp"
#include "staged.hpp"

using namespace motcpp::rt;

namespace {
thread_local std::string g_err;
float P(const float* p, int n, int i, float dflt) { return (p && i < n) ? p[i] : dflt; }

Staged* make(std::shared_ptr<Device> dev, int kind, const float* p, int np) {
  switch (kind) {
    case 0: return make_sort(dev, P(p, np, 0, 0.3f), (int)P(p, np, 1, 1), (int)P(p, np, 2, 50), (int)P(p, np, 3, 3), P(p, np, 4, 0.3f));
    case 1: return make_bytetrack(dev, P(p, np, 0, 0.1f), P(p, np, 1, 0.45f), P(p, np, 2, 0.8f), (int)P(p, np, 3, 25),
                                  (int)P(p, np, 4, 30), (int)P(p, np, 5, 30), (int)P(p, np, 6, 50));
    case 2:
      if ((int)P(p, np, 11, 0.f) < 0 || (int)P(p, np, 11, 0.f) > 5) throw Error("OC-SORT: association measure (param 11) must be a mot_assoc value in [0, 5]");
      return make_ocsort(dev, P(p, np, 0, 0.2f), (int)P(p, np, 1, 30), (int)P(p, np, 2, 50), (int)P(p, np, 3, 3), P(p, np, 4, 0.3f),
                               P(p, np, 5, 0.1f), (int)P(p, np, 6, 3), P(p, np, 7, 0.2f), P(p, np, 8, 0.f) != 0.f, P(p, np, 9, 0.01f),
                               P(p, np, 10, 0.0001f), (int)P(p, np, 11, 0.f));
    case 3: return make_botsort(dev, P(p, np, 0, 0.5f), P(p, np, 1, 0.1f), P(p, np, 2, 0.6f), (int)P(p, np, 3, 30), P(p, np, 4, 0.8f),
                                P(p, np, 5, 0.5f), P(p, np, 6, 0.25f), (int)P(p, np, 7, 30), P(p, np, 8, 0.f) != 0.f,
                                P(p, np, 9, 1.f) != 0.f, (int)P(p, np, 10, 30), (int)P(p, np, 11, 50));
    case 4:  // det_thresh, max_age, max_obs, min_hits, iou_thr, delta_t, inertia, w_emb, alpha_fixed, aw_param, emb_off, cmc_off, aw_off, q_xy, q_s, asso
      if ((int)P(p, np, 15, 0.f) < 0 || (int)P(p, np, 15, 0.f) > 5) throw Error("DeepOC-SORT: association measure (param 15) must be a mot_assoc value in [0, 5]");
      return make_deepocsort(dev, P(p, np, 0, 0.3f), (int)P(p, np, 1, 30), (int)P(p, np, 2, 50), (int)P(p, np, 3, 3), P(p, np, 4, 0.3f),
                             (int)P(p, np, 5, 3), P(p, np, 6, 0.2f), P(p, np, 7, 0.5f), P(p, np, 8, 0.95f), P(p, np, 9, 0.5f),
                             P(p, np, 10, 0.f) != 0.f, P(p, np, 11, 0.f) != 0.f, P(p, np, 12, 0.f) != 0.f, P(p, np, 13, 0.01f),
                             P(p, np, 14, 0.0001f), (int)P(p, np, 15, 0.f));
    case 5:  // min_conf, max_cos_dist, max_iou_dist, n_init, nn_budget, mc_lambda, ema_alpha, max_age
      return make_strongsort(dev, P(p, np, 0, 0.1f), P(p, np, 1, 0.2f), P(p, np, 2, 0.7f), (int)P(p, np, 3, 3), (int)P(p, np, 4, 100), P(p, np, 5, 0.98f),
                             P(p, np, 6, 0.9f), (int)P(p, np, 7, 30));
    case 8: {  // HybridSORT: det_thresh, max_age, min_hits, iou_threshold, asso (0 iou, 1 hmiou), low_thresh, use_byte, track_thresh,
               // EG_weight_high_score, EG_weight_low_score, TCM_first_step, TCM_byte_step, TCM_byte_step_weight, with_reid (no embeddings)
      HybridParams q;
      q.det_thresh = P(p, np, 0, 0.7f); q.max_age = (int)P(p, np, 1, 30); q.min_hits = (int)P(p, np, 2, 3); q.iou_threshold = P(p, np, 3, 0.15f);
      q.asso = (int)P(p, np, 4, 1.f); q.low_thresh = P(p, np, 5, 0.1f); q.use_byte = P(p, np, 6, 1.f) != 0.f; q.track_thresh = P(p, np, 7, 0.5f);
      q.eg_high = P(p, np, 8, 4.6f); q.eg_low = P(p, np, 9, 1.3f); q.tcm_first = P(p, np, 10, 1.f) != 0.f; q.tcm_byte = P(p, np, 11, 1.f) != 0.f;
      q.tcm_byte_weight = P(p, np, 12, 1.0f); q.with_reid = P(p, np, 13, 0.f) != 0.f;
      return make_hybridsort(dev, q);
    }
    case 7: {  // BoostTrack, motion only: det_thresh, max_age, min_hits, iou_threshold, min_box_area, aspect_ratio_thresh, lambda_iou, lambda_mhd,
               // lambda_shape, use_dlo_boost, use_duo_boost, dlo_boost_coef, use_sb, use_vt
      BoostParams q;
      q.det_thresh = P(p, np, 0, 0.6f); q.max_age = (int)P(p, np, 1, 60); q.min_hits = (int)P(p, np, 2, 3); q.iou_threshold = P(p, np, 3, 0.3f);
      q.min_box_area = (int)P(p, np, 4, 10); q.aspect_ratio_thresh = P(p, np, 5, 1.6f); q.lambda_iou = P(p, np, 6, 0.5f); q.lambda_mhd = P(p, np, 7, 0.25f);
      q.lambda_shape = P(p, np, 8, 0.25f); q.use_dlo = P(p, np, 9, 1.f) != 0.f; q.use_duo = P(p, np, 10, 1.f) != 0.f; q.dlo_coef = P(p, np, 11, 0.65f);
      q.use_sb = P(p, np, 12, 0.f) != 0.f; q.use_vt = P(p, np, 13, 0.f) != 0.f; q.with_reid = P(p, np, 14, 0.f) != 0.f;
      return make_boosttrack(dev, q);
    }
    case 6: {  // det_thresh, max_age, a1, a2, wx, wy, vmax, fps (dt = 1.0 / fps in double precision, as the evaluation tool forms it), high_score
      UcmcParams q;
      q.det_thresh = P(p, np, 0, 0.3f); q.max_age = (int)P(p, np, 1, 30); q.a1 = P(p, np, 2, 100.f); q.a2 = P(p, np, 3, 100.f);
      q.wx = P(p, np, 4, 5.f); q.wy = P(p, np, 5, 5.f); q.vmax = P(p, np, 6, 10.f); q.dt = 1.0 / static_cast<double>(P(p, np, 7, 30.f));
      q.high_score = P(p, np, 8, 0.5f);
      return make_ucmc(dev, q);
    }
  }
  throw Error("unknown tracker kind");
}
}  // namespace

struct motcpp_tracker {
  std::shared_ptr<Device> dev;
  std::unique_ptr<Staged> impl;          // host stage machine, or
  std::unique_ptr<PooledStream> pooled;  // a stream of a shared device-lifecycle batch (motcpp_tracker_create_pooled)
  std::vector<float> colmajor;
};
struct motcpp_batch {
  std::shared_ptr<Device> dev;
  std::vector<std::unique_ptr<motcpp_tracker>> trk;
  std::vector<std::vector<float>> colmajor;
  long frames = 0;
  std::unique_ptr<Team> team;  // nullptr: the caller's thread does everything
};

namespace {
void to_colmajor(const float* rows, int n, std::vector<float>& out) {
  out.resize(static_cast<size_t>(6) * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 6; ++k) out[static_cast<size_t>(k) * n + i] = rows[static_cast<size_t>(i) * 6 + k];
}
FrameIn frame_in(const std::vector<float>& cm, int n, const float* embs, int d) {
  FrameIn in;
  in.dets = cm.data(); in.n = n; in.ld = n;
  if (embs && d > 0 && n > 0) { in.embs = embs; in.emb_dim = d; in.emb_ld = d; in.embs_rowmajor = true; }
  in.img_w = 1920; in.img_h = 1080;
  return in;
}
int copy_rows(const std::vector<float>& rows, float* out, int cap) {
  const int m = static_cast<int>(rows.size() / 8);
  if (m > cap) return -m - 1000000;
  if (m) std::memcpy(out, rows.data(), sizeof(float) * rows.size());
  return m;
}
}  // namespace

extern "C" {

const char* motcpp_last_error(void) { return g_err.c_str(); }

motcpp_tracker* motcpp_tracker_create(int kind, const float* params, int nparams, int device) {
  try {
    auto t = std::make_unique<motcpp_tracker>();
    t->dev = Device::shared(device);
    t->impl.reset(make(t->dev, kind, params, nparams));
    return t.release();
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
// UCMCTrack with a calibrated camera: params as for kind 6, Ki 3 x 4 and Ko 4 x 4 row-major (CameraMapper, ucmc.cpp:57-83)
motcpp_tracker* motcpp_ucmc_create(const float* p, int np, const double* Ki12, const double* Ko16, int device) {
  try {
    auto t = std::make_unique<motcpp_tracker>();
    t->dev = Device::shared(device);
    UcmcParams q;
    q.det_thresh = P(p, np, 0, 0.3f); q.max_age = (int)P(p, np, 1, 30); q.a1 = P(p, np, 2, 100.f); q.a2 = P(p, np, 3, 100.f);
    q.wx = P(p, np, 4, 5.f); q.wy = P(p, np, 5, 5.f); q.vmax = P(p, np, 6, 10.f); q.dt = 1.0 / static_cast<double>(P(p, np, 7, 30.f));
    q.high_score = P(p, np, 8, 0.5f);
    if (Ki12 && Ko16) { q.has_camera = true; std::memcpy(q.Ki, Ki12, sizeof(q.Ki)); std::memcpy(q.Ko, Ko16, sizeof(q.Ko)); }
    t->impl.reset(make_ucmc(t->dev, q));
    return t.release();
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
// UCMCTrack's tracks in list order, rows of 26 doubles [id, state, death, birth, det_idx, age, x(4), P(16)]; returns the rows (-needed - 1000000)
int motcpp_tracker_dump_f64(motcpp_tracker* t, double* out, int cap_rows) {
  try {
    std::vector<double> rows;
    if (t->pooled || !t->impl->f64_states(&rows)) { g_err = "this tracker keeps no double-precision states"; return -1; }
    const int n = static_cast<int>(rows.size() / 26);
    if (n > cap_rows) return -n - 1000000;
    if (n) std::memcpy(out, rows.data(), rows.size() * sizeof(double));
    return n;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
motcpp_tracker* motcpp_tracker_create_pooled(int kind, const float* params, int nparams, int device) {
  try {
    auto t = std::make_unique<motcpp_tracker>();
    std::vector<float> dp;
    const int pk = pooled_params(kind, params, nparams, &dp);
    t->pooled = std::make_unique<PooledStream>(device, pk, dp.data(), static_cast<int>(dp.size()));
    return t.release();
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
int motcpp_pool_stats(long* out8, int reset) {
  const PoolStats s = pool_stats(reset != 0);
  out8[0] = s.rounds; out8[1] = s.frames; out8[2] = s.moves; out8[3] = s.max_round;
  out8[4] = static_cast<long>(s.us_window); out8[5] = static_cast<long>(s.us_gather); out8[6] = static_cast<long>(s.us_run); out8[7] = static_cast<long>(s.us_enqueue);
  return 0;
}
int motcpp_tracker_pool_level(motcpp_tracker* t) { return t->pooled ? t->pooled->level() : -1; }
void motcpp_tracker_destroy(motcpp_tracker* t) { delete t; }
int motcpp_tracker_reset(motcpp_tracker* t) {
  try { if (t->pooled) t->pooled->reset(); else t->impl->reset(); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int motcpp_tracker_set_camera_motion(motcpp_tracker* t, const float* warp2x3) {
  if (t->pooled) { t->pooled->set_camera_motion(warp2x3); return 0; }
  if (t->impl->set_camera_motion(warp2x3)) return 0;
  g_err = "this tracker has no camera-motion step (BoT-SORT only)";
  return -1;
}
int motcpp_tracker_update(motcpp_tracker* t, const float* dets, int n, const float* embs, int d, float* out, int cap) {
  try {
    to_colmajor(dets, n, t->colmajor);
    if (t->pooled) {
      PooledFrame f;
      f.dets = t->colmajor.data(); f.n = n; f.ld = n;
      if (embs && d > 0 && n > 0) { f.embs = embs; f.emb_ld = d; f.emb_dim = d; f.embs_rowmajor = true; }
      f.img_w = 1920; f.img_h = 1080;
      const float* rows = nullptr;
      const int m = t->pooled->update(f, &rows);
      if (m > cap) return -m - 1000000;
      if (m) std::memcpy(out, rows, sizeof(float) * 8 * static_cast<size_t>(m));
      return m;
    }
    FrameIn in = frame_in(t->colmajor, n, embs, d);
    Staged* s = t->impl.get();
    run_frame_combined(t->dev, s, in);  // (merged with the calls other threads make on trackers of the same Device)
    return copy_rows(s->rows(), out, cap);
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int motcpp_tracker_lap_count(motcpp_tracker* t) { return t->pooled ? 0 : static_cast<int>(t->impl->laps().size()); }
int motcpp_tracker_lap_get(motcpp_tracker* t, int k, int* n, int* m, int* x, int* y, int cap) {
  if (t->pooled) return -1;  // (the assignments of a pooled stream stay on the device)
  const auto& v = t->impl->laps();
  if (k < 0 || k >= static_cast<int>(v.size())) return -1;
  *n = static_cast<int>(v[k].x.size()); *m = static_cast<int>(v[k].y.size());
  if (*n > cap || *m > cap) return -2;
  if (*n) std::memcpy(x, v[k].x.data(), sizeof(int) * *n);
  if (*m) std::memcpy(y, v[k].y.data(), sizeof(int) * *m);
  return 0;
}
int motcpp_tracker_dump_states(motcpp_tracker* t, float* out, int cap_floats, int* width) {
  try {
    if (t->pooled) {
      std::vector<int> ids;
      std::vector<float> mean, cov;
      const int rows = t->pooled->dump(&ids, &mean, &cov, nullptr, nullptr);
      const int D = t->pooled->state_dim(), w = 1 + D + D * D;
      *width = w;
      if (static_cast<size_t>(rows) * w > static_cast<size_t>(cap_floats)) return -rows - 1000000;
      for (int r = 0; r < rows; ++r) {
        float* o = out + static_cast<size_t>(r) * w;
        o[0] = static_cast<float>(ids[r]);
        std::memcpy(o + 1, mean.data() + static_cast<size_t>(r) * D, sizeof(float) * D);
        std::memcpy(o + 1 + D, cov.data() + static_cast<size_t>(r) * D * D, sizeof(float) * D * D);
      }
      return rows;
    }
    {
      std::vector<float> own;
      int w = 0;
      if (t->impl->f32_states(&own, &w)) {
        *width = w;
        const int rows = w > 0 ? static_cast<int>(own.size() / w) : 0;
        if (own.size() > static_cast<size_t>(cap_floats)) return -rows - 1000000;
        if (!own.empty()) std::memcpy(out, own.data(), own.size() * sizeof(float));
        return rows;
      }
    }
    std::vector<int> ids, slots;
    t->impl->live_tracks(&ids, &slots);
    Core& c = t->impl->core();
    const int D = mot_kf_dim(c.kf_kind()), w = 1 + D + D * D, cap = c.cap();
    *width = w;
    const int rows = static_cast<int>(ids.size());
    if (static_cast<size_t>(rows) * w > static_cast<size_t>(cap_floats)) return -rows - 1000000;
    if (rows == 0) return 0;
    const size_t rec = static_cast<size_t>(D) * (D + 1);  // slab record: mean then covariance
    std::vector<float> slab(rec * cap);
    c.dev().check(mot_memcpy_d2h(c.dev().ctx, slab.data(), c.d_mean(), slab.size() * sizeof(float)), "state readback");
    c.dev().check(mot_ctx_sync(c.dev().ctx), "state readback");
    for (int r = 0; r < rows; ++r) {
      float* o = out + static_cast<size_t>(r) * w;
      o[0] = static_cast<float>(ids[r]);
      for (size_t k = 0; k < rec; ++k) o[1 + k] = slab[rec * slots[r] + k];
    }
    return rows;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// BoT-SORT: smooth features of the live tracks, dump_states order, rows of *dim floats (a track without a feature yet: zeros)
int motcpp_tracker_dump_features(motcpp_tracker* t, float* out, int cap_floats, int* dim) {
  try {
    if (t->pooled) {
      std::vector<int> ids;
      std::vector<float> mean, cov, feats;
      std::vector<unsigned char> hasf;
      const int rows = t->pooled->dump(&ids, &mean, &cov, &feats, &hasf);
      const int E = rows > 0 ? static_cast<int>(feats.size() / static_cast<size_t>(rows)) : 0;
      *dim = E;
      if (E <= 0) return rows;
      if (static_cast<size_t>(rows) * E > static_cast<size_t>(cap_floats)) return -rows - 1000000;
      for (int r = 0; r < rows; ++r) {
        if (hasf[r]) std::memcpy(out + static_cast<size_t>(r) * E, feats.data() + static_cast<size_t>(r) * E, sizeof(float) * E);
        else std::memset(out + static_cast<size_t>(r) * E, 0, sizeof(float) * E);
      }
      return rows;
    }
    std::vector<int> ids, slots;
    t->impl->live_tracks(&ids, &slots);
    std::vector<char> has;
    const float* slab = t->impl->feature_slab(dim, &has);
    const int rows = static_cast<int>(ids.size());
    if (!slab || *dim <= 0) { *dim = 0; return rows; }
    if (static_cast<size_t>(rows) * *dim > static_cast<size_t>(cap_floats)) return -rows - 1000000;
    Core& c = t->impl->core();
    for (int r = 0; r < rows; ++r) {
      float* o = out + static_cast<size_t>(r) * *dim;
      if (has[r]) c.dev().check(mot_memcpy_d2h(c.dev().ctx, o, slab + static_cast<size_t>(slots[r]) * *dim, sizeof(float) * *dim), "feature readback");
      else std::memset(o, 0, sizeof(float) * *dim);
    }
    c.dev().check(mot_ctx_sync(c.dev().ctx), "feature readback");
    return rows;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

static motcpp_batch* batch_create(int kind, const float* params, int nparams, int nstreams, int device, bool private_dev) {
  try {
    auto b = std::make_unique<motcpp_batch>();
    b->dev = private_dev ? std::make_shared<Device>(device) : Device::shared(device);
    for (int s = 0; s < nstreams; ++s) {
      auto t = std::make_unique<motcpp_tracker>();
      t->dev = b->dev;
      t->impl.reset(make(b->dev, kind, params, nparams));
      b->trk.push_back(std::move(t));
    }
    b->colmajor.resize(nstreams);
    return b.release();
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
motcpp_batch* motcpp_batch_create(int kind, const float* params, int nparams, int nstreams, int device) {
  return batch_create(kind, params, nparams, nstreams, device, false);
}
motcpp_batch* motcpp_batch_create_private(int kind, const float* params, int nparams, int nstreams, int device) {
  return batch_create(kind, params, nparams, nstreams, device, true);
}
int motcpp_batch_profile(motcpp_batch* b, int enable) { b->dev->profile = enable != 0; if (enable) b->dev->reset_stats(); return 0; }
int motcpp_batch_profile_stats(motcpp_batch* b, double* out, int cap_rows) {
  const int n = F_COUNT < cap_rows ? F_COUNT : cap_rows;
  for (int f = 0; f < n; ++f) {
    const FamilyStat& s = b->dev->stats[f];
    out[f * 5 + 0] = s.ms; out[f * 5 + 1] = static_cast<double>(s.launches); out[f * 5 + 2] = static_cast<double>(s.tasks);
    out[f * 5 + 3] = s.bytes; out[f * 5 + 4] = s.flops;
  }
  return n;
}
void motcpp_batch_destroy(motcpp_batch* b) { delete b; }
int motcpp_batch_set_threads(motcpp_batch* b, int threads) {
  try {
    if (threads > Device::kMaxHostThreads) threads = Device::kMaxHostThreads;
    b->team.reset();
    if (threads > 1) b->team = std::make_unique<Team>(threads);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int motcpp_batch_pin_threads(motcpp_batch* b, int first_cpu) {
  if (first_cpu < 0) return 0;
  if (b->team) b->team->pin(first_cpu);
  else Team(1).pin(first_cpu);
  return 0;
}
int motcpp_batch_record_laps(motcpp_batch* b, int on) { for (auto& t : b->trk) t->impl->record_laps = on != 0; return 0; }
int motcpp_batch_tracker_count(motcpp_batch* b) { return static_cast<int>(b->trk.size()); }
motcpp_tracker* motcpp_batch_tracker(motcpp_batch* b, int s) { return b->trk[s].get(); }
int motcpp_batch_counters(motcpp_batch* b, long* out3) {
  out3[0] = b->frames; out3[1] = b->dev->counters.flushes; out3[2] = b->dev->counters.launches;
  return 0;
}
int motcpp_batch_host_ms(motcpp_batch* b, double* out4) {
  out4[0] = b->dev->counters.ms_begin; out4[1] = b->dev->counters.ms_flush; out4[2] = b->dev->counters.ms_advance;
  out4[3] = b->dev->counters.ms_sync_wait;
  return 0;
}
static int batch_step_impl(motcpp_batch* b, const float* dets, const int* counts, int max_n, const float* d_dets,
                           const float* embs, int d, float* out, int* out_counts, int cap, const float* d_embs = nullptr) {
  try {
    const int S = static_cast<int>(b->trk.size());
    for (int s = 0; s < S; ++s)
      if (counts[s] < 0 || counts[s] > max_n) { g_err = "motcpp_batch_step: counts[s] must be in [0, max_n]"; return -1; }
    std::vector<FrameIn> in(S);
    std::vector<Staged*> st(S);
    auto for_streams = [&](const std::function<void(int)>& fn) {
      if (b->team && S > 8) b->team->parallel_for(S, fn);
      else for (int s = 0; s < S; ++s) fn(s);
    };
    for_streams([&](int s) {
      to_colmajor(dets + static_cast<size_t>(s) * max_n * 6, counts[s], b->colmajor[s]);
      in[s] = frame_in(b->colmajor[s], counts[s], embs ? embs + static_cast<size_t>(s) * max_n * d : nullptr, d);
      if (d_dets) { in[s].d_dets = d_dets + static_cast<size_t>(s) * 6 * max_n; in[s].d_ld = max_n; }
      if (d_embs && d > 0) { in[s].d_embs = d_embs + static_cast<size_t>(s) * max_n * d; in[s].emb_dim = d; in[s].embs_rowmajor = true; }
      st[s] = b->trk[s]->impl.get();
    });
    run_frame(*b->dev, st.data(), in.data(), S, b->team.get());
    b->frames += S;
    std::atomic<int> bad{0};
    for_streams([&](int s) {
      const int m = copy_rows(st[s]->rows(), out + static_cast<size_t>(s) * cap * 8, cap);
      if (m < 0) bad.store(1, std::memory_order_relaxed);
      else out_counts[s] = m;
    });
    if (bad.load()) { g_err = "output capacity too small"; return -1; }
    return S;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int motcpp_batch_step(motcpp_batch* b, const float* dets, const int* counts, int max_n, const float* embs, int d, float* out,
                      int* out_counts, int cap) {
  return batch_step_impl(b, dets, counts, max_n, nullptr, embs, d, out, out_counts, cap);
}
int motcpp_batch_step_resident(motcpp_batch* b, const float* dets, const int* counts, int max_n, const void* d_dets_soa,
                               const float* embs, int d, float* out, int* out_counts, int cap) {
  return batch_step_impl(b, dets, counts, max_n, static_cast<const float*>(d_dets_soa), embs, d, out, out_counts, cap);
}
int motcpp_batch_step_resident_embs(motcpp_batch* b, const float* dets, const int* counts, int max_n, const void* d_dets_soa,
                                    const void* d_embs, int d, float* out, int* out_counts, int cap) {
  return batch_step_impl(b, dets, counts, max_n, static_cast<const float*>(d_dets_soa), nullptr, d, out, out_counts, cap,
                         static_cast<const float*>(d_embs));
}
int motcpp_profile(int device, int enable) {
  try {
    auto dev = Device::shared(device);
    dev->profile = enable != 0;
    if (enable) dev->reset_stats();
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int motcpp_profile_stats(int device, double* out, int cap_rows) {
  try {
    auto dev = Device::shared(device);
    const int n = F_COUNT < cap_rows ? F_COUNT : cap_rows;
    for (int f = 0; f < n; ++f) {
      const FamilyStat& s = dev->stats[f];
      out[f * 5 + 0] = s.ms; out[f * 5 + 1] = static_cast<double>(s.launches); out[f * 5 + 2] = static_cast<double>(s.tasks);
      out[f * 5 + 3] = s.bytes; out[f * 5 + 4] = s.flops;
    }
    return n;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
