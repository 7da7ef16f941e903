#include "runtime.hpp"

#include <algorithm>
#include <chrono>
#include <map>

#include "team.hpp"

namespace motcpp::rt {

// ---- Arena ---------------------------------------------------------------------------------
Arena::Arena(mot_ctx* ctx, size_t chunk_bytes, bool host_mirror) : ctx_(ctx), chunk_bytes_(chunk_bytes), host_(host_mirror) {
  chunks_.reserve(1024);  // the vector never reallocates while other threads index it
  leases_.resize(Device::kMaxHostThreads);
}
Arena::~Arena() {
  for (auto& c : chunks_) {
    if (c->d) mot_free(ctx_, c->d);
    if (c->h) mot_host_free(ctx_, c->h);
  }
}
void Arena::raw_alloc(size_t bytes, void** h, void** d) {
  bytes = (bytes + 255) & ~size_t(255);
  if (bytes == 0) bytes = 256;
  // Small requests are served from a per-thread lease (a private slice of the current chunk): the shared bump pointer
  // is one cache line, and with tens of host threads on two sockets hammering it per allocation it becomes the
  // bottleneck of the whole host side. A lease dies at every transfer (its unused tail has already been copied).
  if (bytes <= kLeaseBytes / 4) {
    const int th = Team::worker_id();
    Lease& L = leases_[th < Device::kMaxHostThreads ? th : 0];
    const uint64_t ep = epoch_.load(std::memory_order_relaxed);
    if (L.epoch != ep || L.off + bytes > L.end) {
      void *lh, *ld;
      shared_alloc(kLeaseBytes, &lh, &ld);
      L.h = static_cast<char*>(lh); L.d = static_cast<char*>(ld);
      L.off = 0; L.end = kLeaseBytes; L.epoch = ep;
    }
    *h = L.h ? L.h + L.off : nullptr;
    *d = L.d + L.off;
    L.off += bytes;
    return;
  }
  shared_alloc(bytes, h, d);
}
void Arena::shared_alloc(size_t bytes, void** h, void** d) {
  while (true) {
    const size_t ci = cur_.load(std::memory_order_acquire);
    if (ci < n_chunks_.load(std::memory_order_acquire)) {
      Chunk& c = *chunks_[ci];
      if (bytes <= c.cap) {
        const size_t off = c.top.fetch_add(bytes, std::memory_order_relaxed);
        if (off + bytes <= c.cap) {
          *h = c.h ? c.h + off : nullptr;
          *d = c.d + off;
          return;
        }
      }
    }
    // slow path: this chunk is exhausted (or none exists yet) -> move on / grow, one thread at a time
    std::lock_guard<std::mutex> g(grow_mu_);
    if (cur_.load(std::memory_order_relaxed) != ci) continue;  // somebody else already advanced
    const size_t next = (ci < n_chunks_.load(std::memory_order_relaxed)) ? ci + 1 : ci;
    if (next >= n_chunks_.load(std::memory_order_relaxed)) {  // append a chunk (an existing but too-small next chunk is simply skipped)
      if (chunks_.size() >= 1024) throw Error("arena: too many chunks");
      auto c = std::make_unique<Chunk>();
      c->cap = std::max(chunk_bytes_, bytes);
      void* dp = nullptr;
      if (mot_malloc(ctx_, c->cap, &dp) != MOT_OK) throw Error(std::string("arena: device allocation failed: ") + mot_ctx_last_error(ctx_));
      c->d = static_cast<char*>(dp);
      if (host_) {
        void* hp = nullptr;
        if (mot_host_alloc(ctx_, c->cap, &hp) != MOT_OK) throw Error("arena: pinned host allocation failed");
        c->h = static_cast<char*>(hp);
      }
      chunks_.push_back(std::move(c));
      n_chunks_.store(chunks_.size(), std::memory_order_release);
    }
    cur_.store(next, std::memory_order_release);
  }
}
void Arena::reset() {
  epoch_.fetch_add(1, std::memory_order_relaxed);
  for (auto& c : chunks_) { c->top.store(0, std::memory_order_relaxed); c->mark = 0; }
  cur_.store(0, std::memory_order_release);
}
void Arena::upload() {
  epoch_.fetch_add(1, std::memory_order_relaxed);
  for (auto& c : chunks_) {
    const size_t top = std::min(c->top.load(std::memory_order_relaxed), c->cap);
    if (top > c->mark) {
      if (mot_memcpy_h2d(ctx_, c->d + c->mark, c->h + c->mark, top - c->mark) != MOT_OK) throw Error("arena upload failed");
      c->mark = top;
    }
  }
}
void Arena::download() {
  epoch_.fetch_add(1, std::memory_order_relaxed);
  for (auto& c : chunks_) {
    const size_t top = std::min(c->top.load(std::memory_order_relaxed), c->cap);
    if (top > c->mark) {
      if (mot_memcpy_d2h(ctx_, c->h + c->mark, c->d + c->mark, top - c->mark) != MOT_OK) throw Error("arena download failed");
      c->mark = top;
    }
  }
}
void Arena::clear_pending() {
  for (auto& c : chunks_) {
    const size_t top = std::min(c->top.load(std::memory_order_relaxed), c->cap);
    if (top > c->mark && mot_memset(ctx_, c->d + c->mark, 0, top - c->mark) != MOT_OK) throw Error("arena clear failed");
  }
}
size_t Arena::bytes_in_flight() const {
  size_t s = 0;
  for (const auto& c : chunks_) {
    const size_t top = std::min(c->top.load(std::memory_order_relaxed), c->cap);
    if (top > c->mark) s += top - c->mark;
  }
  return s;
}

// ---- Device --------------------------------------------------------------------------------
Device::Device(int device_index) : index(device_index) {
  int rc = mot_ctx_create(device_index, nullptr, &ctx);
  if (rc != MOT_OK)
    throw Error("motcpp_amd: no usable gfx950 (MI355X) device " + std::to_string(device_index) +
                " (mot_ctx_create=" + std::to_string(rc) + "); there is no CPU fallback");
  up = std::make_unique<Arena>(ctx, size_t(8) << 20, true);
  down = std::make_unique<Arena>(ctx, size_t(4) << 20, true);
  tmp = std::make_unique<Arena>(ctx, size_t(64) << 20, false);
  zdown = std::make_unique<Arena>(ctx, size_t(1) << 20, true);
  lists.resize(kMaxHostThreads);
}
Device::~Device() {
  up.reset();
  down.reset();
  tmp.reset();
  zdown.reset();
  if (ctx) mot_ctx_destroy(ctx);
}
std::shared_ptr<Device> Device::shared(int device_index) {
  // One Device per GPU for the life of the process (round 6; a weak cache before). A Device is a context, a stream, four arenas (~90 MB of device and
  // page-locked memory), the merged-round state of its host-lifecycle trackers and their worker team: an application that creates a tracker per
  // camera session would otherwise rebuild all of it whenever its last tracker went away — tests/test_gpu_error_isolation.py does so 80 times in a row,
  // and two of nineteen full GPU-suite runs ended inside those cycles in glibc's "double free or corruption (!prev)", only with the HIP runtime that
  // PyTorch ships loaded first, never under AddressSanitizer, gdb or an LD_PRELOADed handler (what was tried). The map is never
  // destroyed: HIP calls from static destructors run after the runtime's own teardown.
  static std::mutex m;
  static auto* cache = new std::map<int, std::shared_ptr<Device>>();
  std::lock_guard<std::mutex> g(m);
  auto& sp = (*cache)[device_index];
  if (!sp) sp = std::make_shared<Device>(device_index);
  return sp;
}
void Device::check(int rc, const char* what) {
  if (rc != MOT_OK) throw Error(std::string(what) + " failed: " + mot_ctx_last_error(ctx));
}
void Device::begin_frame() {
  up->reset();
  down->reset();
  tmp->reset();
  zdown->reset();
}
bool Device::TaskLists::empty() const {
  for (int k = 0; k < 3; ++k)
    if (!det[k].empty() || !kf_init[k].empty() || !kf_upd[k].empty() || !kf_pred[k].empty() || !kf_box[k].empty() || !kf_warp[k].empty() || !kf_predw[k].empty()) return false;
  return det[3].empty() && feat_set.empty() && feat_ema.empty() && feat_late.empty() && ss_nn.empty() && gate.empty() && ss_iou.empty() && hyb[0].empty() && hyb[1].empty() && hyb[2].empty() && hyb[3].empty() && hyb[4].empty() && boost[0].empty() && boost[1].empty() && boost[2].empty() && boost[3].empty() && boost[4].empty() && boost[5].empty() && ucmc[0].empty() && ucmc[1].empty() && ucmc[2].empty() && ucmc[3].empty() && ucmc[4].empty() && cos.empty() && dot.empty() && deep.empty() && iou.empty() && oc.empty() && lap.empty();
}
void Device::TaskLists::clear() {
  for (int k = 0; k < 3; ++k) { det[k].clear(); kf_init[k].clear(); kf_upd[k].clear(); kf_pred[k].clear(); kf_box[k].clear(); kf_warp[k].clear(); kf_predw[k].clear(); }
  det[3].clear(); feat_late.clear(); ss_nn.clear(); gate.clear(); ss_iou.clear();
  for (auto& v : ucmc) v.clear();
  for (auto& v : boost) v.clear();
  for (auto& v : hyb) v.clear();
  feat_set.clear(); feat_ema.clear(); cos.clear(); dot.clear(); deep.clear(); iou.clear(); oc.clear(); lap.clear();
  lap_geom = false;
  lap_assoc = false;
  lap_appearance = false;
}
namespace {
template <class T>
void move_back(std::vector<T>& dst, std::vector<T>& src) {
  dst.insert(dst.end(), src.begin(), src.end());
  src.clear();
}
}  // namespace
void Device::TaskLists::append(TaskLists& o) {
  for (int k = 0; k < 3; ++k) {
    move_back(det[k], o.det[k]); move_back(kf_init[k], o.kf_init[k]); move_back(kf_upd[k], o.kf_upd[k]);
    move_back(kf_pred[k], o.kf_pred[k]); move_back(kf_box[k], o.kf_box[k]); move_back(kf_warp[k], o.kf_warp[k]); move_back(kf_predw[k], o.kf_predw[k]);
  }
  move_back(det[3], o.det[3]);
  move_back(feat_set, o.feat_set); move_back(feat_ema, o.feat_ema); move_back(feat_late, o.feat_late); move_back(ss_nn, o.ss_nn); move_back(gate, o.gate); move_back(ss_iou, o.ss_iou); for (int k = 0; k < 5; ++k) move_back(ucmc[k], o.ucmc[k]); for (int k = 0; k < 6; ++k) move_back(boost[k], o.boost[k]); for (int k = 0; k < 5; ++k) move_back(hyb[k], o.hyb[k]); move_back(cos, o.cos); move_back(dot, o.dot); move_back(deep, o.deep); move_back(iou, o.iou);
  move_back(oc, o.oc); move_back(lap, o.lap);
  lap_geom = lap_geom || o.lap_geom;
  o.lap_geom = false;
  lap_assoc = lap_assoc || o.lap_assoc;
  o.lap_assoc = false;
  lap_appearance = lap_appearance || o.lap_appearance;
  o.lap_appearance = false;
}
Device::TaskLists& Device::q() {
  const int t = Team::worker_id();
  return lists[t < kMaxHostThreads ? t : 0];
}
bool Device::pending() const {
  for (const TaskLists& l : lists)
    if (!l.empty()) return true;
  return up->bytes_in_flight() > 0 || down->bytes_in_flight() > 0 || zdown->bytes_in_flight() > 0;
}

namespace {
template <class T>
const T* stage_tasks(Arena& up, const std::vector<T>& v) {
  if (v.empty()) return nullptr;
  Span<T> s = up.alloc<T>(v.size());
  std::memcpy(s.h, v.data(), sizeof(T) * v.size());
  return s.d;
}
}  // namespace

void Device::reset_stats() {
  for (FamilyStat& f : stats) f = FamilyStat();
}
void* Device::get_event() {
  if (!event_pool_.empty()) { void* e = event_pool_.back(); event_pool_.pop_back(); return e; }
  void* e = nullptr;
  check(mot_event_create(ctx, &e), "mot_event_create");
  return e;
}
void Device::time_begin(int family) {
  if (!profile) return;
  Timed t{family, get_event(), get_event()};
  check(mot_event_record(ctx, t.e0), "mot_event_record");
  timed_.push_back(t);
}
void Device::time_end() {
  if (!profile) return;
  check(mot_event_record(ctx, timed_.back().e1), "mot_event_record");
}

void Device::flush() {
  TaskLists& L = lists[0];
  for (size_t i = 1; i < lists.size(); ++i)
    if (!lists[i].empty()) L.append(lists[i]);
  auto& det = L.det;
  auto& kf_init = L.kf_init;
  auto& kf_upd = L.kf_upd;
  auto& kf_pred = L.kf_pred;
  auto& kf_box = L.kf_box;
  auto& kf_warp = L.kf_warp;
  auto& kf_predw = L.kf_predw;
  auto& feat_set = L.feat_set;
  auto& feat_ema = L.feat_ema;
  auto &cos = L.cos;
  auto &iou = L.iou;
  auto &oc = L.oc;
  auto &lap = L.lap;
  const bool lap_geom = L.lap_geom, lap_assoc = L.lap_assoc, lap_appearance = L.lap_appearance;
  const mot_det_task* d_det[4];
  d_det[3] = stage_tasks(*up, L.det[3]);
  const mot_kf_task *d_init[3], *d_upd[3], *d_pred[3], *d_box[3], *d_warp[3], *d_predw[3];
  for (int k = 0; k < 3; ++k) {
    d_det[k] = stage_tasks(*up, det[k]);
    d_init[k] = stage_tasks(*up, kf_init[k]);
    d_upd[k] = stage_tasks(*up, kf_upd[k]);
    d_pred[k] = stage_tasks(*up, kf_pred[k]);
    d_box[k] = stage_tasks(*up, kf_box[k]);
    d_warp[k] = stage_tasks(*up, kf_warp[k]);
    d_predw[k] = stage_tasks(*up, kf_predw[k]);
  }
  const mot_feat_task* d_fset = stage_tasks(*up, feat_set);
  const mot_feat_task* d_fema = stage_tasks(*up, feat_ema);
  const mot_feat_task* d_flate = stage_tasks(*up, L.feat_late);
  const mot_ss_nn_task* d_ssnn = stage_tasks(*up, L.ss_nn);
  const mot_gate_task* d_gate = stage_tasks(*up, L.gate);
  const mot_ss_iou_task* d_ssiou = stage_tasks(*up, L.ss_iou);
  const mot_hyb_task* d_hyb[5];
  for (int k = 0; k < 5; ++k) d_hyb[k] = stage_tasks(*up, L.hyb[k]);
  const mot_boost_task* d_boost[6];
  for (int k = 0; k < 6; ++k) d_boost[k] = stage_tasks(*up, L.boost[k]);
  const mot_ucmc_task* d_ucmc[5];
  for (int k = 0; k < 5; ++k) d_ucmc[k] = stage_tasks(*up, L.ucmc[k]);
  const mot_cos_task* d_cos = stage_tasks(*up, cos);
  const mot_cos_task* d_dot = stage_tasks(*up, L.dot);
  const mot_deep_task* d_deep = stage_tasks(*up, L.deep);
  const mot_iou_task* d_iou = stage_tasks(*up, iou);
  const mot_ocsort_task* d_oc = stage_tasks(*up, oc);
  const mot_lap_task* d_lap = stage_tasks(*up, lap);
  up->upload();
  zdown->clear_pending();

  auto maxn = [](const auto& v, auto get) { int m = 0; for (const auto& t : v) m = std::max(m, get(t)); return m; };
  // One launch per non-empty kernel family; with profile on, each launch is bracketed by an event pair and its
  // algorithmic bytes (DESIGN.md "Kernels") are accumulated for the roofline report.
  auto run = [&](int family, size_t ntasks, double bytes, double flops, auto&& launch) {
    if (ntasks == 0) return;
    time_begin(family);
    launch();
    time_end();
    ++counters.launches;
    FamilyStat& f = stats[family];
    ++f.launches; f.tasks += static_cast<long>(ntasks); f.bytes += bytes; f.flops += flops;
  };
  auto kf_bytes = [](const std::vector<mot_kf_task>& v, int kind, double per_state_rw) {
    const double D = mot_kf_dim(kind);
    double b = 0;
    for (const mot_kf_task& t : v) b += t.n * (4.0 * (D + D * D) * per_state_rw + 16.0 + 8.0);
    return b;
  };
  for (int k = 0; k < 4; ++k) {
    double b = 0;
    for (const mot_det_task& t : det[k]) b += t.n * (16.0 + 32.0);
    run(F_DET, det[k].size(), b, 0, [&] { check(mot_det_prepare(ctx, k, d_det[k], (int)det[k].size(), maxn(det[k], [](const mot_det_task& t) { return t.n; })), "mot_det_prepare"); });
  }
  {
    double b = 0;
    for (const mot_feat_task& t : feat_set) b += 4.0 * t.n * t.d * 2;
    run(F_FEAT, feat_set.size(), b, 0, [&] { check(mot_feat_update(ctx, d_fset, (int)feat_set.size(), maxn(feat_set, [](const mot_feat_task& t) { return t.n; })), "mot_feat_update"); });
  }
  auto run_ucmc = [&](int op, int family) {  // UCMCTrack (ucmc.cpp): 20 doubles of state per track, 6 per mapped detection
    auto& v = L.ucmc[op];
    double b = 0;
    for (const mot_ucmc_task& t : v) b += (op == MOT_UCMC_COST) ? 4.0 * t.n * (double)t.m + 80.0 * t.n + 48.0 * t.m : (op == MOT_UCMC_MAP ? 64.0 * t.n : 368.0 * t.n);
    run(family, v.size(), b, 0, [&] { check(mot_ucmc_run(ctx, op, d_ucmc[op], (int)v.size(), maxn(v, [](const mot_ucmc_task& t) { return t.n; }), maxn(v, [](const mot_ucmc_task& t) { return t.m; })), "mot_ucmc_run"); });
  };
  auto run_boost = [&](int op, int family) {  // BoostTrack (boosttrack.cpp): 288-byte records
    auto& v = L.boost[op];
    double b = 0;
    for (const mot_boost_task& t : v)
      b += (op == MOT_BOOST_COST) ? 4.0 * t.n * (double)t.m + 16.0 * t.n + 48.0 * t.m : (op == MOT_BOOST_DLO ? 24.0 * t.n + 20.0 * t.m : (op == MOT_BOOST_BOXES ? 32.0 * t.n : 592.0 * t.n));
    run(family, v.size(), b, 0, [&] { check(mot_boost_run(ctx, op, d_boost[op], (int)v.size(), maxn(v, [](const mot_boost_task& t) { return t.n; }), maxn(v, [](const mot_boost_task& t) { return t.m; })), "mot_boost_run"); });
  };
  auto run_hyb = [&](int op, int family) {  // HybridSORT (hybridsort.cpp): 360-byte records
    auto& v = L.hyb[op];
    double b = 0;
    for (const mot_hyb_task& t : v) b += (op == MOT_HYB_PAIR) ? 8.0 * t.n * (double)t.m + 20.0 * (t.n + t.m) : (op == MOT_HYB_BOXES ? 36.0 * t.n : 740.0 * t.n);
    run(family, v.size(), b, 0, [&] { check(mot_hyb_run(ctx, op, d_hyb[op], (int)v.size(), maxn(v, [](const mot_hyb_task& t) { return t.n; }), maxn(v, [](const mot_hyb_task& t) { return t.m; })), "mot_hyb_run"); });
  };
  run_hyb(MOT_HYB_UPDATE, F_KF_UPDATE);
  run_hyb(MOT_HYB_INIT, F_KF_INIT);
  run_boost(MOT_BOOST_INIT, F_KF_INIT);
  run_boost(MOT_BOOST_UPDATE, F_KF_UPDATE);
  run_ucmc(MOT_UCMC_MAP, F_DET);
  run_ucmc(MOT_UCMC_INIT, F_KF_INIT);
  run_ucmc(MOT_UCMC_UPDATE, F_KF_UPDATE);
  for (int k = 0; k < 3; ++k)
    run(F_KF_INIT, kf_init[k].size(), kf_bytes(kf_init[k], k, 1.0), 0, [&] { check(mot_kf_initiate(ctx, k, d_init[k], (int)kf_init[k].size(), maxn(kf_init[k], [](const mot_kf_task& t) { return t.n; })), "mot_kf_initiate"); });
  for (int k = 0; k < 3; ++k)
    run(F_KF_UPDATE, kf_upd[k].size(), kf_bytes(kf_upd[k], k, 2.0), 0, [&] { check(mot_kf_update(ctx, k, d_upd[k], (int)kf_upd[k].size(), maxn(kf_upd[k], [](const mot_kf_task& t) { return t.n; })), "mot_kf_update"); });
  {
    double b = 0;
    for (const mot_feat_task& t : feat_ema) b += 4.0 * t.n * t.d * 3;
    run(F_FEAT, feat_ema.size(), b, 0, [&] { check(mot_feat_update(ctx, d_fema, (int)feat_ema.size(), maxn(feat_ema, [](const mot_feat_task& t) { return t.n; })), "mot_feat_update"); });
  }
  {
    auto& fl = L.feat_late;
    double b = 0;
    for (const mot_feat_task& t : fl) b += 4.0 * t.n * t.d * 2;
    run(F_FEAT, fl.size(), b, 0, [&] { check(mot_feat_update(ctx, d_flate, (int)fl.size(), maxn(fl, [](const mot_feat_task& t) { return t.n; })), "mot_feat_update"); });
  }
  for (int k = 0; k < 3; ++k)
    run(F_KF_PREDICT, kf_pred[k].size(), kf_bytes(kf_pred[k], k, 2.0), 0, [&] { check(mot_kf_predict(ctx, k, d_pred[k], (int)kf_pred[k].size(), maxn(kf_pred[k], [](const mot_kf_task& t) { return t.n; })), "mot_kf_predict"); });
  run_ucmc(MOT_UCMC_PREDICT, F_KF_PREDICT);
  run_boost(MOT_BOOST_PREDICT, F_KF_PREDICT);
  run_hyb(MOT_HYB_PREDICT, F_KF_PREDICT);
  run_hyb(MOT_HYB_BOXES, F_KF_BOXES);
  run_boost(MOT_BOOST_DLO, F_IOU);
  run_boost(MOT_BOOST_BOXES, F_KF_BOXES);
  for (int k = 0; k < 3; ++k)  // predict + camera-motion warp in one pass over the states
    run(F_KF_PREDICT, kf_predw[k].size(), kf_bytes(kf_predw[k], k, 2.0), 0, [&] { check(mot_kf_predict_warp(ctx, k, d_predw[k], (int)kf_predw[k].size(), maxn(kf_predw[k], [](const mot_kf_task& t) { return t.n; })), "mot_kf_predict_warp"); });
  for (int k = 0; k < 3; ++k)  // camera-motion warp of states that are not predicted this frame (their slots are disjoint from the predicted ones)
    run(F_KF_PREDICT, kf_warp[k].size(), kf_bytes(kf_warp[k], k, 2.0), 0, [&] { check(mot_kf_warp(ctx, k, d_warp[k], (int)kf_warp[k].size(), maxn(kf_warp[k], [](const mot_kf_task& t) { return t.n; })), "mot_kf_warp"); });
  for (int k = 0; k < 3; ++k) {
    double b = 0;
    for (const mot_kf_task& t : kf_box[k]) b += t.n * (16.0 + 16.0 + 4.0);
    run(F_KF_BOXES, kf_box[k].size(), b, 0, [&] { check(mot_kf_boxes(ctx, k, d_box[k], (int)kf_box[k].size(), maxn(kf_box[k], [](const mot_kf_task& t) { return t.n; })), "mot_kf_boxes"); });
  }
  {
    double b = 0, fl = 0;
    for (const mot_cos_task& t : cos) { b += 4.0 * ((double)(t.n + t.m) * t.d + (double)t.n * t.m); fl += 2.0 * t.n * (double)t.m * t.d; }
    run(F_COSINE, cos.size(), b, fl, [&] {
      check(mot_cosine_cost(ctx, d_cos, (int)cos.size(), maxn(cos, [](const mot_cos_task& t) { return t.n; }), maxn(cos, [](const mot_cos_task& t) { return t.m; })), "mot_cosine_cost");
      ++counters.launches;
    });
  }
  {
    auto& dot = L.dot;
    double b = 0, fl = 0;
    for (const mot_cos_task& t : dot) { b += 4.0 * ((double)(t.n + t.m) * t.d + (double)t.n * t.m); fl += 2.0 * t.n * (double)t.m * t.d; }
    run(F_COSINE, dot.size(), b, fl, [&] {
      check(mot_embedding_cost(ctx, MOT_EMB_DOT, d_dot, (int)dot.size(), maxn(dot, [](const mot_cos_task& t) { return t.n; }), maxn(dot, [](const mot_cos_task& t) { return t.m; })), "mot_embedding_cost");
    });
  }
  {  // StrongSORT: nearest-sample distances, then the motion gate on them, and its IoU cost (strongsort.cpp:239-275, 449-492, 500-583)
    auto& nn = L.ss_nn;
    double b = 0;
    for (const mot_ss_nn_task& t : nn) b += 4.0 * ((double)(t.soff ? 1 : 0) + 1.0) * t.n * (double)t.m;
    run(F_COSINE, nn.size(), b, 0, [&] { check(mot_ss_nn_cost(ctx, d_ssnn, (int)nn.size(), maxn(nn, [](const mot_ss_nn_task& t) { return t.n; }), maxn(nn, [](const mot_ss_nn_task& t) { return t.m; })), "mot_ss_nn_cost"); });
    auto& gt = L.gate;
    b = 0;
    for (const mot_gate_task& t : gt) b += 8.0 * t.n * (double)t.m;
    run(F_IOU, gt.size(), b, 0, [&] { check(mot_gate_cost(ctx, MOT_KF_XYAH, d_gate, (int)gt.size(), maxn(gt, [](const mot_gate_task& t) { return t.n; }), maxn(gt, [](const mot_gate_task& t) { return t.m; })), "mot_gate_cost"); });
    auto& si = L.ss_iou;
    b = 0;
    for (const mot_ss_iou_task& t : si) b += 4.0 * t.n * (double)t.m + 16.0 * (t.n + t.m);
    run(F_IOU, si.size(), b, 0, [&] { check(mot_ss_iou_cost(ctx, d_ssiou, (int)si.size(), maxn(si, [](const mot_ss_iou_task& t) { return t.n; }), maxn(si, [](const mot_ss_iou_task& t) { return t.m; })), "mot_ss_iou_cost"); });
  }
  {
    double b = 0;
    for (const mot_iou_task& t : iou) b += 16.0 * (t.n + t.m) + (t.cost ? 4.0 * t.n * (double)t.m : 0.0) + (t.emb ? 4.0 * t.n * (double)t.m : 0.0);
    run(F_IOU, iou.size(), b, 0, [&] { check(mot_iou_cost_ex(ctx, d_iou, (int)iou.size(), maxn(iou, [](const mot_iou_task& t) { return t.n; }), maxn(iou, [](const mot_iou_task& t) { return t.m; }), maxn(iou, [](const mot_iou_task& t) { return t.assoc; }) == 0 ? MOT_COST_F_IOU_ONLY : 0), "mot_iou_cost"); });
  }
  {
    double b = 0;
    for (const mot_ocsort_task& t : oc) b += 20.0 * t.nd + 44.0 * t.nt + 8.0 * t.nd * (double)t.nt;
    run(F_OCSORT, oc.size(), b, 0, [&] { check(mot_ocsort_cost_ex(ctx, d_oc, (int)oc.size(), maxn(oc, [](const mot_ocsort_task& t) { return t.nd; }), maxn(oc, [](const mot_ocsort_task& t) { return t.nt; }), maxn(oc, [](const mot_ocsort_task& t) { return t.assoc; }) == 0 ? MOT_COST_F_IOU_ONLY : 0), "mot_ocsort_cost"); });
  }
  {
    auto& deep = L.deep;
    double b = 0;
    for (const mot_deep_task& t : deep) b += 16.0 * t.nd * (double)t.nt;
    run(F_OCSORT, deep.size(), b, 0, [&] {
      check(mot_deepoc_cost(ctx, d_deep, (int)deep.size(), maxn(deep, [](const mot_deep_task& t) { return t.nd; }), maxn(deep, [](const mot_deep_task& t) { return t.nt; })), "mot_deepoc_cost");
      ++counters.launches;
    });
  }
  run_ucmc(MOT_UCMC_COST, F_IOU);
  run_boost(MOT_BOOST_COST, F_IOU);
  run_hyb(MOT_HYB_PAIR, F_IOU);
  {
    double b = 0;
    for (const mot_lap_task& t : lap)
      b += (t.geom.a ? 20.0 * (t.n + t.m) : 4.0 * t.n * (double)t.m) + (t.iou ? 4.0 * t.n * (double)t.m : 0.0) + 4.0 * (t.n + t.m);
    run(F_LAP, lap.size(), b, 0, [&] { check(mot_lap_solve(ctx, d_lap, (int)lap.size(), maxn(lap, [](const mot_lap_task& t) { return t.n; }), maxn(lap, [](const mot_lap_task& t) { return t.m; }), (lap_geom ? MOT_LAP_F_GEOM : 0) | (lap_assoc ? MOT_LAP_F_ASSOC : 0) | (lap_appearance ? 0 : MOT_LAP_F_PLAIN)), "mot_lap_solve"); });
  }
  down->download();
  zdown->download();
  {
    const auto w0 = std::chrono::steady_clock::now();
    check(mot_ctx_sync(ctx), "mot_ctx_sync");
    counters.ms_sync_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
  for (const Timed& t : timed_) {
    float ms = 0.f;
    check(mot_event_elapsed(ctx, t.e0, t.e1, &ms), "mot_event_elapsed");
    stats[t.family].ms += ms;
    event_pool_.push_back(t.e0);
    event_pool_.push_back(t.e1);
  }
  timed_.clear();
  ++counters.flushes;
  L.clear();
}

// ---- Core ----------------------------------------------------------------------------------
Core::Core(std::shared_ptr<Device> dev, int kf_kind) : dev_(std::move(dev)), kind_(kf_kind), D_(mot_kf_dim(kf_kind)) {}
Core::~Core() {
  if (mean_) mot_free(dev_->ctx, mean_);
}
// The slab is an array of records (mean[D] then the D x D covariance, mot_kf_task's layout): growing it is one copy.
void Core::grow(int pcap, int scap) {
  const int ncap = pcap + scap;
  const size_t rec = static_cast<size_t>(D_) * (D_ + 1);
  void* nm = nullptr;
  dev_->check(mot_malloc(dev_->ctx, sizeof(float) * rec * ncap, &nm), "slab alloc");
  if (mean_ && next_ > 0) {
    dev_->check(mot_memcpy_d2d(dev_->ctx, nm, mean_, sizeof(float) * rec * next_), "slab copy");
    dev_->check(mot_ctx_sync(dev_->ctx), "slab copy sync");
  }
  if (mean_) mot_free(dev_->ctx, mean_);
  mean_ = static_cast<float*>(nm);
  cov_ = mean_ + D_;
  pcap_ = pcap; scap_ = scap; cap_ = ncap;
}
void Core::reserve(int extra_persistent, int scratch) {
  const int avail = (pcap_ - next_) + static_cast<int>(free_.size());
  int pcap = pcap_, scap = scap_;
  // growth is a slab re-allocation + plane copies + a sync: start generous and double, so it happens O(1) times per stream
  if (avail < extra_persistent) pcap = std::max(pcap_ * 2, round_up(2 * (next_ + extra_persistent) + 64, 64));
  if (scap_ < scratch) scap = std::max(scap_ * 2, round_up(2 * scratch + 64, 64));
  if (pcap != pcap_ || scap != scap_ || !mean_) grow(std::max(pcap, 512), std::max(scap, 512));
}
int Core::new_slot() {
  if (!free_.empty()) { int s = free_.back(); free_.pop_back(); return s; }
  if (next_ >= pcap_) throw Error("Core::new_slot: slab exhausted (reserve() not called with enough headroom)");
  return next_++;
}
void Core::release_slot(int s) { free_.push_back(s); }
void Core::clear_slots() { free_.clear(); next_ = 0; }

Core::Dets Core::upload_dets(const float* colmajor, int n, int ld, int det_kind, const float* resident, int resident_ld) {
  Dets d;
  d.n = n;
  if (n <= 0) return d;
  if (resident) {
    d.d_raw = resident;
    d.ld_raw = resident_ld;
  } else {
    Span<float> raw = dev_->up->alloc<float>(static_cast<size_t>(6) * n);
    for (int k = 0; k < 6; ++k) std::memcpy(raw.h + static_cast<size_t>(k) * n, colmajor + static_cast<size_t>(k) * ld, sizeof(float) * n);
    d.d_raw = raw.d;
    d.ld_raw = n;
  }
  d.d_box = dev_->tmp->alloc<float>(static_cast<size_t>(4) * n).d;
  d.d_meas = dev_->tmp->alloc<float>(static_cast<size_t>(4) * n).d;
  mot_det_task t{};
  t.dets = d.d_raw; t.ld = d.ld_raw; t.n = n; t.box = d.d_box; t.ldb = n; t.meas = d.d_meas; t.ldm = n;
  dev_->q().det[det_kind].push_back(t);
  return d;
}
Span<int32_t> Core::ints(const std::vector<int>& v) {
  Span<int32_t> s = dev_->up->alloc<int32_t>(v.size());
  if (!v.empty()) std::memcpy(s.h, v.data(), sizeof(int32_t) * v.size());
  return s;
}
Span<uint8_t> Core::bytes(const std::vector<uint8_t>& v) {
  Span<uint8_t> s = dev_->up->alloc<uint8_t>(v.size());
  if (!v.empty()) std::memcpy(s.h, v.data(), v.size());
  return s;
}
Span<float> Core::floats(const std::vector<float>& v) {
  Span<float> s = dev_->up->alloc<float>(v.size());
  if (!v.empty()) std::memcpy(s.h, v.data(), sizeof(float) * v.size());
  return s;
}

float* Core::predict(const std::vector<int>& src, const std::vector<int>* dst, const std::vector<uint8_t>* flags, Span<float>* boxes_dl,
                     const float* warp9) {
  const int n = static_cast<int>(src.size());
  if (n == 0) return nullptr;
  Span<int32_t> s = ints(src), d;
  if (dst) d = ints(*dst);
  Span<uint8_t> f;
  if (flags) f = bytes(*flags);
  float* bx;
  if (boxes_dl) { *boxes_dl = dev_->down->alloc<float>(static_cast<size_t>(4) * n); bx = boxes_dl->d; }
  else bx = dev_->tmp->alloc<float>(static_cast<size_t>(4) * n).d;
  mot_kf_task t{};
  t.mean = mean_; t.cov = cov_; t.cap = cap_; t.n = n; t.src = s.d; t.dst = dst ? d.d : nullptr; t.flags = flags ? f.d : nullptr;
  t.boxes = bx; t.ldb = n; t.q[0] = q[0]; t.q[1] = q[1]; t.q[2] = q[2];
  if (warp9) { std::memcpy(t.warp, warp9, sizeof(t.warp)); dev_->q().kf_predw[kind_].push_back(t); }
  else dev_->q().kf_pred[kind_].push_back(t);
  return bx;
}
void Core::warp(const std::vector<int>& slots, const float* warp9) {
  const int n = static_cast<int>(slots.size());
  if (n == 0) return;
  Span<int32_t> s = ints(slots);
  mot_kf_task t{};
  t.mean = mean_; t.cov = cov_; t.cap = cap_; t.n = n; t.src = s.d;
  std::memcpy(t.warp, warp9, sizeof(t.warp));
  dev_->q().kf_warp[kind_].push_back(t);
}
float* Core::boxes(const std::vector<int>& slots, Span<float>* boxes_dl) {
  const int n = static_cast<int>(slots.size());
  if (n == 0) return nullptr;
  Span<int32_t> s = ints(slots);
  float* bx;
  if (boxes_dl) { *boxes_dl = dev_->down->alloc<float>(static_cast<size_t>(4) * n); bx = boxes_dl->d; }
  else bx = dev_->tmp->alloc<float>(static_cast<size_t>(4) * n).d;
  mot_kf_task t{};
  t.mean = mean_; t.cov = cov_; t.cap = cap_; t.n = n; t.src = s.d; t.boxes = bx; t.ldb = n;
  t.reserved = box_style;
  dev_->q().kf_box[kind_].push_back(t);
  return bx;
}
void Core::update(const std::vector<int>& src, const std::vector<int>& dst, const std::vector<int>& midx, const Dets& dets) {
  const int n = static_cast<int>(src.size());
  if (n == 0) return;
  Span<int32_t> s = ints(src), d = ints(dst), m = ints(midx);
  mot_kf_task t{};
  t.mean = mean_; t.cov = cov_; t.cap = cap_; t.n = n; t.src = s.d; t.dst = d.d; t.meas = dets.d_meas; t.ldm = dets.n; t.midx = m.d;
  t.q[0] = q[0]; t.q[1] = q[1]; t.q[2] = q[2];
  dev_->q().kf_upd[kind_].push_back(t);
}
void Core::initiate(const std::vector<int>& dst, const std::vector<int>& midx, const Dets& dets) {
  const int n = static_cast<int>(dst.size());
  if (n == 0) return;
  Span<int32_t> d = ints(dst), m = ints(midx);
  mot_kf_task t{};
  t.mean = mean_; t.cov = cov_; t.cap = cap_; t.n = n; t.src = d.d; t.dst = d.d; t.meas = dets.d_meas; t.ldm = dets.n; t.midx = m.d;
  dev_->q().kf_init[kind_].push_back(t);
}
float* Core::iou_cost(const IouArgs& a, int* ldc) {
  const int ld = round_up(std::max(a.m, 1), 4);
  float* cost = dev_->tmp->alloc<float>(static_cast<size_t>(std::max(a.n, 1)) * ld).d;
  mot_iou_task t{};
  t.n = a.n; t.m = a.m; t.a = a.a; t.lda = a.lda; t.aidx = a.aidx; t.b = a.b; t.ldb = a.ldb; t.bidx = a.bidx; t.bconf = a.bconf;
  t.cost = cost; t.ldc = ld; t.mode = a.mode; t.emb = a.emb; t.lde = a.lde; t.prox_thresh = a.prox; t.app_thresh = a.app; t.fuse = a.fuse;
  t.assoc = a.assoc; t.frame_diag = a.frame_diag;
  dev_->q().iou.push_back(t);
  *ldc = ld;
  return cost;
}
Core::Lap Core::lap(const float* cost, int ldc, int n, int m, float thresh, int mode, const float* iou, int ldi, float gate, bool want_xval) {
  Lap r;
  r.n = n; r.m = m;
  r.x = dev_->down->alloc<int32_t>(std::max(n, 1));
  r.y = dev_->down->alloc<int32_t>(std::max(m, 1));
  r.info = dev_->down->alloc<int32_t>(4);
  if (want_xval) r.xval = dev_->down->alloc<float>(std::max(n, 1));
  mot_lap_task t{};
  t.n = n; t.m = m; t.cost = cost; t.ldc = ldc; t.thresh = thresh; t.x = r.x.d; t.y = r.y.d; t.mode = mode; t.iou = iou; t.ldi = ldi; t.gate = gate;
  t.xval = want_xval ? r.xval.d : nullptr; t.info = r.info.d;
  t.work = dev_->tmp->alloc<char>(mot_lap_work_bytes(n, m)).d;
  if (static_cast<long long>(n) * m >= 16384) t.rowlist = dev_->tmp->alloc<char>(mot_lap_rowlist_bytes(n)).d;
  dev_->q().lap.push_back(t);
  r.queued = true;
  return r;
}

Core::Lap Core::lap_geom(const IouArgs& a, float thresh, int mode, float gate, bool want_xval) {
  Lap r;
  r.n = a.n; r.m = a.m;
  r.x = dev_->down->alloc<int32_t>(std::max(a.n, 1));
  r.y = dev_->down->alloc<int32_t>(std::max(a.m, 1));
  r.info = dev_->down->alloc<int32_t>(4);
  if (want_xval) r.xval = dev_->down->alloc<float>(std::max(a.n, 1));
  mot_lap_task t{};
  t.n = a.n; t.m = a.m; t.thresh = thresh; t.x = r.x.d; t.y = r.y.d; t.mode = mode; t.gate = gate;
  t.xval = want_xval ? r.xval.d : nullptr; t.info = r.info.d;
  t.work = dev_->tmp->alloc<char>(mot_lap_work_bytes(a.n, a.m)).d;
  t.geom.n = a.n; t.geom.m = a.m; t.geom.a = a.a; t.geom.lda = a.lda; t.geom.aidx = a.aidx; t.geom.b = a.b; t.geom.ldb = a.ldb;
  t.geom.bidx = a.bidx; t.geom.bconf = a.bconf; t.geom.mode = a.mode; t.geom.emb = a.emb; t.geom.lde = a.lde;
  t.geom.prox_thresh = a.prox; t.geom.app_thresh = a.app; t.geom.fuse = a.fuse;
  t.geom.assoc = a.assoc; t.geom.frame_diag = a.frame_diag;
  dev_->q().lap.push_back(t);
  dev_->q().lap_geom = true;
  if (a.assoc != MOT_ASSOC_IOU) dev_->q().lap_assoc = true;
  if (a.mode == MOT_COST_BOTSORT) dev_->q().lap_appearance = true;
  r.queued = true;
  return r;
}

}  // namespace motcpp::rt
