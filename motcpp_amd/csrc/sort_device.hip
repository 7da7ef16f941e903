// SORT with the per-stream lifecycle ON THE DEVICE (reference: src/trackers/sort.cpp:102-255), same construction as
// bt_device.hip: track records indexed by Kalman slot, the track list an array of slots in the reference's order, every
// "for ... push_back" an order-preserving wavefront compaction, a frame a fixed sequence of launches for all streams:
//   sort_begin -> det_prepare, kf_predict (in place) -> sort_assoc (NaN rule) -> lap -> sort_apply -> kf_initiate,
//   kf_update, kf_boxes -> sort_emit, then one copy of the output tables.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "lifecycle_common.hpp"

namespace {
using mot::lifecycle::compact;
using mot::lifecycle::kW;
using mot::lifecycle::FrameDev;

struct SortParams {
  float det_thresh, iou_thr;
  int max_age, min_hits;
};

struct SortStream {
  // persistent
  int frame_count, next_id, next_slot, n_free, n_trk, cur, err;
  int* free_stack;
  int* trk[2];  // slots in list order, ping-pong
  int *t_id, *t_cls, *t_det, *t_hits, *t_tsu, *t_age;
  float* t_conf;
  // frame
  const float* dets; int ld, n;
  int skip;  // the stream sits this frame out (pooled form: counts[s] < 0)
  int* valid; int n_valid;
  int* keep; int n_keep;      // positions (into the predicted-box planes) of the tracks that survive the NaN rule
  int *x, *y;
  int *upd_slot, *upd_meas; int n_upd;
  int *init_slot, *init_meas; int n_init;
  int* out_slot; int n_out;
  float* pbox;  // [4][CAP] predicted boxes, column = position in the track list at frame start
  float* obox;  // [4][CAP] (round 5: unused — sort_emit computes a row's box from the updated state)
  float* kmean; // this stream's Kalman records (56 floats per slot: 7 + 7 x 7)
};


// detections with conf >= det_thresh (:112-120), ++age / ++time_since_update of every track (SortTrack::predict :43-51)
__global__ void __launch_bounds__(kW) sort_begin(SortStream* streams, SortParams P, int CAP, int D, FrameDev FD, const float* dets_base,
                                                  mot_det_task* det_t, mot_kf_task* pred_t) {
  SortStream& S = streams[blockIdx.x];
  const int t = threadIdx.x;
  const int n = FD.counts[blockIdx.x];
  if (n < 0) {  // not this stream's frame
    if (t == 0) { S.skip = 1; det_t[blockIdx.x].n = 0; pred_t[blockIdx.x].n = 0; }
    return;
  }
  int ldd = D;
  const float* dets = mot::lifecycle::frame_dets(FD, dets_base, blockIdx.x, D, ldd);
  const float* conf = dets + static_cast<size_t>(4) * ldd;
  float* dbox = det_t[blockIdx.x].box;
  float* dmeas = det_t[blockIdx.x].meas;
  const int dldb = det_t[blockIdx.x].ldb, dldm = det_t[blockIdx.x].ldm;
  int nv = 0;
  for (int i0 = 0; i0 < n; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < n && conf[i] >= P.det_thresh;
    const int p = compact(v, nv);
    if (v) S.valid[p] = i;
    if (i < n && n <= D) {  // the detection's box and measurement (round 5: here instead of det_kernel<MOT_DET_XYSR>, ops.hpp:188-197: the same operations)
      const float x1 = dets[i], y1 = dets[static_cast<size_t>(ldd) + i], x2 = dets[static_cast<size_t>(2) * ldd + i], y2 = dets[static_cast<size_t>(3) * ldd + i];
      const float w = x2 - x1, h = y2 - y1;
      const float zz[4] = {x1 + w * 0.5f, y1 + h * 0.5f, w * h, (h > 1e-6f) ? (w / h) : 0.0f};
      const float bb[4] = {x1, y1, x2, y2};
#pragma unroll
      for (int q = 0; q < 4; ++q) { dbox[static_cast<size_t>(q) * dldb + i] = bb[q]; dmeas[static_cast<size_t>(q) * dldm + i] = zz[q]; }
    }
  }
  const int* trk = S.trk[S.cur];
  for (int i = t; i < S.n_trk; i += kW) { const int slot = trk[i]; S.t_age[slot] += 1; S.t_tsu[slot] += 1; }
  if (t == 0) {
    S.frame_count += 1;
    S.dets = dets; S.ld = ldd; S.n = n; S.n_valid = nv; S.skip = 0;
    if (n > D) S.err = 1;
    det_t[blockIdx.x].dets = dets; det_t[blockIdx.x].ld = ldd; det_t[blockIdx.x].n = (n <= D) ? n : 0;
    pred_t[blockIdx.x].n = S.n_trk; pred_t[blockIdx.x].src = trk;
  }
}

// NaN rule (:132-150): tracks whose predicted box has a NaN are dropped, the survivors are associated
__global__ void __launch_bounds__(kW) sort_assoc(SortStream* streams, int CAP, mot_lap_task* lap_t, unsigned long long* stats) {
  SortStream& S = streams[blockIdx.x];
  const int t = threadIdx.x;
  if (S.skip) {
    if (t == 0) { mot_lap_task& L = lap_t[blockIdx.x]; L.n = 0; L.m = 0; L.geom.n = 0; L.geom.m = 0; }
    return;
  }
  const int nt = S.n_trk;
  const int* trk = S.trk[S.cur];
  int* kept = S.trk[S.cur ^ 1];
  int nk = 0, free_top = S.n_free;
  for (int i0 = 0; i0 < nt; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < nt;
    float s = 0.f;
    if (v) s = S.pbox[i] + S.pbox[static_cast<size_t>(CAP) + i] + S.pbox[static_cast<size_t>(2) * CAP + i] + S.pbox[static_cast<size_t>(3) * CAP + i];
    const bool ok = v && !(s != s);
    const int slot = v ? trk[i] : 0;
    const int p = compact(ok, nk);
    if (ok) { kept[p] = slot; S.keep[p] = i; }
    const bool dead = v && !ok;
    const int pf = compact(dead, free_top);
    if (dead) S.free_stack[pf] = slot;
  }
  if (t == 0) {
    S.n_trk = nk; S.cur ^= 1; S.n_keep = nk; S.n_free = free_top;
    mot_lap_task& L = lap_t[blockIdx.x];
    const bool q = nk > 0 && S.n_valid > 0;
    L.n = q ? nk : 0; L.m = q ? S.n_valid : 0;
    L.geom.n = L.n; L.geom.m = L.m;
    if (stats && q) { atomicAdd(&stats[(blockIdx.x & 63) * 2], 1ull); atomicAdd(&stats[(blockIdx.x & 63) * 2 + 1], static_cast<unsigned long long>(nk + S.n_valid)); }
  }
}

// matches -> SortTrack::update (:53-70), unmatched detections -> new trackers (:196-204), deaths (:208-216), rows to emit
__global__ void __launch_bounds__(kW) sort_apply(SortStream* streams, SortParams P, int CAP, mot_kf_task* init_t, mot_kf_task* upd_t,
                                                  mot_kf_task* box_t) {
  SortStream& S = streams[blockIdx.x];
  const int t = threadIdx.x;
  if (S.skip) {
    if (t == 0) { init_t[blockIdx.x].n = 0; upd_t[blockIdx.x].n = 0; box_t[blockIdx.x].n = 0; }
    return;
  }
  const int nt = S.n_trk, nd = S.n_valid;
  const bool have = nt > 0 && nd > 0;
  const int* trk = S.trk[S.cur];
  int* next = S.trk[S.cur ^ 1];
  int n_upd = 0;
  for (int i0 = 0; i0 < nt; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < nt;
    const int x = (v && have) ? S.x[i] : -1;
    const bool m = v && x >= 0;
    const int slot = v ? trk[i] : 0;
    const int p = compact(m, n_upd);
    if (m) {
      const int det = S.valid[x];
      S.t_conf[slot] = S.dets[static_cast<size_t>(4) * S.ld + det];
      S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
      S.t_det[slot] = det;
      S.t_hits[slot] += 1; S.t_tsu[slot] = 0;
      S.upd_slot[p] = slot; S.upd_meas[p] = det;
    }
  }
  __syncthreads();
  // survivors of the age rule, in order (new tracks have time_since_update 0 and always stay)
  int n_next = 0, free_top = S.n_free;
  for (int i0 = 0; i0 < nt; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < nt;
    const int slot = v ? trk[i] : 0;
    const bool k = v && S.t_tsu[slot] <= P.max_age;
    const int p = compact(k, n_next);
    if (k) next[p] = slot;
    const bool dead = v && !k;
    const int pf = compact(dead, free_top);
    if (dead) S.free_stack[pf] = slot;
  }
  // births, ids in detection order. Slots: the free stack as it was at kernel entry first (the slots freed just above sit
  // on top of it and are left alone: a dying track's state must not be overwritten before it is gone), then fresh ones.
  int n_init = 0, err = 0;
  int avail = S.n_free, next_slot = S.next_slot;
  for (int j0 = 0; j0 < nd; j0 += kW) {
    const int j = j0 + t;
    const bool b = j < nd && (!have || S.y[j] < 0);
    const int base0 = n_init;
    const int p = compact(b, n_init);
    const int births = n_init - base0;
    if (b) {
      const int r = p - base0;
      int slot;
      if (r < avail) slot = S.free_stack[avail - 1 - r];
      else { slot = next_slot + (r - avail); if (slot >= CAP) { slot = CAP - 1; err = 1; } }
      const int det = S.valid[j];
      S.t_id[slot] = S.next_id + p + 1;
      S.t_conf[slot] = S.dets[static_cast<size_t>(4) * S.ld + det];
      S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
      S.t_det[slot] = det;
      S.t_hits[slot] = 1; S.t_tsu[slot] = 0; S.t_age[slot] = 1;
      S.init_slot[p] = slot; S.init_meas[p] = det;
      if (n_next + p < CAP) next[n_next + p] = slot; else err = 1;
    }
    const int from_free = (births < avail) ? births : avail;
    next_slot += births - from_free;
    avail -= from_free;
  }
  err = __any(err) ? 1 : 0;
  // the free stack now: [0, avail) untouched old entries, [S.n_free, free_top) the slots freed above -> close the gap
  __syncthreads();
  const int freed = free_top - S.n_free;
  if (avail < S.n_free) {
    for (int i0 = 0; i0 < freed; i0 += kW) {  // ascending copy to lower addresses: chunk by chunk, read before write
      const int i = i0 + t;
      const int v = (i < freed) ? S.free_stack[S.n_free + i] : 0;
      __syncthreads();
      if (i < freed) S.free_stack[avail + i] = v;
      __syncthreads();
    }
  }
  const int n_all = n_next + n_init;
  __syncthreads();
  // the new tracks' Kalman records (round 5: here instead of a launch of kf_kernel<XYSR, initiate>; KalmanBoxTracker's constructor as
  // kf_kernels.hip::xysr_init writes it: mean = (z, 0, 0, 0), P = diag(10, 10, 10, 10, 1000, 1000, 1000))
  for (int i = t; i < n_init; i += kW) {
    const int slot = S.init_slot[i], det = S.init_meas[i];
    if (slot < 0 || slot >= CAP) continue;  // (only on a stream that has already raised its error flag)
    const float* zm = init_t[blockIdx.x].meas;
    const int ldm = init_t[blockIdx.x].ldm;
    float flat[56];
#pragma unroll
    for (int q = 0; q < 56; ++q) flat[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) flat[q] = zm[static_cast<size_t>(q) * ldm + det];
#pragma unroll
    for (int q = 0; q < 7; ++q) flat[7 + 8 * q] = (q < 4) ? 10.0f : 10.0f * 100.0f;
    float4* rec = reinterpret_cast<float4*>(S.kmean + static_cast<size_t>(slot) * 56);
#pragma unroll
    for (int q = 0; q < 14; ++q) rec[q] = make_float4(flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]);
  }
  // rows to emit (:218-246)
  int n_out = 0;
  for (int i0 = 0; i0 < n_all && i0 < CAP; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < n_all && i < CAP;
    const int slot = v ? next[i] : 0;
    const bool o = v && S.t_tsu[slot] == 0 && (S.t_hits[slot] >= P.min_hits || S.frame_count <= P.min_hits);
    const int p = compact(o, n_out);
    if (o) S.out_slot[p] = slot;
  }
  if (t == 0) {
    if (n_all > CAP) err = 1;
    S.n_trk = (n_all <= CAP) ? n_all : CAP; S.cur ^= 1;
    S.n_upd = n_upd; S.n_init = n_init; S.n_out = n_out;
    S.next_id += n_init; S.next_slot = next_slot; S.n_free = avail + freed;
    if (err) S.err = 1;
    init_t[blockIdx.x].n = n_init;
    upd_t[blockIdx.x].n = n_upd;
    box_t[blockIdx.x].n = n_out;
  }
}

__global__ void __launch_bounds__(kW) sort_emit(SortStream* streams, int CAP, float* out, int* out_counts, int cap_out, int* max_tracks, int* alive, int* err) {
  SortStream& S = streams[blockIdx.x];
  const int t = threadIdx.x;
  if (S.skip) {
    if (t == 0) { out_counts[blockIdx.x] = 0; alive[blockIdx.x] = S.err ? -S.err : S.n_trk; atomicMax(&max_tracks[blockIdx.x & 63], S.n_trk); if (S.err) atomicMax(err, S.err); }
    return;
  }
  float* rows = out + static_cast<size_t>(blockIdx.x) * cap_out * 8;
  const int n = S.n_out;
  for (int k = t; k < n && k < cap_out; k += kW) {
    const int slot = S.out_slot[k];
    float* r = rows + static_cast<size_t>(k) * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = 0.f;
    {  // the box of the updated state (round 5: here instead of kf_kernel<XYSR, boxes>; kf_kernels.hip::xysr_box, ops.hpp:202-211)
      const float4 m = *reinterpret_cast<const float4*>(S.kmean + static_cast<size_t>(slot) * 56);
      const float w = sqrtf(m.z * m.w);
      const float h = m.z / w;
      r[0] = m.x - w * 0.5f; r[1] = m.y - h * 0.5f; r[2] = m.x + w * 0.5f; r[3] = m.y + h * 0.5f;
    }
    r[4] = static_cast<float>(S.t_id[slot]); r[5] = S.t_conf[slot];
    r[6] = static_cast<float>(S.t_cls[slot]); r[7] = static_cast<float>(S.t_det[slot]);
  }
  if (t == 0) {
    if (n > cap_out) S.err = 2;
    out_counts[blockIdx.x] = (n <= cap_out) ? n : -n;
    atomicMax(&max_tracks[blockIdx.x & 63], S.n_trk);
    alive[blockIdx.x] = S.err ? -S.err : S.n_trk;  // (a stream in error reports -(error code): its caller alone gets the error)
    if (S.err) atomicMax(err, S.err);  // the batch's error word (round 5: gathered here; a kernel of its own before)
  }
}

}  // namespace

struct mot_sort_batch {
  mot_ctx* ctx = nullptr;
  int S = 0, CAP = 0, D = 0;
  SortParams prm{};
  mot::lifecycle::Allocs mem;
  SortStream* d_streams = nullptr;
  std::vector<SortStream> h_streams;
  int *d_counts = nullptr, *d_err = nullptr, *d_maxt = nullptr, *d_alive = nullptr;
  int bound_n = 0;
  float* d_out = nullptr; int* d_out_counts = nullptr; int out_cap = 0;
  mot_det_task* det_t = nullptr;
  mot_kf_task *pred_t = nullptr, *init_t = nullptr, *upd_t = nullptr, *box_t = nullptr;
  mot_lap_task* lap_t = nullptr;
  float* mean = nullptr;  // [S][CAP] records of 7 + 49 floats (mot_kf_task's slab)
  bool profile = false;
  hipEvent_t ev[12] = {};
  double lap_ms = 0.0, frame_ms = 0.0;
  long frames = 0;
  unsigned long long* d_stats = nullptr;  // [64][2]: problems, sum of n + m
  mot::lifecycle::Flights flights;        // mot_sort_enqueue_packed / mot_sort_collect_packed
  const float* d_rows_last = nullptr; const int* d_offsets_last = nullptr; const int* d_counts_last = nullptr;  // mot_sort_device_output
  template <class T>
  T* dalloc(size_t n) { return mem.get<T>(n); }
};


extern "C" {

void mot_sort_destroy(mot_sort_batch* b) {
  if (!b) return;
  b->mem.release();
  b->flights.release();
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  delete b;
}

int mot_sort_reset(mot_sort_batch* b) {  // sort.cpp:97-100: the tracks go, the id counter keeps counting
  MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));
  std::vector<SortStream> cur(b->S);
  MOT_LC_HIP(b, hipMemcpy(cur.data(), b->d_streams, sizeof(SortStream) * b->S, hipMemcpyDeviceToHost));
  std::vector<SortStream> h = b->h_streams;
  for (int s = 0; s < b->S; ++s) h[s].next_id = cur[s].next_id;
  MOT_LC_HIP(b, hipMemcpy(b->d_streams, h.data(), sizeof(SortStream) * b->S, hipMemcpyHostToDevice));
  MOT_LC_HIP(b, hipMemset(b->d_err, 0, sizeof(int)));
  b->bound_n = 0;
  b->flights.drop_all();  // (frames still in flight finished before the copies above: they are dropped with the tracks)
  return MOT_OK;
}

int mot_sort_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, const float* p5, mot_sort_batch** out) {
  if (!ctx || !out || nstreams <= 0 || cap_tracks <= 0 || max_dets <= 0) return MOT_ERR_INVALID;
  auto* b = new mot_sort_batch();
  b->ctx = ctx; b->S = nstreams; b->CAP = cap_tracks; b->D = max_dets;
  b->prm.det_thresh = p5 ? p5[0] : 0.3f;
  b->prm.max_age = p5 ? static_cast<int>(p5[1]) : 1;
  b->prm.min_hits = p5 ? static_cast<int>(p5[3]) : 3;
  b->prm.iou_thr = p5 ? p5[4] : 0.3f;
  const int S = nstreams, CAP = cap_tracks, D = max_dets;
  const size_t ints_per = static_cast<size_t>(CAP) * 14 + static_cast<size_t>(D) * 4;
  int* ip = b->dalloc<int>(ints_per * S);
  float* fp = b->dalloc<float>((static_cast<size_t>(CAP) * 9 + static_cast<size_t>(D) * 8) * S);
  b->mean = b->dalloc<float>(static_cast<size_t>(S) * 56 * CAP);
  b->d_streams = b->dalloc<SortStream>(S);
  b->d_counts = b->dalloc<int>(S);
  b->d_err = b->dalloc<int>(1);
  b->d_maxt = b->dalloc<int>(64);
  b->d_alive = b->dalloc<int>(S);
  b->flights.with_alive = true;
  b->d_stats = b->dalloc<unsigned long long>(128);
  if (b->d_stats) (void)hipMemset(b->d_stats, 0, 128 * sizeof(unsigned long long));
  for (auto& e : b->ev) (void)hipEventCreate(&e);
  b->det_t = b->dalloc<mot_det_task>(S);
  b->pred_t = b->dalloc<mot_kf_task>(S); b->init_t = b->dalloc<mot_kf_task>(S); b->upd_t = b->dalloc<mot_kf_task>(S); b->box_t = b->dalloc<mot_kf_task>(S);
  b->lap_t = b->dalloc<mot_lap_task>(S);
  const size_t wb = (mot::lap_scratch_bytes(CAP, D) + 255) & ~size_t(255);
  char* work = b->dalloc<char>(wb * S);
  int* info = b->dalloc<int>(static_cast<size_t>(4) * S);
  if (!ip || !fp || !b->mean || !b->d_streams || !b->d_counts || !b->d_err || !b->d_maxt || !b->d_alive || !b->det_t || !b->pred_t || !b->init_t ||
      !b->upd_t || !b->box_t || !b->lap_t || !work || !info) {
    mot_sort_destroy(b);
    return MOT_ERR_NOMEM;
  }
  std::vector<SortStream> hs(S);
  std::vector<mot_det_task> det(S);
  std::vector<mot_kf_task> pred(S), init(S), upd(S), box(S);
  std::vector<mot_lap_task> lap(S);
  for (int s = 0; s < S; ++s) {
    SortStream& T = hs[s];
    std::memset(&T, 0, sizeof(T));
    int* i = ip + ints_per * s;
    auto I = [&](int n) { int* r = i; i += n; return r; };
    T.free_stack = I(CAP); T.trk[0] = I(CAP); T.trk[1] = I(CAP);
    T.t_id = I(CAP); T.t_cls = I(CAP); T.t_det = I(CAP); T.t_hits = I(CAP); T.t_tsu = I(CAP); T.t_age = I(CAP);
    T.keep = I(CAP); T.x = I(CAP); T.upd_slot = I(CAP); T.upd_meas = I(CAP); T.out_slot = I(CAP);  // 14 CAP-sized arrays
    T.valid = I(D); T.y = I(D); T.init_slot = I(D); T.init_meas = I(D);
    float* f = fp + (static_cast<size_t>(CAP) * 9 + static_cast<size_t>(D) * 8) * s;
    auto F = [&](int n) { float* r = f; f += n; return r; };
    T.t_conf = F(CAP); T.pbox = F(4 * CAP); T.obox = F(4 * CAP);
    float* d_box = F(4 * D); float* d_meas = F(4 * D);
    float* mean = b->mean + static_cast<size_t>(s) * 56 * CAP;
    T.kmean = mean;
    float* cov = mean + 7;
    std::memset(&det[s], 0, sizeof(mot_det_task));
    det[s].box = d_box; det[s].ldb = D; det[s].meas = d_meas; det[s].ldm = D;
    auto kf = [&](mot_kf_task& k) {
      std::memset(&k, 0, sizeof(k));
      k.mean = mean; k.cov = cov; k.cap = CAP;
      k.q[0] = 0.01f; k.q[1] = 0.01f; k.q[2] = 0.0001f;  // xysr_kf.cpp:52-65
    };
    kf(pred[s]); pred[s].boxes = T.pbox; pred[s].ldb = CAP;  // in place: dst = src (set per frame: the list buffer alternates)
    kf(init[s]); init[s].src = T.init_slot; init[s].dst = T.init_slot; init[s].meas = d_meas; init[s].ldm = D; init[s].midx = T.init_meas;
    kf(upd[s]); upd[s].src = T.upd_slot; upd[s].dst = T.upd_slot; upd[s].meas = d_meas; upd[s].ldm = D; upd[s].midx = T.upd_meas;
    kf(box[s]); box[s].src = T.out_slot; box[s].boxes = T.obox; box[s].ldb = CAP;
    mot_lap_task& L = lap[s];
    std::memset(&L, 0, sizeof(L));
    L.x = T.x; L.y = T.y; L.thresh = 1.0f - b->prm.iou_thr; L.mode = MOT_LAP_PLAIN; L.info = info + static_cast<size_t>(s) * 4;
    L.work = work + static_cast<size_t>(s) * wb;
    L.geom.a = T.pbox; L.geom.lda = CAP; L.geom.aidx = T.keep; L.geom.b = d_box; L.geom.ldb = D; L.geom.bidx = T.valid;
    L.geom.mode = MOT_COST_IOU_DIST;
  }
  b->h_streams = hs;
  hipStream_t st = ctx->stream;
#define SD_UP(dst, vec) MOT_LC_HIP(b, hipMemcpyAsync(dst, vec.data(), sizeof(vec[0]) * vec.size(), hipMemcpyHostToDevice, st))
  SD_UP(b->d_streams, hs); SD_UP(b->det_t, det); SD_UP(b->pred_t, pred); SD_UP(b->init_t, init); SD_UP(b->upd_t, upd); SD_UP(b->box_t, box);
  SD_UP(b->lap_t, lap);
#undef SD_UP
  MOT_LC_HIP(b, hipMemsetAsync(b->d_err, 0, sizeof(int), st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  *out = b;
  return MOT_OK;
}

// queues one frame of every stream (counts: host memory that stays valid until the copy has run — the caller's array for the synchronous
// calls, the flight's page-locked copy for frames in flight); cap_out = staging rows per stream; bound = live tracks any stream may have
static int sort_enqueue_frame(mot_sort_batch* b, const float* d_dets, const int* counts, int cap_out, int bound, hipEvent_t* ev,
                              const mot::lifecycle::FrameDev* fd = nullptr) {
  const bool prof = ev != nullptr;
  hipStream_t st = b->ctx->stream;
  const int S = b->S, CAP = b->CAP, D = b->D;
  if (cap_out > b->out_cap) {
    b->d_out = b->dalloc<float>(static_cast<size_t>(S) * cap_out * 8);
    b->d_out_counts = b->d_out_counts ? b->d_out_counts : b->dalloc<int>(S);
    if (!b->d_out || !b->d_out_counts) return MOT_ERR_NOMEM;
    b->out_cap = cap_out;
  }
  mot::lifecycle::FrameDev FD;
  if (fd) FD = *fd;
  else {
    MOT_LC_HIP(b, hipMemcpyAsync(b->d_counts, counts, sizeof(int) * S, hipMemcpyHostToDevice, st));
    FD.counts = b->d_counts;
  }
  if (!b->flights.maxt_clean) MOT_LC_HIP(b, hipMemsetAsync(b->d_maxt, 0, 64 * sizeof(int), st));  // (else: the last frame's pack_offsets cleared them)
  b->flights.maxt_clean = false;
  int bd = 1;
  for (int s = 0; s < S; ++s) bd = (counts[s] > bd) ? counts[s] : bd;
  if (bd > D) bd = D;
  const int bn = (bound < 1) ? 1 : (bound > CAP ? CAP : bound);  // tracks alive after the previous frame (+ what frames in flight may add)
  int active = 0;  // streams with a frame: a launch with a handful of problems is tuned for latency (mot::launch_lap)
  for (int s = 0; s < S; ++s) active += (counts[s] >= 0) ? 1 : 0;
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[0], st));
  hipLaunchKernelGGL(sort_begin, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, D, FD, d_dets, b->det_t, b->pred_t);
  // (round 5: detection preparation in sort_begin, initiations in sort_apply, output boxes and the error word in sort_emit: four launches fewer)
  MOT_LC_HIP(b, mot::launch_kf_op(1, MOT_KF_XYSR, b->pred_t, S, bn, st));
  hipLaunchKernelGGL(sort_assoc, dim3(S), dim3(kW), 0, st, b->d_streams, CAP, b->lap_t, prof ? b->d_stats : nullptr);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[1], st));
  MOT_LC_HIP(b, mot::launch_lap(b->lap_t, S, bn, bd, true, false, true, st, 0, 0, true, nullptr, nullptr, nullptr, active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[2], st));
  hipLaunchKernelGGL(sort_apply, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, b->init_t, b->upd_t, b->box_t);
  MOT_LC_HIP(b, mot::launch_kf_op(2, MOT_KF_XYSR, b->upd_t, S, bn, st));
  hipLaunchKernelGGL(sort_emit, dim3(S), dim3(kW), 0, st, b->d_streams, CAP, b->d_out, b->d_out_counts, cap_out, b->d_maxt, b->d_alive, b->d_err);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[3], st));
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}
static int sort_account(mot_sort_batch* b, const int* maxt, hipEvent_t* ev) {
  b->bound_n = 0;
  for (int i = 0; i < 64; ++i) b->bound_n = (maxt[i] > b->bound_n) ? maxt[i] : b->bound_n;
  if (ev) {
    float ms = 0.f;
    MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[1], ev[2])); b->lap_ms += ms;
    MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[0], ev[3])); b->frame_ms += ms;
    b->frames += 1;
  }
  return MOT_OK;
}

int mot_sort_step(mot_sort_batch* b, const float* d_dets, const int* h_counts, float* out, int* out_counts, int cap_out) {
  hipStream_t st = b->ctx->stream;
  const int S = b->S;
  if (b->flights.count > 0) { b->ctx->err = "mot_sort_step: frames are in flight (collect them first)"; return MOT_ERR_INVALID; }
  const bool prof = b->profile;
  const int rc = sort_enqueue_frame(b, d_dets, h_counts, cap_out, b->bound_n, prof ? b->ev : nullptr);
  if (rc != MOT_OK) return rc;
  int err = 0, maxt[64];
  MOT_LC_HIP(b, hipMemcpyAsync(out, b->d_out, sizeof(float) * static_cast<size_t>(S) * cap_out * 8, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(out_counts, b->d_out_counts, sizeof(int) * S, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(&err, b->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(maxt, b->d_maxt, sizeof(maxt), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  const int ra = sort_account(b, maxt, prof ? b->ev : nullptr);
  if (ra != MOT_OK) return ra;
  if (err) { b->ctx->err = "mot_sort_step: a stream exceeded cap_tracks / max_dets / cap_out"; return MOT_ERR_CAPACITY; }
  return MOT_OK;
}

// ---- packed output + frames in flight (as mot_bt_*; Sort::update, src/trackers/sort.cpp:102-255, emits at most one row per live track) ----
static int sort_enqueue_flight(mot_sort_batch* b, const float* d_dets, const int* h_counts, int rows_cap, const mot_frame_in* in) {
  if (b->flights.count >= 2) { b->ctx->err = "mot_sort_enqueue: two frames are already in flight (collect one first)"; return MOT_ERR_INVALID; }
  const int slot = b->flights.slot_for_enqueue();
  int* counts_in = nullptr;
  int bd = 0;
  MOT_LC_HIP(b, b->flights.prepare(b->mem, slot, b->S, rows_cap, h_counts, &counts_in, &bd, in != nullptr, b->profile));
  mot::lifecycle::Flight& F = b->flights.fl[slot];
  mot::lifecycle::FrameDev fd;
  if (in) MOT_LC_HIP(b, b->flights.upload_block(b->mem, slot, b->S, in->h_counts, in->h_det_ld, in->h_det_off, nullptr, b->ctx->stream, &fd));
  const int bound = b->bound_n + b->flights.pending_bd();  // a frame still in flight adds at most one track per detection
  const int rc = sort_enqueue_frame(b, d_dets, counts_in, b->CAP, bound, F.prof ? F.ev : nullptr, in ? &fd : nullptr);
  if (rc != MOT_OK) return rc;
  MOT_LC_HIP(b, b->flights.finish(slot, b->ctx->stream, b->d_out, b->CAP, b->d_out_counts, b->S, b->d_err, b->d_maxt, nullptr, rows_cap, bd, nullptr, nullptr, b->d_alive));
  return MOT_OK;
}
static int sort_pop_flight(mot_sort_batch* b, mot::lifecycle::Flight** out, int* total) {
  if (b->flights.count <= 0) { b->ctx->err = "mot_sort_collect: no frame in flight"; return MOT_ERR_INVALID; }
  mot::lifecycle::Flight* F = nullptr;
  MOT_LC_HIP(b, b->flights.pop(&F));
  const int ra = sort_account(b, b->flights.maxt_of(*F), F->prof ? F->ev : nullptr);
  if (ra != MOT_OK) return ra;
  *total = F->h_meta[0];
  *out = F;
  b->d_rows_last = F->view ? F->h_rows : F->d_packed; b->d_offsets_last = F->d_offsets; b->d_counts_last = F->d_counts;
  if (F->h_meta[1]) { b->ctx->err = "mot_sort_collect: a stream exceeded cap_tracks / max_dets"; return MOT_ERR_CAPACITY; }
  if (*total > F->rows_cap) { b->ctx->err = "mot_sort_collect: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  return MOT_OK;
}
int mot_sort_enqueue_packed(mot_sort_batch* b, const float* d_dets, const int* h_counts, int rows_cap) {
  if (!b || !d_dets || !h_counts || rows_cap <= 0) return MOT_ERR_INVALID;
  return sort_enqueue_flight(b, d_dets, h_counts, rows_cap, nullptr);
}
int mot_sort_collect_packed(mot_sort_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b || !out_counts) return MOT_ERR_INVALID;  // rows == NULL: the table stays on the device (mot_sort_device_output), only the counts come back
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = sort_pop_flight(b, &F, &total);
  if (F) std::memcpy(out_counts, b->flights.counts_of(*F), sizeof(int) * b->S);
  if (total_rows) *total_rows = total;
  if (rc != MOT_OK) return rc;
  if (!rows) return MOT_OK;
  if (total > rows_cap) { b->ctx->err = "mot_sort_collect_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  MOT_LC_HIP(b, b->flights.copy_rows(*F, rows, total));
  return MOT_OK;
}
// ---- pooled form (see mot_bt_enqueue_frame) ----
int mot_sort_enqueue_frame(mot_sort_batch* b, const mot_frame_in* in, int rows_cap) {
  if (!b || !in || !in->d_dets || !in->h_counts || !in->h_det_ld || !in->h_det_off || rows_cap <= 0) return MOT_ERR_INVALID;
  return sort_enqueue_flight(b, in->d_dets, in->h_counts, rows_cap, in);
}
int mot_sort_collect_view(mot_sort_batch* b, mot_frame_view* out) {
  if (!b || !out) return MOT_ERR_INVALID;
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = sort_pop_flight(b, &F, &total);
  if (!F) return rc;
  if (!F->view) { b->ctx->err = "mot_sort_collect_view: the frame was queued with mot_sort_enqueue_packed"; return MOT_ERR_INVALID; }
  out->rows = F->h_rows; out->counts = b->flights.counts_of(*F); out->alive = b->flights.alive_of(*F, b->S); out->total = total;
  return rc;
}
int mot_sort_reset_stream(mot_sort_batch* b, int s, int fresh) {
  if (!b || s < 0 || s >= b->S) return MOT_ERR_INVALID;
  hipLaunchKernelGGL(mot::lifecycle::reset_stream_kernel<SortStream>, dim3(1), dim3(64), 0, b->ctx->stream, b->d_streams, s, b->h_streams[s], fresh ? 0 : 1, b->d_err);
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}
namespace {
__global__ void __launch_bounds__(256) sort_move(const SortStream* from, SortStream* to, const float* mean_from, float* mean_to, int cap) {
  using mot::lifecycle::move_array;
  const SortStream& A = *from;
  SortStream& B = *to;
  const size_t n = static_cast<size_t>(cap);
  move_array(B.free_stack, A.free_stack, n); move_array(B.trk[0], A.trk[A.cur], n);
  move_array(B.t_id, A.t_id, n); move_array(B.t_cls, A.t_cls, n); move_array(B.t_det, A.t_det, n); move_array(B.t_hits, A.t_hits, n);
  move_array(B.t_tsu, A.t_tsu, n); move_array(B.t_age, A.t_age, n); move_array(B.t_conf, A.t_conf, n);
  move_array(mean_to, mean_from, n * 56);
  __syncthreads();
  if (threadIdx.x == 0) {
    B.frame_count = A.frame_count; B.next_id = A.next_id; B.next_slot = A.next_slot; B.n_free = A.n_free; B.n_trk = A.n_trk; B.err = A.err;
    B.cur = 0; B.skip = 1;
  }
}
}  // namespace
int mot_sort_move_stream(mot_sort_batch* src, int s, mot_sort_batch* dst, int s2) {
  if (!src || !dst || s < 0 || s >= src->S || s2 < 0 || s2 >= dst->S || dst->CAP < src->CAP || dst->D < src->D) return MOT_ERR_INVALID;
  MOT_LC_HIP(src, hipStreamSynchronize(src->ctx->stream));
  hipStream_t st = dst->ctx->stream;
  hipLaunchKernelGGL(sort_move, dim3(1), dim3(256), 0, st, src->d_streams + s, dst->d_streams + s2, src->mean + static_cast<size_t>(s) * 56 * src->CAP,
                     dst->mean + static_cast<size_t>(s2) * 56 * dst->CAP, src->CAP);
  MOT_LC_HIP(dst, hipGetLastError());
  SortStream h;
  MOT_LC_HIP(dst, hipMemcpyAsync(&h, dst->d_streams + s2, sizeof(SortStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(dst, hipStreamSynchronize(st));
  if (h.n_trk > dst->bound_n) dst->bound_n = h.n_trk;
  return MOT_OK;
}
int mot_sort_step_packed(mot_sort_batch* b, const float* d_dets, const int* h_counts, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b) return MOT_ERR_INVALID;
  if (b->flights.count > 0) { b->ctx->err = "mot_sort_step_packed: frames are in flight (collect them first)"; return MOT_ERR_INVALID; }
  const int rc = mot_sort_enqueue_packed(b, d_dets, h_counts, rows_cap);
  return (rc != MOT_OK) ? rc : mot_sort_collect_packed(b, rows, rows_cap, out_counts, total_rows);
}
int mot_sort_device_output(mot_sort_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts) {
  if (!b || !b->d_rows_last || !b->d_offsets_last) return MOT_ERR_INVALID;
  if (d_rows) *d_rows = b->d_rows_last;
  if (d_offsets) *d_offsets = b->d_offsets_last;
  if (d_counts) *d_counts = b->d_counts_last;
  return MOT_OK;
}

int mot_sort_profile(mot_sort_batch* b, int enable) {
  b->profile = enable != 0;
  if (enable) {
    b->lap_ms = b->frame_ms = 0.0;
    b->frames = 0;
    MOT_LC_HIP(b, hipMemset(b->d_stats, 0, 128 * sizeof(unsigned long long)));
  }
  return MOT_OK;
}
int mot_sort_profile_stats(mot_sort_batch* b, double* out8) {
  unsigned long long raw[128];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long pr = 0, nm = 0;
  for (int i = 0; i < 64; ++i) { pr += raw[2 * i]; nm += raw[2 * i + 1]; }
  out8[0] = b->lap_ms; out8[1] = 0.0; out8[2] = b->frame_ms; out8[3] = static_cast<double>(b->frames);
  out8[4] = static_cast<double>(pr); out8[5] = static_cast<double>(nm); out8[6] = 0.0; out8[7] = 0.0;
  return MOT_OK;
}

int mot_sort_dump(mot_sort_batch* b, int s, int* ids, float* mean, float* cov, int cap) {
  hipStream_t st = b->ctx->stream;
  SortStream h;
  MOT_LC_HIP(b, hipMemcpyAsync(&h, b->d_streams + s, sizeof(SortStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  const int n = h.n_trk;
  if (n > cap) return -n;
  std::vector<int> slots(n), tid(b->CAP);
  if (n) MOT_LC_HIP(b, hipMemcpyAsync(slots.data(), h.trk[h.cur], sizeof(int) * n, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(tid.data(), h.t_id, sizeof(int) * b->CAP, hipMemcpyDeviceToHost, st));
  const int C = b->CAP;
  std::vector<float> m(static_cast<size_t>(56) * C);
  MOT_LC_HIP(b, hipMemcpyAsync(m.data(), b->mean + static_cast<size_t>(s) * 56 * C, sizeof(float) * m.size(), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {
    const int sl = slots[i];
    ids[i] = tid[sl];
    for (int k = 0; k < 7; ++k) mean[static_cast<size_t>(i) * 7 + k] = m[static_cast<size_t>(sl) * 56 + k];
    for (int k = 0; k < 49; ++k) cov[static_cast<size_t>(i) * 49 + k] = m[static_cast<size_t>(sl) * 56 + 7 + k];
  }
  return n;
}

}  // extern "C"
