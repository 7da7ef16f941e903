// OC-SORT with the per-stream lifecycle ON THE DEVICE (reference: src/trackers/ocsort.cpp:285-738; the host-side stage machine
// with the same semantics is host/ocsort.cpp).
//
// Same construction as bt_device.hip / bot_device.hip: the tracker list is an array of Kalman slots in the reference's order,
// per-track records (age, hits, hit streak, last observation, the observation history that k_previous_obs looks into, the
// velocity direction) are indexed by slot, and the reference's bookkeeping runs in small kernels (one wavefront per stream)
// between the numeric ones. A frame of S streams is a FIXED launch sequence:
//   oc_begin -> det_prepare, kf_predict (x6 clamp, in place) -> oc_nan (NaN-row rule, velocity / k-previous-observation planes)
//   -> ocsort cost matrix -> assignment 1 (trivial-case shortcut or lapjv) -> oc_after_first (IoU filter, quirk Q4)
//   [-> assignment BYTE -> oc_after_byte] -> assignment OCR (last observations, -IoU, min gate) -> oc_finish (spawns, update
//   rounds, rows) -> kf_initiate, kf_update x kRounds, kf_boxes -> oc_emit (output rows newest first, age-out).
// Quirk Q4 (ocsort.cpp:699-714 + :716-727): a pair the assignment made but the IoU filter rejects puts its detection and its
// track on the unmatched lists TWICE; the lists therefore carry repeats, a track can be updated more than once in a frame (in
// list order: the Kalman updates run in kRounds sequential launches, a slot's r-th update of the frame in launch r) and a
// detection left over twice spawns two tracks.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "lifecycle_common.hpp"

namespace mot {
hipError_t launch_ocsort(const mot_ocsort_task*, int, int, int, bool, hipStream_t);
}

namespace {
using mot::lifecycle::compact;
using mot::lifecycle::kW;
using mot::lifecycle::FrameDev;

constexpr int kRounds = 4;  // Kalman updates one slot can receive in a frame before the stream reports an error

struct OcParams {
  float det_thresh, thr, min_conf, inertia, frame_diag;
  int max_age, min_hits, delta_t, use_byte, asso, K;
};

struct OcStream {
  // ---- persistent ----
  int frame_count, next_id, next_slot, n_free, n_trk, cur, err;
  int* free_stack;
  int* trk[2];
  int *t_id, *t_age, *t_hits, *t_streak, *t_tsu, *t_cls, *t_det, *t_nobs, *t_need;
  float *t_conf, *t_last, *t_vel;  // [CAP], [CAP][5], [CAP][2]
  int* t_oage;                      // [CAP][K]
  float* t_obs;                     // [CAP][K][5]
  // ---- frame ----
  const float* dets; int ld, n;
  int skip;  // the stream sits this frame out (pooled form: counts[s] < 0)
  int *high, *second; int n_high, n_second;
  int nt0, silent, lap1_q, byte_q, rem_q;
  float *pbox, *vel, *prev, *lbox, *sbox;  // [4][CAP], [2][CAP], [5][CAP], [4][CAP], [4][CAP] (sbox: unused since round 5)
  const float* kmean;  // this stream's Kalman records (56 floats per slot)
  int *x1, *y1, *xb, *yb, *xr, *yr;
  float *xval1, *xvalb, *xvalr;
  int *info1, *infob, *infor;
  unsigned char *md, *mt, *rm_d, *rm_t;
  int *um_dets, *um_trks; int n_umd, n_umt;
  int* left;
  int *upd_slot, *upd_meas, *upd_round; int n_upd;
  int* slot_cnt;
  int *init_dst, *init_meas; int n_init;
  int* need_slot; int n_need;
  int* r_slot[kRounds]; int* r_meas[kRounds];
};

struct OcTasks {
  mot_det_task* det;
  mot_kf_task *pred, *init, *upd /*[kRounds][S]*/, *sbox;
  mot_ocsort_task* cost;
  mot_lap_task *lap1, *lapb, *lapr;
};

// k_previous_obs, ocsort.cpp:24-51: the observation made at age - k, else age - k + 1, ..., else the newest one
__device__ __forceinline__ void k_previous(const OcStream& S, int K, int slot, int k, float out[5]) {
  const int n = S.t_nobs[slot];
  if (n == 0) { for (int c = 0; c < 5; ++c) out[c] = -1.0f; return; }
  const int age = S.t_age[slot];
  const int* oa = S.t_oage + static_cast<size_t>(slot) * K;
  const float* ob = S.t_obs + static_cast<size_t>(slot) * K * 5;
  for (int i = 0; i < k; ++i) {
    const int key = age - (k - i);
    for (int q = 0; q < n; ++q)
      if (oa[q] == key) { for (int c = 0; c < 5; ++c) out[c] = ob[q * 5 + c]; return; }
  }
  for (int c = 0; c < 5; ++c) out[c] = ob[(n - 1) * 5 + c];
}
__device__ __forceinline__ void speed_direction(const float* b1, const float* b2, float out[2]) {  // :160-172
  const float cx1 = (b1[0] + b1[2]) / 2.0f, cy1 = (b1[1] + b1[3]) / 2.0f;
  const float cx2 = (b2[0] + b2[2]) / 2.0f, cy2 = (b2[1] + b2[3]) / 2.0f;
  const float dy = cy2 - cy1, dx = cx2 - cx1;
  const float norm = sqrtf(dy * dy + dx * dx) + 1e-6f;
  out[0] = dy / norm; out[1] = dx / norm;
}
// KalmanBoxTracker::update :89-130 without the Kalman part (queued by the caller)
__device__ __forceinline__ void apply(OcStream& S, const OcParams& P, int slot, int det) {
  S.t_det[slot] = det;
  const float conf = S.dets[static_cast<size_t>(4) * S.ld + det];
  S.t_conf[slot] = conf;
  S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
  float b[4];
  for (int k = 0; k < 4; ++k) b[k] = S.dets[static_cast<size_t>(k) * S.ld + det];
  float* last = S.t_last + static_cast<size_t>(slot) * 5;
  const float ls = last[0] + last[1] + last[2] + last[3];
  if (ls >= 0) {
    float pb[5], v[2];
    k_previous(S, P.K, slot, P.delta_t, pb);
    if (pb[0] + pb[1] + pb[2] + pb[3] >= 0) speed_direction(pb, b, v);
    else speed_direction(last, b, v);
    S.t_vel[slot * 2] = v[0]; S.t_vel[slot * 2 + 1] = v[1];
  }
  for (int k = 0; k < 4; ++k) last[k] = b[k];
  last[4] = conf;
  int n = S.t_nobs[slot];
  int* oa = S.t_oage + static_cast<size_t>(slot) * P.K;
  float* ob = S.t_obs + static_cast<size_t>(slot) * P.K * 5;
  const int age = S.t_age[slot];
  if (n > 0 && oa[n - 1] == age) {
    for (int c = 0; c < 5; ++c) ob[(n - 1) * 5 + c] = last[c];
  } else {
    if (n == P.K) {  // keep the newest delta_t + 2 entries (older ones are never looked up)
      for (int q = 1; q < n; ++q) { oa[q - 1] = oa[q]; for (int c = 0; c < 5; ++c) ob[(q - 1) * 5 + c] = ob[q * 5 + c]; }
      n -= 1;
    }
    oa[n] = age;
    for (int c = 0; c < 5; ++c) ob[n * 5 + c] = last[c];
    S.t_nobs[slot] = n + 1;
  }
  S.t_tsu[slot] = 0; S.t_hits[slot] += 1; S.t_streak[slot] += 1;
}

// One chunk of (slot, det) updates in list order: appends them to the frame's update list with the round each belongs to
// and applies them — lanes that hit the same track go one after the other, in lane order.
__device__ __forceinline__ void apply_chunk(OcStream& S, const OcParams& P, bool valid, int slot, int det, int& n_upd, int* lds_slot) {
  const int lane = static_cast<int>(threadIdx.x);
  lds_slot[lane] = valid ? slot : -1;
  __syncthreads();
  int rank = 0;
  if (valid)
    for (int j = 0; j < lane; ++j) rank += (lds_slot[j] == slot) ? 1 : 0;
  const int base = valid ? S.slot_cnt[slot] : 0;
  __syncthreads();
  const int p = compact(valid, n_upd);
  if (valid) {
    S.upd_slot[p] = slot; S.upd_meas[p] = det; S.upd_round[p] = base + rank;
    atomicAdd(&S.slot_cnt[slot], 1);
  }
  int maxr = rank;
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(maxr, o, 64); maxr = (v > maxr) ? v : maxr; }
  for (int r = 0; r <= maxr; ++r) {
    if (valid && rank == r) apply(S, P, slot, det);
    __syncthreads();
  }
}

// removes every entry e of list[0..n) with flag[e] != 0, order preserved, in place; returns the new length
__device__ __forceinline__ int filter_list(int* list, int n, const unsigned char* flag) {
  const int t = static_cast<int>(threadIdx.x);
  int m = 0;
  for (int k0 = 0; k0 < n; k0 += kW) {
    const int k = k0 + t;
    const int e = (k < n) ? list[k] : 0;
    const bool keep = k < n && flag[e] == 0;
    __syncthreads();
    const int p = compact(keep, m);  // p <= k
    if (keep) list[p] = e;
    __syncthreads();
  }
  return m;
}

// ---- K0: detection split, KalmanBoxTracker::predict's counters (:132-148), the predict task ----
__global__ void __launch_bounds__(kW) oc_begin(OcStream* streams, OcParams P, int CAP, int D, int S_total, FrameDev FD, const float* dets_base, OcTasks K) {
  const int s = blockIdx.x;
  OcStream& S = streams[s];
  const int t = static_cast<int>(threadIdx.x);
  int n = FD.counts[s];
  if (n < 0) {  // not this stream's frame: its state stays as it is, every task it owns is empty
    if (t == 0) {
      S.skip = 1;
      K.det[s].n = 0; K.pred[s].n = 0; K.init[s].n = 0; K.sbox[s].n = 0;
      for (int r = 0; r < kRounds; ++r) K.upd[static_cast<size_t>(r) * S_total + s].n = 0;
      K.cost[s].nd = 0; K.cost[s].nt = 0;
      K.lap1[s].n = 0; K.lap1[s].m = 0;
      K.lapb[s].n = 0; K.lapb[s].m = 0; K.lapb[s].geom.n = 0; K.lapb[s].geom.m = 0;
      K.lapr[s].n = 0; K.lapr[s].m = 0; K.lapr[s].geom.n = 0; K.lapr[s].geom.m = 0;
    }
    return;
  }
  int ldd = D;
  const float* dets = mot::lifecycle::frame_dets(FD, dets_base, s, D, ldd);
  const bool over = n > D;
  if (over) n = 0;
  const float* conf = dets + static_cast<size_t>(4) * ldd;
  float* dbox = K.det[s].box;
  float* dmeas = K.det[s].meas;
  const int dldb = K.det[s].ldb, dldm = K.det[s].ldm;
  int nh = 0, ns = 0;
  for (int i0 = 0; i0 < n; i0 += kW) {
    const int i = i0 + t;
    const float c = (i < n) ? conf[i] : 0.f;
    const bool lo = i < n && c > P.min_conf && c < P.det_thresh;
    const bool hi = i < n && c > P.det_thresh;
    const int pl = compact(lo, ns);
    if (lo) S.second[pl] = i;
    const int ph = compact(hi, nh);
    if (hi) S.high[ph] = i;
    if (i < n) {  // the detection's box and measurement (round 5: here instead of det_kernel<MOT_DET_XYSR>, ops.hpp:188-197: the same operations)
      const float x1 = dets[i], y1 = dets[static_cast<size_t>(ldd) + i], x2 = dets[static_cast<size_t>(2) * ldd + i], y2 = dets[static_cast<size_t>(3) * ldd + i];
      const float w = x2 - x1, h = y2 - y1;
      const float zz[4] = {x1 + w * 0.5f, y1 + h * 0.5f, w * h, (h > 1e-6f) ? (w / h) : 0.0f};
      const float bb[4] = {x1, y1, x2, y2};
#pragma unroll
      for (int q = 0; q < 4; ++q) { dbox[static_cast<size_t>(q) * dldb + i] = bb[q]; dmeas[static_cast<size_t>(q) * dldm + i] = zz[q]; }
    }
  }
  const int* trk = S.trk[S.cur];
  for (int i = t; i < S.n_trk; i += kW) {
    const int slot = trk[i];
    S.t_age[slot] += 1;
    if (S.t_tsu[slot] > 0) S.t_streak[slot] = 0;
    S.t_tsu[slot] += 1;
    S.slot_cnt[slot] = 0;
  }
  if (t == 0) {
    S.frame_count += 1;
    S.dets = dets; S.ld = ldd; S.n = n; S.skip = 0;
    if (over) S.err = 1;
    S.n_high = nh; S.n_second = ns; S.nt0 = S.n_trk;
    S.n_upd = 0; S.n_umd = 0; S.n_umt = 0; S.n_init = 0; S.n_need = 0; S.silent = 0; S.lap1_q = 0; S.byte_q = 0; S.rem_q = 0;
    K.det[s].dets = dets; K.det[s].ld = ldd; K.det[s].n = n;
    K.pred[s].n = S.n_trk; K.pred[s].src = trk;
  }
}

// ---- K1: NaN-row rule (:353-364), the cost kernel's per-track planes, the first association's tasks ----
__global__ void __launch_bounds__(kW) oc_nan(OcStream* streams, OcParams P, int CAP, OcTasks K, unsigned long long* stats) {
  const int s = blockIdx.x;
  OcStream& S = streams[s];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip) return;
  const int nt = S.nt0;
  const int* trk = S.trk[S.cur];
  int* kept = S.trk[S.cur ^ 1];
  int nk = 0, free_top = S.n_free;
  for (int i0 = 0; i0 < nt; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < nt;
    bool bad = false;
    if (v)
      for (int c = 0; c < 4; ++c) { const float x = S.pbox[static_cast<size_t>(c) * CAP + i]; bad = bad || (x != x); }
    const bool ok = v && !bad;
    const int slot = v ? trk[i] : 0;
    const int p = compact(ok, nk);
    if (ok) kept[p] = slot;
    const bool dead = v && bad;
    const int pf = compact(dead, free_top);
    if (dead) S.free_stack[pf] = slot;
  }
  __syncthreads();
  // rows of `trks` are the FIRST nk predicted boxes (:363-364), velocities / k-previous observations those of the survivors
  for (int i = t; i < nk; i += kW) {
    const int slot = kept[i];
    S.vel[i] = S.t_vel[slot * 2]; S.vel[static_cast<size_t>(CAP) + i] = S.t_vel[slot * 2 + 1];
    float kp[5];
    k_previous(S, P.K, slot, P.delta_t, kp);
    for (int c = 0; c < 5; ++c) S.prev[static_cast<size_t>(c) * CAP + i] = kp[c];
  }
  if (t == 0) {
    S.n_trk = nk; S.cur ^= 1; S.n_free = free_top;
    S.silent = (nk == 0) ? 1 : 0;  // :366-383
    const bool q = nk > 0 && S.n_high > 0;
    S.lap1_q = q;
    mot_ocsort_task& C = K.cost[s];
    C.nd = q ? S.n_high : 0; C.nt = q ? nk : 0; C.dconf = S.dets + static_cast<size_t>(4) * S.ld;
    mot_lap_task& L = K.lap1[s];
    L.n = q ? S.n_high : 0; L.m = q ? nk : 0;
    K.lapb[s].n = 0; K.lapb[s].m = 0; K.lapb[s].geom.n = 0; K.lapb[s].geom.m = 0;
    K.lapr[s].n = 0; K.lapr[s].m = 0; K.lapr[s].geom.n = 0; K.lapr[s].geom.m = 0;
    if (stats && q) { unsigned long long* st = stats + (s & 63) * 8; atomicAdd(&st[0], 1ull); atomicAdd(&st[1], static_cast<unsigned long long>(S.n_high) * nk); }
  }
}

// OCR rematch (:475-540): leftover detections against the LAST OBSERVATIONS of the leftover tracks
__device__ __forceinline__ void queue_rematch(OcStream& S, int ldl, mot_lap_task& L) {
  const int t = static_cast<int>(threadIdx.x);
  const bool q = S.n_umd > 0 && S.n_umt > 0;
  if (q) {
    const int* trk = S.trk[S.cur];
    for (int k = t; k < S.n_umt; k += kW) {
      const float* last = S.t_last + static_cast<size_t>(trk[S.um_trks[k]]) * 5;
      for (int c = 0; c < 4; ++c) S.lbox[static_cast<size_t>(c) * ldl + k] = last[c];
    }
    for (int k = t; k < S.n_umd; k += kW) S.left[k] = S.high[S.um_dets[k]];
  }
  if (t == 0) {
    S.rem_q = q;
    L.n = q ? S.n_umd : 0; L.m = q ? S.n_umt : 0; L.geom.n = L.n; L.geom.m = L.m;
  }
}

// ---- K2: ocsort_assoc::associate's filter (:699-727), Q4, the BYTE / OCR tasks ----
__global__ void __launch_bounds__(kW) oc_after_first(OcStream* streams, OcParams P, int CAP, int D, OcTasks K) {
  __shared__ int lds_slot[kW];
  const int s = blockIdx.x;
  OcStream& S = streams[s];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip) return;
  const int nd = S.n_high, nt = S.n_trk;
  int n_umd = 0, n_umt = 0, n_upd = 0;
  if (S.silent) {  // no tracker left: every detection is unmatched, nothing is emitted or aged out this frame
    for (int i = t; i < nd; i += kW) S.um_dets[i] = i;
    if (t == 0) { S.n_umd = nd; S.n_umt = 0; S.n_upd = 0; S.byte_q = 0; S.rem_q = 0; }
    return;
  }
  for (int i = t; i < nd; i += kW) S.md[i] = 0;
  for (int j = t; j < nt; j += kW) S.mt[j] = 0;
  __syncthreads();
  const int* trk = S.trk[S.cur];
  if (S.lap1_q) {
    const int path = S.info1[0];
    for (int i0 = 0; i0 < nd; i0 += kW) {
      const int i = i0 + t;
      const int j = (i < nd) ? S.x1[i] : -1;
      const bool has = i < nd && j >= 0;
      const bool acc = has && (path == 1 || S.xval1[i] >= P.thr);
      const bool q4 = has && !acc;
      if (acc) { S.md[i] = 1; S.mt[j] = 1; }
      const int pq = compact(q4, n_umd);
      if (q4) { S.um_dets[pq] = i; S.um_trks[pq] = j; }
      apply_chunk(S, P, acc, acc ? trk[j] : 0, acc ? S.high[i] : 0, n_upd, lds_slot);
    }
    n_umt = n_umd;
  }
  __syncthreads();
  for (int i0 = 0; i0 < nd; i0 += kW) {
    const int i = i0 + t;
    const bool u = i < nd && S.md[i] == 0;
    const int p = compact(u, n_umd);
    if (u) S.um_dets[p] = i;
  }
  for (int j0 = 0; j0 < nt; j0 += kW) {
    const int j = j0 + t;
    const bool u = j < nt && S.mt[j] == 0;
    const int p = compact(u, n_umt);
    if (u) S.um_trks[p] = j;
  }
  __syncthreads();
  if (t == 0) { S.n_umd = n_umd; S.n_umt = n_umt; S.n_upd = n_upd; }
  __syncthreads();
  if (P.use_byte) {  // :430-472
    if (t == 0) {
      const bool q = S.n_second > 0 && n_umt > 0;
      S.byte_q = q;
      mot_lap_task& L = K.lapb[s];
      L.n = q ? S.n_second : 0; L.m = q ? n_umt : 0; L.geom.n = L.n; L.geom.m = L.m;
    }
  } else {
    queue_rematch(S, CAP + D, K.lapr[s]);
  }
}

// ---- K3 (use_byte): BYTE association applied, leftover tracks filtered, OCR task ----
__global__ void __launch_bounds__(kW) oc_after_byte(OcStream* streams, OcParams P, int CAP, int D, OcTasks K) {
  __shared__ int lds_slot[kW];
  const int s = blockIdx.x;
  OcStream& S = streams[s];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip || S.silent) return;
  if (S.byte_q && S.infob[0] != 2) {
    const int* trk = S.trk[S.cur];
    int n_upd = S.n_upd;
    for (int j = t; j < S.n_trk; j += kW) S.rm_t[j] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < S.n_second; i0 += kW) {
      const int i = i0 + t;
      const int j = (i < S.n_second) ? S.xb[i] : -1;
      const bool ok = i < S.n_second && j >= 0 && !(-S.xvalb[i] < P.thr);
      const int ti = ok ? S.um_trks[j] : 0;
      if (ok) S.rm_t[ti] = 1;
      apply_chunk(S, P, ok, ok ? trk[ti] : 0, ok ? S.second[i] : 0, n_upd, lds_slot);
    }
    __syncthreads();
    const int m = filter_list(S.um_trks, S.n_umt, S.rm_t);
    if (t == 0) { S.n_umt = m; S.n_upd = n_upd; }
    __syncthreads();
  }
  queue_rematch(S, CAP + D, K.lapr[s]);
}

// ---- K4: OCR applied, "None" updates, spawns, Kalman update rounds, rows that need the filter state (:541-587) ----
__global__ void __launch_bounds__(kW) oc_finish(OcStream* streams, OcParams P, int CAP, int D, int S_total, OcTasks K) {
  __shared__ int lds_slot[kW];
  const int s = blockIdx.x;
  OcStream& S = streams[s];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip) return;
  int* trk = S.trk[S.cur];
  int n_upd = S.n_upd;
  if (!S.silent && S.rem_q && S.infor[0] != 2) {
    for (int i = t; i < S.n_high; i += kW) S.rm_d[i] = 0;
    for (int j = t; j < S.n_trk; j += kW) S.rm_t[j] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < S.n_umd; i0 += kW) {
      const int i = i0 + t;
      const int j = (i < S.n_umd) ? S.xr[i] : -1;
      const bool ok = i < S.n_umd && j >= 0 && !(-S.xvalr[i] < P.thr);
      const int di = ok ? S.um_dets[i] : 0, ti = ok ? S.um_trks[j] : 0;
      if (ok) { S.rm_d[di] = 1; S.rm_t[ti] = 1; }
      apply_chunk(S, P, ok, ok ? trk[ti] : 0, ok ? S.high[di] : 0, n_upd, lds_slot);
    }
    __syncthreads();
    const int md = filter_list(S.um_dets, S.n_umd, S.rm_d);
    const int mt = filter_list(S.um_trks, S.n_umt, S.rm_t);
    if (t == 0) { S.n_umd = md; S.n_umt = mt; }
    __syncthreads();
  }
  const int n_umd = S.n_umd, n_umt = S.n_umt;
  for (int k = t; k < n_umt; k += kW) S.t_det[trk[S.um_trks[k]]] = 0;  // update(None): det_ind = 0 (:543-545)
  // spawns (:548-556); a detection listed twice spawns twice (Q4)
  int err = 0;
  const int n_trk = S.n_trk;
  {
    const int free_top = S.n_free, next_slot = S.next_slot;
    for (int k = t; k < n_umd; k += kW) {
      int slot;
      if (k < free_top) slot = S.free_stack[free_top - 1 - k];
      else { slot = next_slot + (k - free_top); if (slot >= CAP) { slot = CAP - 1; err = 1; } }
      const int det = S.high[S.um_dets[k]];
      S.t_id[slot] = S.next_id + k + 1;
      S.t_conf[slot] = S.dets[static_cast<size_t>(4) * S.ld + det];
      S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
      S.t_det[slot] = det;
      S.t_age[slot] = 0; S.t_hits[slot] = 0; S.t_streak[slot] = 0; S.t_tsu[slot] = 0; S.t_nobs[slot] = 0;
      for (int c = 0; c < 5; ++c) S.t_last[static_cast<size_t>(slot) * 5 + c] = -1.0f;
      S.t_vel[slot * 2] = 0.f; S.t_vel[slot * 2 + 1] = 0.f;
      S.init_dst[k] = slot; S.init_meas[k] = det;
      if (n_trk + k < CAP) trk[n_trk + k] = slot; else err = 1;
    }
  }
  err = __any(err) ? 1 : 0;
  __syncthreads();
  const int n_all = (n_trk + n_umd <= CAP) ? n_trk + n_umd : CAP;
  // Kalman update rounds: a slot's r-th update of this frame runs in launch r
  int n_round[kRounds];
  for (int r = 0; r < kRounds; ++r) {
    int m = 0;
    for (int k0 = 0; k0 < n_upd; k0 += kW) {
      const int k = k0 + t;
      const bool v = k < n_upd && S.upd_round[k] == r;
      const int p = compact(v, m);
      if (v) { S.r_slot[r][p] = S.upd_slot[k]; S.r_meas[r][p] = S.upd_meas[k]; }
    }
    n_round[r] = m;
  }
  for (int k = t; k < n_upd; k += kW) if (S.upd_round[k] >= kRounds) err = 1;
  err = __any(err) ? 1 : 0;
  // rows to emit newest first (:562-587); the ones without an observation yet take their box from the filter state
  int n_need = 0;
  if (!S.silent) {
    for (int q0 = 0; q0 < n_all; q0 += kW) {
      const int q = q0 + t;
      const int i = n_all - 1 - q;
      const bool v = q < n_all;
      const int slot = v ? trk[i] : 0;
      const bool e = v && S.t_tsu[slot] < 1 && (S.t_streak[slot] >= P.min_hits || S.frame_count <= P.min_hits);
      const float* last = S.t_last + static_cast<size_t>(slot) * 5;
      const bool need = e && (last[0] + last[1] + last[2] + last[3] < 0);
      const int p = compact(need, n_need);
      if (need) { S.need_slot[p] = slot; S.t_need[slot] = p; }
    }
  }
  if (t == 0) {
    const int from_free = (n_umd < S.n_free) ? n_umd : S.n_free;
    S.next_slot += n_umd - from_free; S.n_free -= from_free;
    S.next_id += n_umd;
    S.n_trk = n_all; S.n_upd = n_upd; S.n_init = n_umd; S.n_need = n_need;
    if (err) S.err = 1;
    K.init[s].n = n_umd;
    for (int r = 0; r < kRounds; ++r) K.upd[static_cast<size_t>(r) * S_total + s].n = n_round[r];
    K.sbox[s].n = n_need;
  }
}

// ---- K5: the output table (newest tracker first) and the age-out (:562-606) ----
__global__ void __launch_bounds__(kW) oc_emit(OcStream* streams, OcParams P, int CAP, float* out, int* out_counts, int cap_out, int* max_tracks, int* alive, int* err) {
  OcStream& S = streams[blockIdx.x];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip || S.silent) {
    if (t == 0) { out_counts[blockIdx.x] = 0; alive[blockIdx.x] = S.err ? -S.err : S.n_trk; atomicMax(&max_tracks[blockIdx.x & 63], S.n_trk); if (S.err) atomicMax(err, S.err); }
    return;
  }
  const int* trk = S.trk[S.cur];
  int* next = S.trk[S.cur ^ 1];
  float* rows = out + static_cast<size_t>(blockIdx.x) * cap_out * 8;
  const int n_all = S.n_trk;
  int n_rows = 0;
  for (int q0 = 0; q0 < n_all; q0 += kW) {
    const int q = q0 + t;
    const int i = n_all - 1 - q;
    const bool v = q < n_all;
    const int slot = v ? trk[i] : 0;
    const bool e = v && S.t_tsu[slot] < 1 && (S.t_streak[slot] >= P.min_hits || S.frame_count <= P.min_hits);
    const int p = compact(e, n_rows);
    if (e && p < cap_out) {
      const float* last = S.t_last + static_cast<size_t>(slot) * 5;
      float b[4] = {last[0], last[1], last[2], last[3]};
      if (b[0] + b[1] + b[2] + b[3] < 0) {
        // the box of the track's state (round 5: here instead of kf_kernel<XYSR, boxes> over the tracks that need it; kf_kernels.hip::xysr_box)
        const float4 m = *reinterpret_cast<const float4*>(S.kmean + static_cast<size_t>(slot) * 56);
        const float w = sqrtf(m.z * m.w);
        const float h = m.z / w;
        b[0] = m.x - w * 0.5f; b[1] = m.y - h * 0.5f; b[2] = m.x + w * 0.5f; b[3] = m.y + h * 0.5f;
      }
      float* r = rows + static_cast<size_t>(p) * 8;
      r[0] = b[0]; r[1] = b[1]; r[2] = b[2]; r[3] = b[3];
      r[4] = static_cast<float>(S.t_id[slot] + 1); r[5] = S.t_conf[slot];
      r[6] = static_cast<float>(S.t_cls[slot]); r[7] = static_cast<float>(S.t_det[slot]);
    }
  }
  int n_keep = 0, free_top = S.n_free;
  for (int i0 = 0; i0 < n_all; i0 += kW) {  // trackers not updated for more than max_age frames go (:598-604)
    const int i = i0 + t;
    const bool v = i < n_all;
    const int slot = v ? trk[i] : 0;
    const bool k = v && !(S.t_tsu[slot] > P.max_age);
    const int p = compact(k, n_keep);
    if (k) next[p] = slot;
    const bool dead = v && !k;
    const int pf = compact(dead, free_top);
    if (dead) S.free_stack[pf] = slot;
  }
  if (t == 0) {
    S.n_trk = n_keep; S.cur ^= 1; S.n_free = free_top;
    if (n_rows > cap_out) S.err = 2;
    out_counts[blockIdx.x] = (n_rows <= cap_out) ? n_rows : -n_rows;
    atomicMax(&max_tracks[blockIdx.x & 63], n_keep);
    alive[blockIdx.x] = S.err ? -S.err : n_keep;  // (a stream in error reports -(error code): its caller alone gets the error, lifecycle_common.hpp)
    if (S.err) atomicMax(err, S.err);  // the batch's error word (round 5: gathered here; a kernel of its own before)
  }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------
struct mot_oc_batch {
  mot_ctx* ctx = nullptr;
  int S = 0, CAP = 0, D = 0;
  OcParams prm{};
  mot::lifecycle::Allocs mem;
  OcStream* d_streams = nullptr;
  std::vector<OcStream> h_streams;
  OcTasks tasks{};
  int* d_counts = nullptr;
  int* d_err = nullptr;
  int* d_maxt = nullptr;
  int* d_alive = nullptr;
  int bound_n = 0;
  // first association: when (nearly) every problem of a frame was declined by the certified sparse solver (duplicated tracks, quirk Q4:
  // the optimum is not unique), the next frames go to the exact kernel directly; the sparse solver is tried again every 8th frame
  bool skip_fast[3] = {false, false, false};  // per association (first, byte, re-association): the sparse solver declined nearly every problem
  int lap1_age = 0;
  float* d_out = nullptr; int* d_out_counts = nullptr;
  const float* d_packed = nullptr; const int* d_offsets = nullptr; const int* d_counts_last = nullptr;  // mot_oc_device_output: the frame collected last
  mot::lifecycle::Flights flights;  // mot_oc_enqueue_packed / mot_oc_collect_packed (mot_oc_step_packed = one after the other)
  float* mean = nullptr;  // [S][CAP] records of 7 + 49 floats
  bool profile = false;
  unsigned long long* d_stats = nullptr;
  hipEvent_t ev[12] = {};
  double lap_ms = 0.0, cost_ms = 0.0, frame_ms = 0.0;
  long frames = 0;
  template <class T>
  T* dalloc(size_t n) { return mem.get<T>(n); }
};

extern "C" {

void mot_oc_destroy(mot_oc_batch* b) {
  if (!b) return;
  b->mem.release();
  b->flights.release();
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  delete b;
}

int mot_oc_reset(mot_oc_batch* b) {  // OCSort::reset: the tracker list goes, the id counter keeps counting (ocsort.hpp:37-39: clear_count() is empty)
  MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));  // (frames still in flight have finished: they are dropped with the tracks)
  std::vector<OcStream> cur(b->S);
  MOT_LC_HIP(b, hipMemcpy(cur.data(), b->d_streams, sizeof(OcStream) * b->S, hipMemcpyDeviceToHost));
  std::vector<OcStream> h = b->h_streams;
  for (int s = 0; s < b->S; ++s) h[s].next_id = cur[s].next_id;
  MOT_LC_HIP(b, hipMemcpy(b->d_streams, h.data(), sizeof(OcStream) * b->S, hipMemcpyHostToDevice));
  MOT_LC_HIP(b, hipMemset(b->d_err, 0, sizeof(int)));
  b->bound_n = 0;
  b->skip_fast[0] = b->skip_fast[1] = b->skip_fast[2] = false;
  b->lap1_age = 0;
  b->flights.drop_all();
  return MOT_OK;
}

int mot_oc_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, const float* p14, mot_oc_batch** out) {
  if (!ctx || !out || nstreams <= 0 || cap_tracks <= 0 || max_dets <= 0) return MOT_ERR_INVALID;
  auto P_ = [&](int i, float d) { return p14 ? p14[i] : d; };
  auto* b = new mot_oc_batch();
  b->ctx = ctx; b->S = nstreams; b->CAP = cap_tracks; b->D = max_dets;
  OcParams& P = b->prm;
  // [det_thresh, max_age, max_obs (unused), min_hits, iou_threshold, min_conf, delta_t, inertia, use_byte, Q_xy, Q_s, asso, w, h]
  P.det_thresh = P_(0, 0.2f); P.max_age = static_cast<int>(P_(1, 30)); P.min_hits = static_cast<int>(P_(3, 3)); P.thr = P_(4, 0.3f);
  P.min_conf = P_(5, 0.1f); P.delta_t = static_cast<int>(P_(6, 3)); P.inertia = P_(7, 0.2f); P.use_byte = P_(8, 0.f) != 0.f;
  const float q_xy = P_(9, 0.01f), q_s = P_(10, 0.0001f);
  P.asso = static_cast<int>(P_(11, 0.f));
  if (P.asso < 0 || P.asso > MOT_ASSOC_CENTROID || P.delta_t < 0 || P.delta_t > 64) { delete b; return MOT_ERR_INVALID; }
  const int iw = static_cast<int>(P_(12, 1920.f)), ih = static_cast<int>(P_(13, 1080.f));
  P.frame_diag = static_cast<float>(std::sqrt(static_cast<double>(iw * iw + ih * ih)));  // iou.hpp:329
  P.K = P.delta_t + 2;
  const int S = nstreams, CAP = cap_tracks, D = max_dets, K = P.K;
  const int UPD = 4 * D + CAP;                 // updates a frame can queue: first + BYTE + OCR matches
  const int UMD = 2 * D, UMT = D + CAP;        // unmatched lists with their Q4 repeats
  const int ldc = (CAP + 3) & ~3;
  const size_t ints_per = static_cast<size_t>(CAP) * (14 + K) + static_cast<size_t>(D) * 5 + static_cast<size_t>(UMD) * 4 + static_cast<size_t>(UMT) +
                          static_cast<size_t>(UPD) * (3 + 2 * kRounds) + 16;
  const size_t floats_per = static_cast<size_t>(CAP) * (23 + 5 * K) + static_cast<size_t>(UMT) * 4 + static_cast<size_t>(D) * 10 + static_cast<size_t>(UMD) + 8;
  const size_t bytes_per = static_cast<size_t>(CAP) * 2 + static_cast<size_t>(D) * 2;
  int* ip = b->dalloc<int>(ints_per * S);
  float* fp = b->dalloc<float>(floats_per * S);
  unsigned char* bp = b->dalloc<unsigned char>(bytes_per * S);
  unsigned char* clamp = b->dalloc<unsigned char>(CAP);
  b->mean = b->dalloc<float>(static_cast<size_t>(S) * 56 * CAP);
  float* mats = b->dalloc<float>(static_cast<size_t>(S) * 2 * D * ldc);  // cost and IoU matrices of the first association
  b->d_streams = b->dalloc<OcStream>(S);
  b->d_counts = b->dalloc<int>(S);
  b->d_err = b->dalloc<int>(1);
  b->d_maxt = b->dalloc<int>(64);
  b->d_alive = b->dalloc<int>(S);
  b->flights.with_alive = true;
  b->d_stats = b->dalloc<unsigned long long>(8 * 64);
  b->d_out = b->dalloc<float>(static_cast<size_t>(S) * CAP * 8);
  b->d_out_counts = b->dalloc<int>(S);
  OcTasks& T = b->tasks;
  T.det = b->dalloc<mot_det_task>(S);
  T.pred = b->dalloc<mot_kf_task>(S); T.init = b->dalloc<mot_kf_task>(S); T.upd = b->dalloc<mot_kf_task>(static_cast<size_t>(kRounds) * S);
  T.sbox = b->dalloc<mot_kf_task>(S);
  T.cost = b->dalloc<mot_ocsort_task>(S);
  T.lap1 = b->dalloc<mot_lap_task>(S); T.lapb = b->dalloc<mot_lap_task>(S); T.lapr = b->dalloc<mot_lap_task>(S);
  const size_t wb1 = (mot::lap_scratch_bytes(D, CAP) + 255) & ~size_t(255);      // rows = detections, columns = tracks
  const size_t wbb = (mot::lap_scratch_bytes(D, UMT) + 255) & ~size_t(255);      // BYTE: low detections x the unmatched-track list
  const size_t wbr = (mot::lap_scratch_bytes(UMD, UMT) + 255) & ~size_t(255);    // OCR: the unmatched lists with repeats
  const size_t wrl = (mot::lap_rowlist_scratch_bytes(D) + 255) & ~size_t(255);   // first association: per-detection lists of the costs below thresh/2
  const size_t wall = wb1 + wbb + wbr + wrl;
  char* work = b->dalloc<char>(wall * S);
  if (!ip || !fp || !bp || !clamp || !b->mean || !mats || !b->d_streams || !b->d_counts || !b->d_err || !b->d_maxt || !b->d_alive || !b->d_stats || !b->d_out ||
      !b->d_out_counts || !T.det || !T.pred || !T.init || !T.upd || !T.sbox || !T.cost || !T.lap1 || !T.lapb || !T.lapr || !work) {
    mot_oc_destroy(b);
    return MOT_ERR_NOMEM;
  }
  (void)hipMemset(b->d_stats, 0, 8 * 64 * sizeof(unsigned long long));
  (void)hipMemset(clamp, MOT_KF_OCSORT_CLAMP, CAP);
  for (auto& e : b->ev) (void)hipEventCreate(&e);
  std::vector<OcStream> hs(S);
  std::vector<mot_det_task> det(S);
  std::vector<mot_kf_task> pred(S), init(S), upd(static_cast<size_t>(kRounds) * S), sbox(S);
  std::vector<mot_ocsort_task> cost(S);
  std::vector<mot_lap_task> lap1(S), lapb(S), lapr(S);
  for (int s = 0; s < S; ++s) {
    OcStream& Z = hs[s];
    std::memset(&Z, 0, sizeof(Z));
    int* i = ip + ints_per * s;
    auto I = [&](size_t n) { int* r = i; i += n; return r; };
    Z.free_stack = I(CAP); Z.trk[0] = I(CAP); Z.trk[1] = I(CAP);
    Z.t_id = I(CAP); Z.t_age = I(CAP); Z.t_hits = I(CAP); Z.t_streak = I(CAP); Z.t_tsu = I(CAP); Z.t_cls = I(CAP); Z.t_det = I(CAP); Z.t_nobs = I(CAP);
    Z.t_need = I(CAP);
    Z.t_oage = I(static_cast<size_t>(CAP) * K);
    Z.slot_cnt = I(CAP); Z.need_slot = I(CAP);
    Z.high = I(D); Z.second = I(D);
    Z.x1 = I(D); Z.xb = I(D); Z.info1 = I(4); Z.infob = I(4); Z.infor = I(4);
    Z.init_dst = I(UMD); Z.init_meas = I(UMD); Z.um_dets = I(UMD); Z.xr = I(UMD); Z.left = I(D);
    Z.um_trks = I(UMT);
    Z.upd_slot = I(UPD); Z.upd_meas = I(UPD); Z.upd_round = I(UPD);
    for (int r = 0; r < kRounds; ++r) { Z.r_slot[r] = I(UPD); Z.r_meas[r] = I(UPD); }
    float* f = fp + floats_per * s;
    auto F = [&](size_t n) { float* r = f; f += n; return r; };
    Z.t_conf = F(CAP); Z.t_last = F(static_cast<size_t>(CAP) * 5); Z.t_vel = F(static_cast<size_t>(CAP) * 2);
    Z.t_obs = F(static_cast<size_t>(CAP) * K * 5);
    Z.pbox = F(4 * CAP); Z.vel = F(2 * CAP); Z.prev = F(5 * CAP); Z.lbox = F(static_cast<size_t>(4) * UMT); Z.sbox = F(4 * CAP);
    float* d_box = F(4 * D); float* d_meas = F(4 * D);
    Z.xval1 = F(D); Z.xvalb = F(D); Z.xvalr = F(UMD);
    unsigned char* u = bp + bytes_per * s;
    Z.mt = u; Z.rm_t = u + CAP; Z.md = u + 2 * CAP; Z.rm_d = u + 2 * CAP + D;
    float* mean = b->mean + static_cast<size_t>(s) * 56 * CAP;
    Z.kmean = mean;
    float* cmat = mats + static_cast<size_t>(s) * 2 * D * ldc;
    float* imat = cmat + static_cast<size_t>(D) * ldc;
    std::memset(&det[s], 0, sizeof(mot_det_task));
    det[s].box = d_box; det[s].ldb = D; det[s].meas = d_meas; det[s].ldm = D;
    auto kf = [&](mot_kf_task& k) {
      std::memset(&k, 0, sizeof(k));
      k.mean = mean; k.cov = mean + 7; k.cap = CAP;
      k.q[0] = 0.01f * q_xy; k.q[1] = 0.01f * q_xy; k.q[2] = 0.0001f * q_s;  // Q5: scaled twice (ocsort.cpp:77-79)
    };
    kf(pred[s]); pred[s].flags = clamp; pred[s].boxes = Z.pbox; pred[s].ldb = CAP;
    kf(init[s]); init[s].src = Z.init_dst; init[s].dst = Z.init_dst; init[s].meas = d_meas; init[s].ldm = D; init[s].midx = Z.init_meas;
    for (int r = 0; r < kRounds; ++r) {
      mot_kf_task& U = upd[static_cast<size_t>(r) * S + s];
      kf(U); U.src = Z.r_slot[r]; U.dst = Z.r_slot[r]; U.meas = d_meas; U.ldm = D; U.midx = Z.r_meas[r];
    }
    kf(sbox[s]); sbox[s].src = Z.need_slot; sbox[s].boxes = Z.sbox; sbox[s].ldb = CAP;
    mot_ocsort_task& C = cost[s];
    std::memset(&C, 0, sizeof(C));
    C.dbox = d_box; C.ldd = D; C.didx = Z.high; C.tbox = Z.pbox; C.ldt = CAP; C.vel = Z.vel; C.ldv = CAP; C.prev = Z.prev; C.ldp = CAP;
    C.vdc_weight = P.inertia; C.cost = cmat; C.iou = imat; C.ldc = ldc; C.assoc = P.asso; C.frame_diag = P.frame_diag;
    // y arrays (track-sided) of the three assignments live in one more int block
    mot_lap_task& L1 = lap1[s];
    std::memset(&L1, 0, sizeof(L1));
    L1.cost = cmat; L1.ldc = ldc; L1.thresh = -P.thr; L1.x = Z.x1; L1.mode = MOT_LAP_OCSORT; L1.iou = imat; L1.ldi = ldc; L1.gate = P.thr;
    L1.xval = Z.xval1; L1.info = Z.info1; L1.work = work + wall * s; L1.rowlist = work + wall * s + wb1 + wbb + wbr;
    auto geom = [&](mot_lap_task& L, int* x, float* xval, int* info, char* w, const int* aidx, const float* bbox, int ldb, const int* bidx) {
      std::memset(&L, 0, sizeof(L));
      L.x = x; L.thresh = -P.thr; L.mode = MOT_LAP_GATE_MIN; L.gate = -P.thr; L.xval = xval; L.info = info; L.work = w;
      L.geom.a = d_box; L.geom.lda = D; L.geom.aidx = aidx; L.geom.b = bbox; L.geom.ldb = ldb; L.geom.bidx = bidx;
      L.geom.mode = MOT_COST_NEG_IOU; L.geom.assoc = P.asso; L.geom.frame_diag = P.frame_diag;
    };
    geom(lapb[s], Z.xb, Z.xvalb, Z.infob, work + wall * s + wb1, Z.second, Z.pbox, CAP, Z.um_trks);
    geom(lapr[s], Z.xr, Z.xvalr, Z.infor, work + wall * s + wb1 + wbb, Z.left, Z.lbox, UMT, nullptr);
  }
  const size_t ys_per = static_cast<size_t>(CAP) + 2 * static_cast<size_t>(UMT);  // y arrays are track-sided
  int* ys = b->dalloc<int>(ys_per * S);
  if (!ys) { mot_oc_destroy(b); return MOT_ERR_NOMEM; }
  for (int s = 0; s < S; ++s) {
    int* y = ys + ys_per * s;
    hs[s].y1 = y; hs[s].yb = y + CAP; hs[s].yr = y + CAP + UMT;
    lap1[s].y = hs[s].y1; lapb[s].y = hs[s].yb; lapr[s].y = hs[s].yr;
  }
  b->h_streams = hs;
  hipStream_t st = ctx->stream;
#define OC_UP(dst, vec) MOT_LC_HIP(b, hipMemcpyAsync(dst, vec.data(), sizeof(vec[0]) * vec.size(), hipMemcpyHostToDevice, st))
  OC_UP(b->d_streams, hs); OC_UP(T.det, det); OC_UP(T.pred, pred); OC_UP(T.init, init); OC_UP(T.upd, upd); OC_UP(T.sbox, sbox);
  OC_UP(T.cost, cost); OC_UP(T.lap1, lap1); OC_UP(T.lapb, lapb); OC_UP(T.lapr, lapr);
#undef OC_UP
  MOT_LC_HIP(b, hipMemsetAsync(b->d_err, 0, sizeof(int), st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  *out = b;
  return MOT_OK;
}

// queues one frame of every stream up to the staged output tables (counts: host memory that stays valid until its copy has run);
// bound = live tracks any stream may have; *declined_out = device counter of the first associations the sparse solver declined
static int oc_enqueue_frame(mot_oc_batch* b, const float* d_dets, const int* counts, int bound, hipEvent_t* ev, int** declined_out /* [3] */,
                            const mot::lifecycle::FrameDev* fd = nullptr) {
  const bool prof = ev != nullptr;
  hipStream_t st = b->ctx->stream;
  const int S = b->S, CAP = b->CAP, D = b->D;
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
  const int bn = (bound < 1) ? 1 : (bound > CAP ? CAP : bound);
  const bool general = b->prm.asso != MOT_ASSOC_IOU;
  const OcTasks& K = b->tasks;
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[0], st));
  hipLaunchKernelGGL(oc_begin, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, D, S, FD, d_dets, K);
  // (round 5: detection preparation in oc_begin, the emitted rows' state boxes and the error word in oc_emit: three launches fewer)
  MOT_LC_HIP(b, mot::launch_kf_op(1, MOT_KF_XYSR, K.pred, S, bn, st));
  hipLaunchKernelGGL(oc_nan, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, K, prof ? b->d_stats : nullptr);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[2], st));
  MOT_LC_HIP(b, mot::launch_ocsort(K.cost, S, bd, bn, !general, st));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[3], st));
  // an association whose problems the sparse solver declined (nine in ten) goes straight to the exact solver, with a retry every eighth frame:
  // with quirk Q4's duplicate tracks no optimum is unique, and a declined attempt costs as much as a successful one
  int active = 0;  // streams with a frame: a launch with a handful of problems is tuned for latency (mot::launch_lap)
  for (int s = 0; s < S; ++s) active += (counts[s] >= 0) ? 1 : 0;
  const bool retry = (b->lap1_age++ % 8) == 0;  // (advanced here, not at collect: two frames queued back to back do not both retry)
  const bool lap1_fast = !b->skip_fast[0] || retry, lapb_fast = !b->skip_fast[1] || retry, lapr_fast = !b->skip_fast[2] || retry;
  int* lap1_declined = nullptr; int* lapb_declined = nullptr; int* lapr_declined = nullptr;
  MOT_LC_HIP(b, mot::launch_lap(K.lap1, S, bd, bn, false, false, true, st, 0, 0, lap1_fast, &lap1_declined, nullptr, nullptr, active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[4], st));
  hipLaunchKernelGGL(oc_after_first, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, D, K);
  if (b->prm.use_byte) {
    MOT_LC_HIP(b, mot::launch_lap(K.lapb, S, bd, (bn + bd > CAP + D) ? CAP + D : bn + bd, true, general, true, st, 0, 0, lapb_fast, &lapb_declined, nullptr, nullptr, active));
    hipLaunchKernelGGL(oc_after_byte, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, D, K);
  }
  MOT_LC_HIP(b, mot::launch_lap(K.lapr, S, 2 * bd, bn + bd, true, general, true, st, 0, 0, lapr_fast, &lapr_declined, nullptr, nullptr, active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[5], st));
  hipLaunchKernelGGL(oc_finish, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, D, S, K);
  MOT_LC_HIP(b, mot::launch_kf_op(0, MOT_KF_XYSR, K.init, S, 2 * bd, st));
  for (int r = 0; r < kRounds; ++r) MOT_LC_HIP(b, mot::launch_kf_op(2, MOT_KF_XYSR, K.upd + static_cast<size_t>(r) * S, S, (r == 0) ? bn : 2 * bd, st));
  hipLaunchKernelGGL(oc_emit, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, b->d_out, b->d_out_counts, CAP, b->d_maxt, b->d_alive, b->d_err);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[1], st));
  MOT_LC_HIP(b, hipGetLastError());
  declined_out[0] = lap1_declined; declined_out[1] = lapb_declined; declined_out[2] = lapr_declined;
  return MOT_OK;
}
// what the host keeps from a finished frame: the launch bound of the next one, whether the sparse solver is worth trying, profile sums
static int oc_account(mot_oc_batch* b, const int* maxt, const int declined[3], hipEvent_t* ev) {
  for (int k = 0; k < 3; ++k)
    if (declined[k] >= 0) b->skip_fast[k] = declined[k] * 10 >= 9 * b->S;
  b->bound_n = 0;
  for (int i = 0; i < 64; ++i) b->bound_n = (maxt[i] > b->bound_n) ? maxt[i] : b->bound_n;
  if (ev) {
    float ms = 0.f;
    MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[0], ev[1])); b->frame_ms += ms;
    MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[2], ev[3])); b->cost_ms += ms;
    MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[3], ev[4])); b->lap_ms += ms;
    b->frames += 1;
  }
  return MOT_OK;
}

// Frames in flight (as mot_bt_enqueue_packed / mot_bt_collect_packed): a frame still in flight may add two tracks per detection
// (quirk Q4: a detection on the unmatched list twice spawns two), which bounds the next frame's launches.
static int oc_enqueue_flight(mot_oc_batch* b, const float* d_dets, const int* h_counts, int rows_cap, const mot_frame_in* in) {
  if (b->flights.count >= 2) { b->ctx->err = "mot_oc_enqueue: two frames are already in flight (collect one first)"; return MOT_ERR_INVALID; }
  const int slot = b->flights.slot_for_enqueue();
  int* counts_in = nullptr;
  int bd = 0;
  MOT_LC_HIP(b, b->flights.prepare(b->mem, slot, b->S, rows_cap, h_counts, &counts_in, &bd, in != nullptr, b->profile));  // (an event set per frame in flight)
  mot::lifecycle::Flight& F = b->flights.fl[slot];
  mot::lifecycle::FrameDev fd;
  if (in) MOT_LC_HIP(b, b->flights.upload_block(b->mem, slot, b->S, in->h_counts, in->h_det_ld, in->h_det_off, nullptr, b->ctx->stream, &fd));
  int* declined[3] = {nullptr, nullptr, nullptr};
  const int rc = oc_enqueue_frame(b, d_dets, counts_in, b->bound_n + 2 * b->flights.pending_bd(), F.prof ? F.ev : nullptr, declined, in ? &fd : nullptr);
  if (rc != MOT_OK) return rc;
  MOT_LC_HIP(b, b->flights.finish(slot, b->ctx->stream, b->d_out, b->CAP, b->d_out_counts, b->S, b->d_err, b->d_maxt, declined[0], rows_cap, bd, declined[1], declined[2], b->d_alive));
  return MOT_OK;
}
static int oc_pop_flight(mot_oc_batch* b, mot::lifecycle::Flight** out, int* total) {
  if (b->flights.count <= 0) { b->ctx->err = "mot_oc_collect: no frame in flight"; return MOT_ERR_INVALID; }
  mot::lifecycle::Flight* F = nullptr;
  MOT_LC_HIP(b, b->flights.pop(&F));
  const int dec[3] = {F->h_meta[mot::lifecycle::kMetaDec], F->h_meta[mot::lifecycle::kMetaDec + 1], F->h_meta[mot::lifecycle::kMetaDec + 2]};
  const int ra = oc_account(b, b->flights.maxt_of(*F), dec, F->prof ? F->ev : nullptr);
  if (ra != MOT_OK) return ra;
  *total = F->h_meta[0];
  *out = F;
  b->d_packed = F->view ? F->h_rows : F->d_packed; b->d_offsets = F->d_offsets;  // mot_oc_device_output: the frame just collected
  b->d_counts_last = F->d_counts;
  if (F->h_meta[1]) { b->ctx->err = "mot_oc_step: a stream exceeded cap_tracks / max_dets"; return MOT_ERR_CAPACITY; }
  if (*total > F->rows_cap) { b->ctx->err = "mot_oc_step_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  return MOT_OK;
}
int mot_oc_enqueue_packed(mot_oc_batch* b, const float* d_dets, const int* h_counts, int rows_cap) {
  if (!b || !d_dets || !h_counts || rows_cap <= 0) return MOT_ERR_INVALID;
  return oc_enqueue_flight(b, d_dets, h_counts, rows_cap, nullptr);
}
int mot_oc_collect_packed(mot_oc_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b || !out_counts) return MOT_ERR_INVALID;  // rows == NULL: the table stays on the device (mot_oc_device_output), only the counts come back
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = oc_pop_flight(b, &F, &total);
  if (F) std::memcpy(out_counts, b->flights.counts_of(*F), sizeof(int) * b->S);
  if (total_rows) *total_rows = total;
  if (rc != MOT_OK) return rc;
  if (!rows) return MOT_OK;
  if (total > rows_cap) { b->ctx->err = "mot_oc_step_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  MOT_LC_HIP(b, b->flights.copy_rows(*F, rows, total));
  return MOT_OK;
}
// ---- pooled form (see mot_bt_enqueue_frame) ----
int mot_oc_enqueue_frame(mot_oc_batch* b, const mot_frame_in* in, int rows_cap) {
  if (!b || !in || !in->d_dets || !in->h_counts || !in->h_det_ld || !in->h_det_off || rows_cap <= 0) return MOT_ERR_INVALID;
  return oc_enqueue_flight(b, in->d_dets, in->h_counts, rows_cap, in);
}
int mot_oc_collect_view(mot_oc_batch* b, mot_frame_view* out) {
  if (!b || !out) return MOT_ERR_INVALID;
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = oc_pop_flight(b, &F, &total);
  if (!F) return rc;
  if (!F->view) { b->ctx->err = "mot_oc_collect_view: the frame was queued with mot_oc_enqueue_packed"; return MOT_ERR_INVALID; }
  out->rows = F->h_rows; out->counts = b->flights.counts_of(*F); out->alive = b->flights.alive_of(*F, b->S); out->total = total;
  return rc;
}
int mot_oc_reset_stream(mot_oc_batch* b, int s, int fresh) {
  if (!b || s < 0 || s >= b->S) return MOT_ERR_INVALID;
  hipLaunchKernelGGL(mot::lifecycle::reset_stream_kernel<OcStream>, dim3(1), dim3(64), 0, b->ctx->stream, b->d_streams, s, b->h_streams[s], fresh ? 0 : 1, b->d_err);
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}
namespace {
__global__ void __launch_bounds__(256) oc_move(const OcStream* from, OcStream* to, const float* mean_from, float* mean_to, int cap, int K) {
  using mot::lifecycle::move_array;
  const OcStream& A = *from;
  OcStream& B = *to;
  const size_t n = static_cast<size_t>(cap);
  move_array(B.free_stack, A.free_stack, n); move_array(B.trk[0], A.trk[A.cur], n);
  move_array(B.t_id, A.t_id, n); move_array(B.t_age, A.t_age, n); move_array(B.t_hits, A.t_hits, n); move_array(B.t_streak, A.t_streak, n);
  move_array(B.t_tsu, A.t_tsu, n); move_array(B.t_cls, A.t_cls, n); move_array(B.t_det, A.t_det, n); move_array(B.t_nobs, A.t_nobs, n);
  move_array(B.t_conf, A.t_conf, n); move_array(B.t_last, A.t_last, n * 5); move_array(B.t_vel, A.t_vel, n * 2);
  move_array(B.t_oage, A.t_oage, n * K); move_array(B.t_obs, A.t_obs, n * K * 5);
  move_array(mean_to, mean_from, n * 56);
  __syncthreads();
  if (threadIdx.x == 0) {
    B.frame_count = A.frame_count; B.next_id = A.next_id; B.next_slot = A.next_slot; B.n_free = A.n_free; B.n_trk = A.n_trk; B.err = A.err;
    B.cur = 0; B.skip = 1;
  }
}
}  // namespace
int mot_oc_move_stream(mot_oc_batch* src, int s, mot_oc_batch* dst, int s2) {
  if (!src || !dst || s < 0 || s >= src->S || s2 < 0 || s2 >= dst->S || dst->CAP < src->CAP || dst->D < src->D || dst->prm.K != src->prm.K) return MOT_ERR_INVALID;
  MOT_LC_HIP(src, hipStreamSynchronize(src->ctx->stream));
  hipStream_t st = dst->ctx->stream;
  hipLaunchKernelGGL(oc_move, dim3(1), dim3(256), 0, st, src->d_streams + s, dst->d_streams + s2, src->mean + static_cast<size_t>(s) * 56 * src->CAP,
                     dst->mean + static_cast<size_t>(s2) * 56 * dst->CAP, src->CAP, src->prm.K);
  MOT_LC_HIP(dst, hipGetLastError());
  OcStream h;
  MOT_LC_HIP(dst, hipMemcpyAsync(&h, dst->d_streams + s2, sizeof(OcStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(dst, hipStreamSynchronize(st));
  if (h.n_trk > dst->bound_n) dst->bound_n = h.n_trk;
  return MOT_OK;
}
int mot_oc_step_packed(mot_oc_batch* b, const float* d_dets, const int* h_counts, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b || !d_dets || !h_counts || !rows || !out_counts) return MOT_ERR_INVALID;
  if (b->flights.count > 0) { b->ctx->err = "mot_oc_step_packed: frames are in flight (collect them first)"; return MOT_ERR_INVALID; }
  const int rc = mot_oc_enqueue_packed(b, d_dets, h_counts, rows_cap);
  return (rc != MOT_OK) ? rc : mot_oc_collect_packed(b, rows, rows_cap, out_counts, total_rows);
}

int mot_oc_device_output(mot_oc_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts) {
  if (!b || !b->d_packed || !b->d_offsets) return MOT_ERR_INVALID;
  if (d_rows) *d_rows = b->d_packed;
  if (d_offsets) *d_offsets = b->d_offsets;
  if (d_counts) *d_counts = b->d_counts_last;
  return MOT_OK;
}

int mot_oc_profile(mot_oc_batch* b, int enable) {
  b->profile = enable != 0;
  if (enable) {
    b->lap_ms = b->cost_ms = b->frame_ms = 0.0;
    b->frames = 0;
    MOT_LC_HIP(b, hipMemsetAsync(b->d_stats, 0, 8 * 64 * sizeof(unsigned long long), b->ctx->stream));
    MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));
  }
  return MOT_OK;
}

int mot_oc_profile_stats(mot_oc_batch* b, double* out8) {
  unsigned long long raw[8 * 64];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long h[2] = {0, 0};
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 2; ++k) h[k] += raw[i * 8 + k];
  out8[0] = b->lap_ms; out8[1] = b->cost_ms; out8[2] = b->frame_ms; out8[3] = static_cast<double>(b->frames);
  out8[4] = static_cast<double>(h[0]); out8[5] = static_cast<double>(h[1]); out8[6] = 0.0; out8[7] = 0.0;
  return MOT_OK;
}

int mot_oc_dump(mot_oc_batch* b, int s, int* ids, float* mean, float* cov, int cap) {
  hipStream_t st = b->ctx->stream;
  OcStream h;
  MOT_LC_HIP(b, hipMemcpyAsync(&h, b->d_streams + s, sizeof(OcStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  const int n = h.n_trk;
  if (n > cap) return -n;
  const int CAP = b->CAP;
  std::vector<int> slots(n), tid(CAP);
  if (n) MOT_LC_HIP(b, hipMemcpyAsync(slots.data(), h.trk[h.cur], sizeof(int) * n, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(tid.data(), h.t_id, sizeof(int) * CAP, hipMemcpyDeviceToHost, st));
  std::vector<float> m(static_cast<size_t>(56) * CAP);
  MOT_LC_HIP(b, hipMemcpyAsync(m.data(), b->mean + static_cast<size_t>(s) * 56 * CAP, sizeof(float) * m.size(), hipMemcpyDeviceToHost, st));
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
