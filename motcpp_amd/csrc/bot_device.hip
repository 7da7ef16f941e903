// BoT-SORT with the per-stream lifecycle ON THE DEVICE (reference: src/trackers/botsort.cpp:260-764; the host-side stage
// machine with the same semantics is host/botsort.cpp).
//
// Same construction as bt_device.hip: the list bookkeeping of BotSort::update runs in four small kernels, one wavefront
// per stream, between the numeric kernels — XYWH Kalman predict (in place, optionally followed by the camera-motion warp),
// the cosine distance of the smooth track features against the frame's normalised detection features on the fp32 matrix
// cores, three assignments whose gated IoU/appearance cost is recomputed inside the solver, Kalman initiate/update, and the
// feature maintenance (set / exponential moving average + renormalise). A frame of S streams is a FIXED launch sequence
// with no host decision in between. Track features live in a per-stream slab [cap_tracks][emb_dim] indexed by Kalman slot.
//
// What BoT-SORT does differently from ByteTrack (and is reproduced here): an empty frame is a no-op (not even the frame
// counter moves, a pending warp is dropped, :267-269); the pool is predicted in place; second-stage detections carry no
// feature; a lost track that is re-found is updated and then DROPPED (prepare_output :678-764 never moves it back to the
// tracked list); unmatched unconfirmed tracks are removed; births need conf >= new_track_thresh; there is no duplicate
// removal between the tracked and the lost list.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "lifecycle_common.hpp"

namespace mot {
hipError_t launch_feat(const mot_feat_task*, int, int, hipStream_t);
hipError_t launch_embed(int metric, const mot_cos_task*, int, int, int, hipStream_t);
hipError_t launch_embed_gated(const mot_cos_task*, const mot_lap_task*, int, int, int, int, hipStream_t);
}  // namespace mot

namespace {
using mot::lifecycle::compact;
using mot::lifecycle::kW;
using mot::lifecycle::FrameDev;

enum St { New = 0, Tracked = 1, Lost = 2, Removed = 3 };

struct BotParams {
  float hi, lo, newt, match, prox, app;
  int fuse_first, with_reid, max_time_lost, E;
};

struct BotStream {
  // ---- persistent ----
  int frame_count, next_id, next_slot, n_free, n_active, n_lost, err;
  int* free_stack;
  int* active[2]; int* lost[2]; int cur;
  int *t_id, *t_state, *t_act, *t_tlen, *t_fid, *t_sf, *t_cls, *t_det, *t_feat;
  float* t_conf;
  // ---- frame input ----
  const float* dets; int ld, n, idle, have_emb, warp;
  const float* embs;  // raw detection features [n][E] of this frame (device)
  // ---- frame scratch ----
  int *first, *second; int n_first, n_second;
  int* pool_slot; int n_pool, n_tracked;
  int* unconf_slot; int n_unconf;
  int *x1, *y1, *x2, *y2, *x3, *y3;
  int *upd_slot, *upd_meas; int n_upd;
  int *set_slot, *set_det; int n_set;
  int *ema_slot, *ema_det; int n_ema;
  int* u_track; int n_utrack;
  int* u_det; int n_udet;
  int* r_pool; int n_r;
  int* rem;
  int lap2_q, lap3_q;
  int *init_dst, *init_meas; int n_init;
  int* lost_new;
  float* abox;  // [4][CAP] (round 5: unused — bot_finish computes an output row's box from the updated mean)
  const float* kmean;  // this stream's Kalman records (72 floats per slot)
};

// every task descriptor of one stream (device arrays of S entries each), so that the kernels take one pointer
struct BotTasks {
  mot_det_task* det;
  mot_feat_task *featn, *fset, *fema;
  mot_kf_task *warp, *pred, *predw, *ubox, *init, *upd, *obox;
  mot_cos_task *cos1, *cos3;
  mot_lap_task *lap1, *lap23;
};

// BotSTrack::update :133-156 / re_activate :111-131 for one matched (slot, det); with_feature: first-stage detection
__device__ __forceinline__ void apply_match(BotStream& S, int slot, int det) {
  if (S.t_state[slot] == Tracked) S.t_tlen[slot] += 1; else S.t_tlen[slot] = 0;
  S.t_fid[slot] = S.frame_count;
  S.t_state[slot] = Tracked; S.t_act[slot] = 1;
  S.t_conf[slot] = S.dets[static_cast<size_t>(4) * S.ld + det];
  S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
  S.t_det[slot] = det;
}

// ---- K0: empty-frame rule, detection split, pools, predict / warp / first-association tasks (:267-330) ----
__global__ void __launch_bounds__(kW) bot_begin(BotStream* streams, BotParams P, int CAP, int D, FrameDev FD, const float* dets_base,
                                                 const float* embs_base, const float* warps6, const int* has_warp, BotTasks K,
                                                 unsigned long long* stats, int* maxt) {
  const int s = blockIdx.x;
  BotStream& S = streams[s];
  const int t = static_cast<int>(threadIdx.x);
  const int n = FD.counts[s];
  int ldd = D;
  const float* dets = (n > 0) ? mot::lifecycle::frame_dets(FD, dets_base, s, D, ldd) : dets_base;
  if (n <= 0 || n > D) {  // :267-269: nothing happens, not even frame_count++ (nor the camera-motion step); n < 0 (pooled form): not this stream's frame
    if (t == 0) {
      S.idle = 1; S.n = 0;
      if (n > D) S.err = 1;
      K.det[s].n = 0; K.featn[s].n = 0; K.fset[s].n = 0; K.fema[s].n = 0;
      K.warp[s].n = 0; K.pred[s].n = 0; K.predw[s].n = 0; K.ubox[s].n = 0; K.init[s].n = 0; K.upd[s].n = 0; K.obox[s].n = 0;
      K.cos1[s].n = 0; K.cos1[s].m = 0; K.cos3[s].n = 0; K.cos3[s].m = 0;
      K.lap1[s].n = 0; K.lap1[s].m = 0; K.lap1[s].geom.n = 0; K.lap1[s].geom.m = 0;
      for (int k = 0; k < 2; ++k) { mot_lap_task& L = K.lap23[2 * s + k]; L.n = 0; L.m = 0; L.geom.n = 0; L.geom.m = 0; }
    }
    return;
  }
  const long long eoff = FD.emb_off ? FD.emb_off[s] : static_cast<long long>(s) * D * P.E;  // (pooled form: < 0 = no features for this stream)
  const int have_emb = (P.with_reid && embs_base != nullptr && P.E > 0 && eoff >= 0) ? 1 : 0;
  const int warp = (has_warp != nullptr && has_warp[s] != 0) ? 1 : 0;
  const float* embs = have_emb ? embs_base + eoff : nullptr;
  if (t == 0) {
    S.idle = 0;
    S.frame_count += 1;
    S.dets = dets; S.ld = ldd; S.n = n; S.have_emb = have_emb; S.warp = warp; S.embs = embs;
  }
  const float* conf = dets + static_cast<size_t>(4) * ldd;
  float* dbox = K.det[s].box;
  float* dmeas = K.det[s].meas;
  const int dldb = K.det[s].ldb, dldm = K.det[s].ldm;
  int nf = 0, ns = 0;
  for (int i0 = 0; i0 < n; i0 += kW) {  // :283-300
    const int i = i0 + t;
    const float c = (i < n) ? conf[i] : 0.f;
    const bool hi = i < n && c > P.hi;
    const bool lo = i < n && !(c > P.hi) && c > P.lo;
    const int ph = compact(hi, nf);
    if (hi) S.first[ph] = i;
    const int pl = compact(lo, ns);
    if (lo) S.second[pl] = i;
    if (i < n) {  // the detection's association box and Kalman measurement (det_kernel<MOT_DET_XYWH>, botsort.cpp:23-36, 171-181: the same operations)
      const float x1 = dets[i], y1 = dets[static_cast<size_t>(ldd) + i], x2 = dets[static_cast<size_t>(2) * ldd + i], y2 = dets[static_cast<size_t>(3) * ldd + i];
      const float w = x2 - x1, h = y2 - y1;
      const float cx = x1 + w / 2.0f, cy = y1 + h / 2.0f;
      const float zz[4] = {cx, cy, w, h};
      const float bb[4] = {cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2};
#pragma unroll
      for (int q = 0; q < 4; ++q) { dbox[static_cast<size_t>(q) * dldb + i] = bb[q]; dmeas[static_cast<size_t>(q) * dldm + i] = zz[q]; }
    }
  }
  const int* act = S.active[S.cur];
  const int* lst = S.lost[S.cur];
  int np = 0, nu = 0;
  for (int i0 = 0; i0 < S.n_active; i0 += kW) {  // unconfirmed / tracked :302-312
    const int i = i0 + t;
    const int slot = (i < S.n_active) ? act[i] : 0;
    const bool a = i < S.n_active && S.t_act[slot] != 0;
    const bool u = i < S.n_active && S.t_act[slot] == 0;
    const int pa = compact(a, np);
    if (a) S.pool_slot[pa] = slot;
    const int pu = compact(u, nu);
    if (u) S.unconf_slot[pu] = slot;
  }
  const int n_tracked = np;
  for (int i0 = 0; i0 < S.n_lost; i0 += kW) {  // joint_stracks(tracked, lost) :314 (the two lists are disjoint by id)
    const int i = i0 + t;
    const bool v = i < S.n_lost;
    const int p = compact(v, np);
    if (v) S.pool_slot[p] = lst[i];
  }
  if (t == 0) {
    S.n_first = nf; S.n_second = ns; S.n_pool = np; S.n_tracked = n_tracked; S.n_unconf = nu;
    S.n_upd = 0; S.n_set = 0; S.n_ema = 0; S.n_utrack = 0; S.n_udet = 0; S.n_r = 0; S.n_init = 0; S.lap2_q = 0; S.lap3_q = 0;
    K.det[s].dets = dets; K.det[s].ld = ldd; K.det[s].n = n;
    mot_feat_task& FN = K.featn[s];  // normalised copies of every detection feature (:38-46)
    FN.n = have_emb ? n : 0; FN.src = embs;
    K.fset[s].src = embs; K.fema[s].src = embs; K.fset[s].n = 0; K.fema[s].n = 0;
    // camera motion: the unconfirmed tracks are warped as they are, the pool after its predict (:317-324)
    mot_kf_task& W = K.warp[s];
    mot_kf_task& PW = K.predw[s];
    W.n = warp ? nu : 0;
    K.pred[s].n = warp ? 0 : np;
    PW.n = warp ? np : 0;
    if (warp) {
      const float* w = warps6 + static_cast<size_t>(s) * 6;
      for (int k = 0; k < 6; ++k) { W.warp[k] = w[k]; PW.warp[k] = w[k]; }
      W.warp[6] = 0.f; W.warp[7] = 0.f; W.warp[8] = 1.f; PW.warp[6] = 0.f; PW.warp[7] = 0.f; PW.warp[8] = 1.f;
    }
    const bool q = np > 0 && nf > 0;
    if (q) atomicMax(&maxt[64 + (s & 63)], np);
    mot_cos_task& C1 = K.cos1[s];
    C1.n = (q && have_emb) ? np : 0; C1.m = (q && have_emb) ? nf : 0;
    mot_lap_task& L = K.lap1[s];
    L.n = q ? np : 0; L.m = q ? nf : 0; L.geom.n = L.n; L.geom.m = L.m;
    L.geom.bconf = conf;
    // appearance term: the matrix when there are features, the constant 1 when ReID is on but the frame has none
    L.geom.emb = have_emb ? C1.out : nullptr;
    L.geom.lde = have_emb ? C1.ldo : (P.with_reid ? -1 : 0);
    K.ubox[s].n = 0; K.cos3[s].n = 0; K.cos3[s].m = 0; K.init[s].n = 0; K.upd[s].n = 0; K.obox[s].n = 0;
    for (int k = 0; k < 2; ++k) { mot_lap_task& L2 = K.lap23[2 * s + k]; L2.n = 0; L2.m = 0; L2.geom.n = 0; L2.geom.m = 0; }
    if (stats && q) {
      unsigned long long* st = stats + (s & 63) * 8;
      atomicAdd(&st[0], 1ull); atomicAdd(&st[1], static_cast<unsigned long long>(np + nf));
      if (have_emb) atomicAdd(&st[2], static_cast<unsigned long long>(np) * static_cast<unsigned long long>(nf));
    }
    if (stats && have_emb) atomicAdd(&stats[(s & 63) * 8 + 3], 2ull * static_cast<unsigned long long>(n));  // feature rows moved: normalise = read + write
  }
}

// ---- K1: apply the first association, queue the second and the unconfirmed one (:332-563) ----
__global__ void __launch_bounds__(kW) bot_after_first(BotStream* streams, BotParams P, int CAP, BotTasks K, unsigned long long* stats, int* maxt) {
  const int s = blockIdx.x;
  BotStream& S = streams[s];
  if (S.idle) return;
  const int t = static_cast<int>(threadIdx.x);
  const int np = S.n_pool, nd = S.n_first;
  const bool have = np > 0 && nd > 0;
  int n_upd = 0, n_set = 0, n_ema = 0, n_ut = 0, n_ud = 0;
  for (int i0 = 0; i0 < np; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < np;
    const int x = (v && have) ? S.x1[i] : -1;
    const int slot = v ? S.pool_slot[i] : 0;
    const bool m = v && x >= 0;
    const int det = m ? S.first[x] : 0;
    const int pu = compact(m, n_upd);
    if (m) { S.upd_slot[pu] = slot; S.upd_meas[pu] = det; apply_match(S, slot, det); }
    // update_features :158-169: the first feature is set, later ones are blended
    const bool f = m && S.have_emb != 0;
    const bool had = f && S.t_feat[slot] != 0;
    const int pe = compact(had, n_ema);
    if (had) { S.ema_slot[pe] = slot; S.ema_det[pe] = det; }
    const bool fresh = f && !had;
    const int ps = compact(fresh, n_set);
    if (fresh) { S.set_slot[ps] = slot; S.set_det[ps] = det; S.t_feat[slot] = 1; }
    const bool um = v && x < 0;
    const int pt = compact(um, n_ut);
    if (um) S.u_track[pt] = i;
  }
  for (int j0 = 0; j0 < nd; j0 += kW) {
    const int j = j0 + t;
    const bool u = j < nd && (!have || S.y1[j] < 0);
    const int p = compact(u, n_ud);
    if (u) S.u_det[p] = j;
  }
  __syncthreads();
  int n_r = 0;  // second association: the unmatched pool members that are still Tracked (:497-505)
  for (int k0 = 0; k0 < n_ut; k0 += kW) {
    const int k = k0 + t;
    const int i = (k < n_ut) ? S.u_track[k] : 0;
    const int slot = (k < n_ut) ? S.pool_slot[i] : 0;
    const bool r = k < n_ut && S.t_state[slot] == Tracked;
    const int p = compact(r, n_r);
    if (r) S.r_pool[p] = i;
  }
  for (int k = t; k < n_ud; k += kW) S.rem[k] = S.first[S.u_det[k]];
  __syncthreads();
  if (t == 0) {
    S.n_upd = n_upd; S.n_set = n_set; S.n_ema = n_ema; S.n_utrack = n_ut; S.n_udet = n_ud; S.n_r = n_r;
    const bool q2 = S.n_second > 0 && n_r > 0;
    const bool q3 = S.n_unconf > 0 && n_ud > 0;
    S.lap2_q = q2; S.lap3_q = q3;
    mot_lap_task& A = K.lap23[2 * s + 0];
    A.n = q2 ? n_r : 0; A.m = q2 ? S.n_second : 0; A.geom.n = A.n; A.geom.m = A.m;
    K.ubox[s].n = q3 ? S.n_unconf : 0;
    mot_cos_task& C3 = K.cos3[s];
    C3.n = (q3 && S.have_emb) ? S.n_unconf : 0; C3.m = (q3 && S.have_emb) ? n_ud : 0;
    mot_lap_task& B = K.lap23[2 * s + 1];
    B.n = q3 ? S.n_unconf : 0; B.m = q3 ? n_ud : 0; B.geom.n = B.n; B.geom.m = B.m;
    B.geom.bconf = S.dets + static_cast<size_t>(4) * S.ld;
    B.geom.emb = S.have_emb ? C3.out : nullptr;
    B.geom.lde = S.have_emb ? C3.ldo : (P.with_reid ? -1 : 0);
    {
      const int mn = (A.n > B.n) ? A.n : B.n, mm = (A.m > B.m) ? A.m : B.m;
      if (mn > 0) { atomicMax(&maxt[128 + (s & 63)], mn); atomicMax(&maxt[192 + (s & 63)], mm); }
    }
    if (stats) {
      unsigned long long* st = stats + (s & 63) * 8;
      const int cnt = (q2 ? 1 : 0) + (q3 ? 1 : 0);
      if (cnt) { atomicAdd(&st[0], static_cast<unsigned long long>(cnt)); atomicAdd(&st[1], static_cast<unsigned long long>(A.n + A.m + B.n + B.m)); }
      if (q3 && S.have_emb) atomicAdd(&st[2], static_cast<unsigned long long>(S.n_unconf) * static_cast<unsigned long long>(n_ud));
    }
  }
}

// ---- K2: apply associations 2 and 3, births, deaths, list algebra, queue the Kalman and feature work (:507-764) ----
__global__ void __launch_bounds__(kW) bot_after_second(BotStream* streams, BotParams P, int CAP, BotTasks K, unsigned long long* stats) {
  const int s = blockIdx.x;
  BotStream& S = streams[s];
  if (S.idle) return;
  const int t = static_cast<int>(threadIdx.x);
  int n_upd = S.n_upd, n_set = S.n_set, n_ema = S.n_ema, n_ln = 0;
  if (S.lap2_q) {
    for (int i0 = 0; i0 < S.n_r; i0 += kW) {
      const int i = i0 + t;
      const bool v = i < S.n_r;
      const int slot = v ? S.pool_slot[S.r_pool[i]] : 0;
      const int j = v ? S.x2[i] : -1;
      const bool m = v && j >= 0;
      const int pu = compact(m, n_upd);
      if (m) { const int det = S.second[j]; S.upd_slot[pu] = slot; S.upd_meas[pu] = det; apply_match(S, slot, det); }  // no feature (:533-541)
      const bool l = v && j < 0 && S.t_state[slot] != Lost;
      const int pl = compact(l, n_ln);
      if (l) { S.t_state[slot] = Lost; S.lost_new[pl] = slot; }
    }
  }
  int* udf = S.y3;  // detections nobody took (positions in `first`), written over y3 once it has been read
  int n_udf = 0;
  if (S.lap3_q) {
    for (int j0 = 0; j0 < S.n_udet; j0 += kW) {
      const int j = j0 + t;
      const bool u = j < S.n_udet && S.y3[j] < 0;
      const int val = (j < S.n_udet) ? S.u_det[j] : 0;
      __syncthreads();
      const int p = compact(u, n_udf);  // p <= j: never overwrites an unread entry
      if (u) udf[p] = val;
      __syncthreads();
    }
    for (int i0 = 0; i0 < S.n_unconf; i0 += kW) {
      const int i = i0 + t;
      const bool v = i < S.n_unconf;
      const int slot = v ? S.unconf_slot[i] : 0;
      const int j = v ? S.x3[i] : -1;
      const bool m = v && j >= 0;
      const int det = m ? S.first[S.u_det[j]] : 0;
      const int pu = compact(m, n_upd);
      if (m) { S.upd_slot[pu] = slot; S.upd_meas[pu] = det; apply_match(S, slot, det); }
      const bool f = m && S.have_emb != 0;
      const bool had = f && S.t_feat[slot] != 0;
      const int pe = compact(had, n_ema);
      if (had) { S.ema_slot[pe] = slot; S.ema_det[pe] = det; }
      const bool fresh = f && !had;
      const int ps = compact(fresh, n_set);
      if (fresh) { S.set_slot[ps] = slot; S.set_det[ps] = det; S.t_feat[slot] = 1; }
      if (v && j < 0) S.t_state[slot] = Removed;  // :641-646
    }
  } else {
    for (int j = t; j < S.n_udet; j += kW) udf[j] = S.u_det[j];
    n_udf = S.n_udet;
  }
  __syncthreads();
  // births (:649-667): ids in list order
  int n_init = 0;
  int free_top = S.n_free, next_slot = S.next_slot, err = 0;
  for (int j0 = 0; j0 < n_udf; j0 += kW) {
    const int k = j0 + t;
    const int det = (k < n_udf) ? S.first[udf[k]] : 0;
    const float c = (k < n_udf) ? S.dets[static_cast<size_t>(4) * S.ld + det] : 0.f;
    const bool b = k < n_udf && !(c < P.newt);
    const int base0 = n_init;
    const int p = compact(b, n_init);
    const int births = n_init - base0;
    if (b) {
      const int r = p - base0;
      int slot;
      if (r < free_top) slot = S.free_stack[free_top - 1 - r];
      else { slot = next_slot + (r - free_top); if (slot >= CAP) { slot = CAP - 1; err = 1; } }
      S.t_id[slot] = S.next_id + p + 1;
      S.t_conf[slot] = c;
      S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
      S.t_det[slot] = det;
      S.t_tlen[slot] = 0; S.t_state[slot] = Tracked;
      S.t_act[slot] = (S.frame_count == 1) ? 1 : 0;
      S.t_fid[slot] = S.frame_count; S.t_sf[slot] = S.frame_count;
      S.t_feat[slot] = S.have_emb;
      S.init_dst[p] = slot; S.init_meas[p] = det;
    }
    const int from_free = (births < free_top) ? births : free_top;
    next_slot += births - from_free;
    free_top -= from_free;
  }
  err = __any(err) ? 1 : 0;
  if (S.have_emb) {  // the new tracks' features (BotSTrack ctor :38-46), after the matched ones
    if (n_set + n_init > CAP) err = 1;
    else {
      for (int i = t; i < n_init; i += kW) { S.set_slot[n_set + i] = S.init_dst[i]; S.set_det[n_set + i] = S.init_meas[i]; }
      n_set += n_init;
    }
  }
  const int* lst = S.lost[S.cur];
  const int* act = S.active[S.cur];
  for (int i = t; i < S.n_lost; i += kW) {  // :669-676 (a lost track matched this frame has t_fid == frame_count)
    const int slot = lst[i];
    if (S.frame_count - S.t_fid[slot] > P.max_time_lost) S.t_state[slot] = Removed;
  }
  __syncthreads();
  // prepare_output :678-764
  int* na = S.active[S.cur ^ 1];
  int* nl = S.lost[S.cur ^ 1];
  int n_na = 0, n_nl = 0;
  for (int i0 = 0; i0 < S.n_active; i0 += kW) {
    const int i = i0 + t;
    const int slot = (i < S.n_active) ? act[i] : 0;
    const int st = (i < S.n_active) ? S.t_state[slot] : -1;
    const bool k = st == Tracked;
    const int p = compact(k, n_na);
    if (k) na[p] = slot;
    const bool dead = st == Removed;
    const int pf = compact(dead, free_top);
    if (dead) S.free_stack[pf] = slot;
  }
  if (n_na + n_init > CAP) err = 1;
  else {
    for (int i = t; i < n_init; i += kW) na[n_na + i] = S.init_dst[i];
    n_na += n_init;
  }
  for (int i0 = 0; i0 < S.n_lost; i0 += kW) {
    const int i = i0 + t;
    const int slot = (i < S.n_lost) ? lst[i] : 0;
    const int st = (i < S.n_lost) ? S.t_state[slot] : -1;
    const bool k = st == Lost;
    const int p = compact(k, n_nl);
    if (k) nl[p] = slot;
    const bool dead = st == Removed || st == Tracked;  // aged out, or re-found: updated this frame and then dropped
    const int pf = compact(dead, free_top);
    if (dead) S.free_stack[pf] = slot;
  }
  if (n_nl + n_ln > CAP) err = 1;
  else {
    for (int i = t; i < n_ln; i += kW) nl[n_nl + i] = S.lost_new[i];
    n_nl += n_ln;
  }
  __syncthreads();
  if (t == 0) {
    S.n_upd = n_upd; S.n_set = n_set; S.n_ema = n_ema; S.n_init = n_init;
    S.next_id += n_init; S.next_slot = next_slot; S.n_free = free_top;
    S.n_active = n_na; S.n_lost = n_nl; S.cur ^= 1;
    if (err) S.err = 1;
    K.init[s].n = n_init;
    K.upd[s].n = n_upd;
    K.fset[s].n = S.have_emb ? n_set : 0;
    K.fema[s].n = S.have_emb ? n_ema : 0;
    if (stats && S.have_emb)  // feature rows moved: set = read + write, moving average = two reads + write
      atomicAdd(&stats[(s & 63) * 8 + 3], 2ull * static_cast<unsigned long long>(n_set) + 3ull * static_cast<unsigned long long>(n_ema));
    mot_kf_task& OB = K.obox[s];
    OB.n = n_na; OB.src = na;
  }
}

// ---- K3: the output table (:742-764): activated members of the new tracked list, in list order ----
__global__ void __launch_bounds__(kW) bot_finish(BotStream* streams, int CAP, float* out, int* out_counts, int cap_out, int* max_tracks, int* alive, int* err) {
  BotStream& S = streams[blockIdx.x];
  const int t = static_cast<int>(threadIdx.x);
  if (S.idle) {
    if (t == 0) {
      out_counts[blockIdx.x] = 0; alive[blockIdx.x] = S.err ? -S.err : S.n_active + S.n_lost; atomicMax(&max_tracks[blockIdx.x & 63], S.n_active + S.n_lost);
      if (S.err) atomicMax(err, S.err);
    }
    return;
  }
  const int* act = S.active[S.cur];
  float* rows = out + static_cast<size_t>(blockIdx.x) * cap_out * 8;
  int n_rows = 0;
  for (int i0 = 0; i0 < S.n_active; i0 += kW) {
    const int i = i0 + t;
    const bool v = i < S.n_active;
    const int slot = v ? act[i] : 0;
    const bool emit = v && S.t_act[slot] != 0;
    const int pr = compact(emit, n_rows);
    if (emit && pr < cap_out) {
      float* r = rows + static_cast<size_t>(pr) * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = 0.f;
      {  // the box of the updated state (kf_kernels.hip::s8_box<MOT_KF_XYWH>): centre -/+ half the size
        const float4 m = *reinterpret_cast<const float4*>(S.kmean + static_cast<size_t>(slot) * 72);
        r[0] = m.x - m.z * 0.5f; r[1] = m.y - m.w * 0.5f; r[2] = m.x + m.z * 0.5f; r[3] = m.y + m.w * 0.5f;
      }
      r[4] = static_cast<float>(S.t_id[slot]); r[5] = S.t_conf[slot];
      r[6] = static_cast<float>(S.t_cls[slot]); r[7] = static_cast<float>(S.t_det[slot]);
    }
  }
  if (t == 0) {
    if (n_rows > cap_out) S.err = 2;
    out_counts[blockIdx.x] = (n_rows <= cap_out) ? n_rows : -n_rows;
    atomicMax(&max_tracks[blockIdx.x & 63], S.n_active + S.n_lost);
    alive[blockIdx.x] = S.err ? -S.err : S.n_active + S.n_lost;  // (a stream in error reports -(error code): its caller alone gets the error)
    if (S.err) atomicMax(err, S.err);  // the batch's error word (round 5: gathered here; a kernel of its own before)
  }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------
struct mot_bot_batch {
  mot_ctx* ctx = nullptr;
  int S = 0, CAP = 0, D = 0, E = 0, ldE = 0;
  BotParams prm{};
  mot::lifecycle::Allocs mem;
  BotStream* d_streams = nullptr;
  std::vector<BotStream> h_streams;
  BotTasks tasks{};
  int* d_counts = nullptr;
  int* d_err = nullptr;
  int* d_maxt = nullptr;  // [4][64]: tracks alive, pool rows of the first association, rows / columns of the second and third
  int hint1_n = 0, hint23_n = 0, hint23_m = 0;  // LDS hints for the next frame's assignment launches
  float* d_warps = nullptr; int* d_has_warp = nullptr;
  int bound_n = 0;
  float* d_out = nullptr; int* d_out_counts = nullptr;
  float* d_packed = nullptr; int* d_offsets = nullptr; int packed_cap = 0;
  mot::lifecycle::Flights flights;  // frames in flight (lifecycle_common.hpp); page-locked tail per stream: has_warp + 6 warp floats
  int* d_alive = nullptr;
  const float* d_rows_last = nullptr; const int* d_offsets_last = nullptr; const int* d_counts_last = nullptr;
  mot::lifecycle::PackMeta pack_meta;  // set by mot_bot_enqueue_packed around its frame (empty: the packed tables only)
  float* mean = nullptr;   // [S][CAP] Kalman records (8 + 64 floats)
  float* feat = nullptr;   // [S][CAP][E] smooth features
  bool profile = false;
  unsigned long long* d_stats = nullptr;
  hipEvent_t ev[12] = {};
  double lap_ms = 0.0, cos_ms = 0.0, frame_ms = 0.0, feat_ms = 0.0;
  long frames = 0;
  template <class T>
  T* dalloc(size_t n) { return mem.get<T>(n); }
};

extern "C" {

void mot_bot_destroy(mot_bot_batch* b) {
  if (!b) return;
  b->flights.release();
  b->mem.release();
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  delete b;
}

int mot_bot_reset(mot_bot_batch* b) {  // BotSort::reset :252-258: ids restart
  std::vector<BotStream> h = b->h_streams;
  MOT_LC_HIP(b, hipMemcpyAsync(b->d_streams, h.data(), sizeof(BotStream) * b->S, hipMemcpyHostToDevice, b->ctx->stream));
  MOT_LC_HIP(b, hipMemsetAsync(b->d_err, 0, sizeof(int), b->ctx->stream));
  MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));  // (frames still in flight have finished by now: they are dropped with the tracks)
  b->flights.drop_all();
  b->bound_n = 0; b->hint1_n = b->hint23_n = b->hint23_m = 0;
  return MOT_OK;
}

int mot_bot_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, int emb_dim, const float* p10, mot_bot_batch** out) {
  if (!ctx || !out || nstreams <= 0 || cap_tracks <= 0 || max_dets <= 0 || emb_dim < 0) return MOT_ERR_INVALID;
  auto* b = new mot_bot_batch();
  b->ctx = ctx; b->S = nstreams; b->CAP = cap_tracks; b->D = max_dets; b->E = emb_dim; b->ldE = (max_dets + 3) & ~3;
  BotParams& P = b->prm;
  P.hi = p10 ? p10[0] : 0.5f; P.lo = p10 ? p10[1] : 0.1f; P.newt = p10 ? p10[2] : 0.6f;
  const int track_buffer = p10 ? static_cast<int>(p10[3]) : 30;
  P.match = p10 ? p10[4] : 0.8f; P.prox = p10 ? p10[5] : 0.5f; P.app = p10 ? p10[6] : 0.25f;
  const int frame_rate = p10 ? static_cast<int>(p10[7]) : 30;
  P.fuse_first = p10 ? (p10[8] != 0.f) : 0; P.with_reid = p10 ? (p10[9] != 0.f) : 1;  // (ReID on without features in a frame: the cosine term is the constant 1)
  P.max_time_lost = static_cast<int>(frame_rate / 30.0f * track_buffer);  // botsort.cpp:236-237
  P.E = emb_dim;
  const int S = nstreams, CAP = cap_tracks, D = max_dets, E = emb_dim, ldE = b->ldE;
  const size_t ints_per = static_cast<size_t>(CAP) * 26 + static_cast<size_t>(D) * 9;
  int* ip = b->dalloc<int>(ints_per * S);
  const size_t floats_per = static_cast<size_t>(CAP) * (1 + 4 * 3) + static_cast<size_t>(D) * 8;
  float* fp = b->dalloc<float>(floats_per * S);
  b->mean = b->dalloc<float>(static_cast<size_t>(S) * 72 * CAP);
  float* emb_norm = nullptr;
  float* emb_out = nullptr;
  if (E > 0) {
    b->feat = b->dalloc<float>(static_cast<size_t>(S) * CAP * E);
    emb_norm = b->dalloc<float>(static_cast<size_t>(S) * D * E);
    emb_out = b->dalloc<float>(static_cast<size_t>(S) * CAP * ldE);
    if (!b->feat || !emb_norm || !emb_out) { mot_bot_destroy(b); return MOT_ERR_NOMEM; }
    (void)hipMemsetAsync(b->feat, 0, sizeof(float) * static_cast<size_t>(S) * CAP * E, ctx->stream);
  }
  b->d_streams = b->dalloc<BotStream>(S);
  b->d_counts = b->dalloc<int>(S);
  b->d_err = b->dalloc<int>(1);
  b->d_maxt = b->dalloc<int>(256);
  b->d_alive = b->dalloc<int>(S);
  b->flights.n_maxt = 256; b->flights.with_alive = true; b->flights.extra_host = 7;
  b->d_warps = b->dalloc<float>(static_cast<size_t>(S) * 6);
  b->d_has_warp = b->dalloc<int>(S);
  b->d_stats = b->dalloc<unsigned long long>(8 * 64);
  b->d_out = b->dalloc<float>(static_cast<size_t>(S) * CAP * 8);
  b->d_out_counts = b->dalloc<int>(S);
  b->d_offsets = b->dalloc<int>(static_cast<size_t>(S) + 1);
  if (b->d_stats) (void)hipMemset(b->d_stats, 0, 8 * 64 * sizeof(unsigned long long));
  for (auto& e : b->ev) (void)hipEventCreate(&e);
  BotTasks& K = b->tasks;
  K.det = b->dalloc<mot_det_task>(S);
  K.featn = b->dalloc<mot_feat_task>(S); K.fset = b->dalloc<mot_feat_task>(S); K.fema = b->dalloc<mot_feat_task>(S);
  K.warp = b->dalloc<mot_kf_task>(S); K.pred = b->dalloc<mot_kf_task>(S); K.predw = b->dalloc<mot_kf_task>(S);
  K.ubox = b->dalloc<mot_kf_task>(S); K.init = b->dalloc<mot_kf_task>(S); K.upd = b->dalloc<mot_kf_task>(S); K.obox = b->dalloc<mot_kf_task>(S);
  K.cos1 = b->dalloc<mot_cos_task>(S); K.cos3 = b->dalloc<mot_cos_task>(S);
  K.lap1 = b->dalloc<mot_lap_task>(S); K.lap23 = b->dalloc<mot_lap_task>(2 * S);
  const size_t wb1 = (mot::lap_scratch_bytes(CAP, D) + 255) & ~size_t(255);
  char* work = b->dalloc<char>(wb1 * 3 * S);
  int* info = b->dalloc<int>(static_cast<size_t>(4) * 3 * S);
  if (!ip || !fp || !b->mean || !b->d_streams || !b->d_counts || !b->d_err || !b->d_maxt || !b->d_alive || !b->d_warps || !b->d_has_warp || !b->d_stats ||
      !b->d_out || !b->d_out_counts || !b->d_offsets || !K.det || !K.featn || !K.fset || !K.fema || !K.warp || !K.pred || !K.predw || !K.ubox ||
      !K.init || !K.upd || !K.obox || !K.cos1 || !K.cos3 || !K.lap1 || !K.lap23 || !work || !info) {
    mot_bot_destroy(b);
    return MOT_ERR_NOMEM;
  }
  std::vector<BotStream> hs(S);
  std::vector<mot_det_task> det(S);
  std::vector<mot_feat_task> featn(S), fset(S), fema(S);
  std::vector<mot_kf_task> warp(S), pred(S), predw(S), ubox(S), init(S), upd(S), obox(S);
  std::vector<mot_cos_task> cos1(S), cos3(S);
  std::vector<mot_lap_task> lap1(S), lap23(2 * S);
  for (int s = 0; s < S; ++s) {
    BotStream& T = hs[s];
    std::memset(&T, 0, sizeof(T));
    int* i = ip + ints_per * s;
    auto I = [&](int n) { int* r = i; i += n; return r; };
    T.free_stack = I(CAP); T.active[0] = I(CAP); T.active[1] = I(CAP); T.lost[0] = I(CAP); T.lost[1] = I(CAP);
    T.t_id = I(CAP); T.t_state = I(CAP); T.t_act = I(CAP); T.t_tlen = I(CAP); T.t_fid = I(CAP); T.t_sf = I(CAP); T.t_cls = I(CAP); T.t_det = I(CAP);
    T.t_feat = I(CAP);
    T.pool_slot = I(CAP); T.unconf_slot = I(CAP); T.x1 = I(CAP); T.x2 = I(CAP); T.x3 = I(CAP); T.upd_slot = I(CAP); T.upd_meas = I(CAP);
    T.set_slot = I(CAP); T.ema_slot = I(CAP); T.u_track = I(CAP); T.r_pool = I(CAP); T.lost_new = I(CAP);  // 26 CAP-sized arrays (set_det / ema_det below)
    T.set_det = nullptr; T.ema_det = nullptr;
    T.first = I(D); T.second = I(D); T.y1 = I(D); T.y2 = I(D); T.y3 = I(D); T.u_det = I(D); T.rem = I(D); T.init_dst = I(D); T.init_meas = I(D);
    float* f = fp + floats_per * s;
    auto F = [&](int n) { float* r = f; f += n; return r; };
    T.t_conf = F(CAP);
    float* pool_box = F(4 * CAP); float* ub = F(4 * CAP); T.abox = F(4 * CAP);
    float* d_box = F(4 * D); float* d_meas = F(4 * D);
    float* mean = b->mean + static_cast<size_t>(s) * 72 * CAP;
    T.kmean = mean;
    float* feat = E ? b->feat + static_cast<size_t>(s) * CAP * E : nullptr;
    float* en = E ? emb_norm + static_cast<size_t>(s) * D * E : nullptr;
    float* eo = E ? emb_out + static_cast<size_t>(s) * CAP * ldE : nullptr;
    std::memset(&det[s], 0, sizeof(mot_det_task));
    det[s].box = d_box; det[s].ldb = D; det[s].meas = d_meas; det[s].ldm = D;
    auto ft = [&](mot_feat_task& k, float* dst, int mode) { std::memset(&k, 0, sizeof(k)); k.d = E; k.feat = dst; k.ldf = E; k.lds = E; k.mode = mode; k.alpha = 0.9f; };
    ft(featn[s], en, 0);
    ft(fset[s], feat, 0); fset[s].slot = T.set_slot;
    ft(fema[s], feat, 1); fema[s].slot = T.ema_slot;
    auto kf = [&](mot_kf_task& k) { std::memset(&k, 0, sizeof(k)); k.mean = mean; k.cov = mean + 8; k.cap = CAP; };
    kf(warp[s]); warp[s].src = T.unconf_slot;
    kf(pred[s]); pred[s].src = T.pool_slot; pred[s].boxes = pool_box; pred[s].ldb = CAP;
    kf(predw[s]); predw[s].src = T.pool_slot; predw[s].boxes = pool_box; predw[s].ldb = CAP;
    kf(ubox[s]); ubox[s].src = T.unconf_slot; ubox[s].boxes = ub; ubox[s].ldb = CAP;
    kf(init[s]); init[s].src = T.init_dst; init[s].dst = T.init_dst; init[s].meas = d_meas; init[s].ldm = D; init[s].midx = T.init_meas;
    kf(upd[s]); upd[s].src = T.upd_slot; upd[s].dst = T.upd_slot; upd[s].meas = d_meas; upd[s].ldm = D; upd[s].midx = T.upd_meas;
    kf(obox[s]); obox[s].boxes = T.abox; obox[s].ldb = CAP;
    auto cs = [&](mot_cos_task& c, const int* aidx, const int* bidx) {
      std::memset(&c, 0, sizeof(c));
      c.d = E; c.a = feat; c.lda = E; c.aidx = aidx; c.b = en; c.ldb = E; c.bidx = bidx; c.out = eo; c.ldo = ldE;
    };
    cs(cos1[s], T.pool_slot, T.first);
    cs(cos3[s], T.unconf_slot, T.rem);
    auto lap = [&](mot_lap_task& L, int k, int* x, int* y, const float* a, const int* aidx, const int* bidx, int mode, float thresh, int fuse) {
      std::memset(&L, 0, sizeof(L));
      L.x = x; L.y = y; L.thresh = thresh; L.mode = MOT_LAP_PLAIN; L.info = info + (static_cast<size_t>(s) * 3 + k) * 4;
      L.work = work + (static_cast<size_t>(s) * 3 + k) * wb1;
      L.geom.a = a; L.geom.lda = CAP; L.geom.aidx = aidx; L.geom.b = d_box; L.geom.ldb = D; L.geom.bidx = bidx; L.geom.mode = mode;
      L.geom.prox_thresh = P.prox; L.geom.app_thresh = P.app; L.geom.fuse = fuse;
    };
    lap(lap1[s], 0, T.x1, T.y1, pool_box, nullptr, T.first, MOT_COST_BOTSORT, P.match, P.fuse_first);      // :433-466
    lap(lap23[2 * s], 1, T.x2, T.y2, pool_box, T.r_pool, T.second, MOT_COST_IOU_DIST, 0.5f, 0);          // :507-531
    lap(lap23[2 * s + 1], 2, T.x3, T.y3, ub, nullptr, T.rem, MOT_COST_BOTSORT, 0.7f, 1);                 // :591-623
  }
  // the detection-index halves of the feature lists: carved from one more allocation
  int* fdet = b->dalloc<int>(static_cast<size_t>(2) * CAP * S);
  if (!fdet) { mot_bot_destroy(b); return MOT_ERR_NOMEM; }
  for (int s = 0; s < S; ++s) {
    hs[s].set_det = fdet + static_cast<size_t>(2) * CAP * s; hs[s].ema_det = hs[s].set_det + CAP;
    fset[s].sidx = hs[s].set_det; fema[s].sidx = hs[s].ema_det;
  }
  b->h_streams = hs;
  hipStream_t st = ctx->stream;
#define BOT_UP(dst, vec) MOT_LC_HIP(b, hipMemcpyAsync(dst, vec.data(), sizeof(vec[0]) * vec.size(), hipMemcpyHostToDevice, st))
  BOT_UP(b->d_streams, hs); BOT_UP(K.det, det); BOT_UP(K.featn, featn); BOT_UP(K.fset, fset); BOT_UP(K.fema, fema);
  BOT_UP(K.warp, warp); BOT_UP(K.pred, pred); BOT_UP(K.predw, predw); BOT_UP(K.ubox, ubox); BOT_UP(K.init, init); BOT_UP(K.upd, upd);
  BOT_UP(K.obox, obox); BOT_UP(K.cos1, cos1); BOT_UP(K.cos3, cos3); BOT_UP(K.lap1, lap1); BOT_UP(K.lap23, lap23);
#undef BOT_UP
  MOT_LC_HIP(b, hipMemsetAsync(b->d_err, 0, sizeof(int), st));
  MOT_LC_HIP(b, hipMemsetAsync(b->d_has_warp, 0, sizeof(int) * S, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  *out = b;
  return MOT_OK;
}

}  // extern "C"

// queues one frame's launches: counts already on their way to b->d_counts, warps (if any) to b->d_warps / b->d_has_warp
static int bot_enqueue(mot_bot_batch* b, const float* d_dets, const int* h_counts, const float* d_embs, bool any_warp, int bound_n,
                       float* d_packed, int* d_offsets, int rows_cap, hipEvent_t* ev, const mot::lifecycle::FrameDev* fd = nullptr) {
  hipStream_t st = b->ctx->stream;
  const int S = b->S, CAP = b->CAP, D = b->D;
  if (!b->flights.maxt_clean) MOT_LC_HIP(b, hipMemsetAsync(b->d_maxt, 0, 256 * sizeof(int), st));  // (else: the last frame's pack_offsets cleared them)
  b->flights.maxt_clean = false;
  int bd = 1;
  for (int s = 0; s < S; ++s) bd = (h_counts[s] > bd) ? h_counts[s] : bd;
  int active = 0;  // streams with a frame: a launch with a handful of problems is tuned for latency (mot::launch_lap)
  for (int s = 0; s < S; ++s) active += (h_counts[s] >= 0) ? 1 : 0;
  if (bd > D) bd = D;
  const int bn = (bound_n < 1) ? 1 : (bound_n > CAP ? CAP : bound_n);
  const int bn2 = (bn + bd > CAP) ? CAP : bn + bd;
  const bool emb = b->prm.with_reid && d_embs != nullptr;
  const bool prof = ev != nullptr;
  const BotTasks& K = b->tasks;
  mot::lifecycle::FrameDev FD;
  if (fd) FD = *fd; else FD.counts = b->d_counts;
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[0], st));
  hipLaunchKernelGGL(bot_begin, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, D, FD, d_dets, emb ? d_embs : nullptr, b->d_warps,
                     any_warp ? b->d_has_warp : nullptr, K, prof ? b->d_stats : nullptr, b->d_maxt);
  // (round 5: the detections' boxes and measurements come out of bot_begin, the output rows' boxes out of bot_finish, which also gathers the error word)
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[7], st));
  if (emb) MOT_LC_HIP(b, mot::launch_feat(K.featn, S, bd, st));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[8], st));
  if (any_warp) {
    MOT_LC_HIP(b, mot::launch_kf_op(4, MOT_KF_XYWH, K.warp, S, bn, st));       // multi_gmc(unconfirmed) :323
    MOT_LC_HIP(b, mot::launch_kf_op(5, MOT_KF_XYWH, K.predw, S, bn, st));      // multi_predict + multi_gmc(pool) :316-322
  }
  MOT_LC_HIP(b, mot::launch_kf_op(1, MOT_KF_XYWH, K.pred, S, bn, st));         // multi_predict :316
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[2], st));
  // (round 5) the appearance distances of the pairs that pass the proximity test only — the others are forced to 1 by the reference and never
  // read by the solvers (cosine_gated.hip); MOT_BOT_DENSE_COSINE=1 writes the whole matrix on the MFMA as before (A/B measurements)
  static const bool dense_cos = std::getenv("MOT_BOT_DENSE_COSINE") && std::getenv("MOT_BOT_DENSE_COSINE")[0] == '1';
  auto appearance = [&](const mot_cos_task* cos, const mot_lap_task* lap, int stride) {
    if (!dense_cos && mot::launch_embed_gated(cos, lap, stride, S, bn, bd, st) == hipSuccess) return hipSuccess;
    return mot::launch_embed(MOT_EMB_COSINE, cos, S, bn, bd, st);
  };
  if (emb) MOT_LC_HIP(b, appearance(K.cos1, K.lap1, 1));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[3], st));
  MOT_LC_HIP(b, mot::launch_lap(K.lap1, S, bn, bd, true, false, false, st, b->hint1_n, 0, true, nullptr, nullptr, nullptr, active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[4], st));
  hipLaunchKernelGGL(bot_after_first, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, K, prof ? b->d_stats : nullptr, b->d_maxt);
  MOT_LC_HIP(b, mot::launch_kf_op(3, MOT_KF_XYWH, K.ubox, S, bn, st));
  if (emb) MOT_LC_HIP(b, appearance(K.cos3, K.lap23 + 1, 2));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[5], st));
  MOT_LC_HIP(b, mot::launch_lap(K.lap23, 2 * S, bn, bd, true, false, false, st, b->hint23_n, b->hint23_m, true, nullptr, nullptr, nullptr, 2 * active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[6], st));
  hipLaunchKernelGGL(bot_after_second, dim3(S), dim3(kW), 0, st, b->d_streams, b->prm, CAP, K, prof ? b->d_stats : nullptr);
  MOT_LC_HIP(b, mot::launch_kf_op(0, MOT_KF_XYWH, K.init, S, bd, st));
  MOT_LC_HIP(b, mot::launch_kf_op(2, MOT_KF_XYWH, K.upd, S, bn, st));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[9], st));
  if (emb) {
    MOT_LC_HIP(b, mot::launch_feat(K.fset, S, bn2, st));
    MOT_LC_HIP(b, mot::launch_feat(K.fema, S, bn, st));
  }
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[10], st));
  hipLaunchKernelGGL(bot_finish, dim3(S), dim3(kW), 0, st, b->d_streams, CAP, b->d_out, b->d_out_counts, CAP, b->d_maxt, b->d_alive, b->d_err);
  hipLaunchKernelGGL(mot::lifecycle::pack_offsets, dim3(1), dim3(1024), 0, st, b->d_out_counts, S, d_offsets, b->pack_meta);
  hipLaunchKernelGGL(mot::lifecycle::pack_rows, dim3(S), dim3(256), 0, st, b->d_out, CAP, b->d_out_counts, d_offsets, d_packed, rows_cap);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[1], st));
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}
static void bot_set_hints(mot_bot_batch* b, const int* maxt) {  // next frame's LDS hints: this frame's largest problems, a quarter more
  int m1 = 0, m2 = 0, m3 = 0;
  for (int i = 0; i < 64; ++i) { m1 = (maxt[64 + i] > m1) ? maxt[64 + i] : m1; m2 = (maxt[128 + i] > m2) ? maxt[128 + i] : m2; m3 = (maxt[192 + i] > m3) ? maxt[192 + i] : m3; }
  b->hint1_n = m1 > 0 ? m1 + m1 / 8 + 48 : 0;
  b->hint23_n = m2 > 0 ? m2 + m2 / 4 + 32 : 0;
  b->hint23_m = m3 > 0 ? m3 + m3 / 4 + 32 : 0;
}
static int bot_account_events(mot_bot_batch* b, hipEvent_t* ev) {
  float ms = 0.f;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[0], ev[1])); b->frame_ms += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[2], ev[3])); b->cos_ms += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[3], ev[4])); b->lap_ms += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[5], ev[6])); b->lap_ms += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[7], ev[8])); b->feat_ms += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[9], ev[10])); b->feat_ms += ms;
  b->frames += 1;
  return MOT_OK;
}

extern "C" {

int mot_bot_step_packed(mot_bot_batch* b, const float* d_dets, const int* h_counts, const float* d_embs, const float* h_warps6,
                        const unsigned char* h_has_warp, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b || !d_dets || !h_counts || !rows || !out_counts) return MOT_ERR_INVALID;
  if (b->flights.count > 0) { b->ctx->err = "mot_bot_step_packed: frames are in flight (collect them first)"; return MOT_ERR_INVALID; }
  hipStream_t st = b->ctx->stream;
  const int S = b->S;
  MOT_LC_HIP(b, hipMemcpyAsync(b->d_counts, h_counts, sizeof(int) * S, hipMemcpyHostToDevice, st));
  bool any_warp = false;
  std::vector<int> hw;
  if (h_warps6 && h_has_warp) {
    hw.resize(S);
    for (int s = 0; s < S; ++s) { hw[s] = h_has_warp[s] ? 1 : 0; any_warp = any_warp || hw[s]; }
  }
  if (any_warp) {
    MOT_LC_HIP(b, hipMemcpyAsync(b->d_warps, h_warps6, sizeof(float) * 6 * S, hipMemcpyHostToDevice, st));
    MOT_LC_HIP(b, hipMemcpyAsync(b->d_has_warp, hw.data(), sizeof(int) * S, hipMemcpyHostToDevice, st));
    MOT_LC_HIP(b, hipStreamSynchronize(st));  // hw is a local
  }
  if (rows_cap > b->packed_cap) { b->d_packed = b->dalloc<float>(static_cast<size_t>(rows_cap) * 8); b->packed_cap = b->d_packed ? rows_cap : 0; }
  if (!b->d_packed) return MOT_ERR_NOMEM;
  const int rc = bot_enqueue(b, d_dets, h_counts, d_embs, any_warp, b->bound_n, b->d_packed, b->d_offsets, rows_cap, b->profile ? b->ev : nullptr);
  if (rc != MOT_OK) return rc;
  b->d_rows_last = b->d_packed; b->d_offsets_last = b->d_offsets; b->d_counts_last = b->d_out_counts;
  int total = 0, err = 0;
  int maxt[256];
  MOT_LC_HIP(b, hipMemcpyAsync(out_counts, b->d_out_counts, sizeof(int) * S, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(&total, b->d_offsets + S, sizeof(int), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(&err, b->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(maxt, b->d_maxt, sizeof(maxt), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  b->bound_n = 0;
  for (int i = 0; i < 64; ++i) b->bound_n = (maxt[i] > b->bound_n) ? maxt[i] : b->bound_n;
  bot_set_hints(b, maxt);
  if (b->profile) { const int rce = bot_account_events(b, b->ev); if (rce != MOT_OK) return rce; }
  if (total_rows) *total_rows = total;
  if (err) { b->ctx->err = "mot_bot_step_packed: a stream exceeded cap_tracks / max_dets"; return MOT_ERR_CAPACITY; }
  if (total > rows_cap) { b->ctx->err = "mot_bot_step_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  if (total > 0) {
    MOT_LC_HIP(b, hipMemcpyAsync(rows, b->d_packed, sizeof(float) * static_cast<size_t>(total) * 8, hipMemcpyDeviceToHost, st));
    MOT_LC_HIP(b, hipStreamSynchronize(st));
  }
  return MOT_OK;
}

// Frames in flight (see mot_bt_enqueue_packed): enqueue returns once the launches are queued, collect waits for the oldest
// pending frame and copies its rows on a second stream while the next frame runs. `in` != nullptr: the pooled form.
static int bot_enqueue_flight(mot_bot_batch* b, const float* d_dets, const int* h_counts, const float* d_embs, const float* h_warps6,
                              const unsigned char* h_has_warp, int rows_cap, const mot_frame_in* in) {
  if (b->flights.count >= 2) { b->ctx->err = "mot_bot_enqueue: two frames are already in flight (collect one first)"; return MOT_ERR_INVALID; }
  hipStream_t st = b->ctx->stream;
  const int S = b->S;
  const int slot = b->flights.slot_for_enqueue();
  int* counts_in = nullptr;
  int bd = 0;
  MOT_LC_HIP(b, b->flights.prepare(b->mem, slot, S, rows_cap, h_counts, &counts_in, &bd, in != nullptr, b->profile));
  mot::lifecycle::Flight& F = b->flights.fl[slot];
  int* hw = b->flights.extra_of(F, S);                  // page-locked: has_warp [S], then the warps [6 S]
  float* wp = reinterpret_cast<float*>(hw + S);
  bool any_warp = false;
  for (int s = 0; s < S; ++s) {
    hw[s] = (h_warps6 && h_has_warp && h_has_warp[s]) ? 1 : 0;
    any_warp = any_warp || hw[s];
  }
  mot::lifecycle::FrameDev fd;
  if (in) MOT_LC_HIP(b, b->flights.upload_block(b->mem, slot, S, in->h_counts, in->h_det_ld, in->h_det_off, in->h_emb_off, st, &fd));
  else MOT_LC_HIP(b, hipMemcpyAsync(b->d_counts, counts_in, sizeof(int) * S, hipMemcpyHostToDevice, st));
  if (any_warp) {
    std::memcpy(wp, h_warps6, sizeof(float) * 6 * S);
    MOT_LC_HIP(b, hipMemcpyAsync(b->d_warps, wp, sizeof(float) * 6 * S, hipMemcpyHostToDevice, st));
    MOT_LC_HIP(b, hipMemcpyAsync(b->d_has_warp, hw, sizeof(int) * S, hipMemcpyHostToDevice, st));
  }
  int bound = b->bound_n + b->flights.pending_bd();
  if (bound > b->CAP) bound = b->CAP;
  mot::lifecycle::PackMeta pm = b->flights.pack_meta(slot, b->d_err, b->d_maxt, nullptr);
  pm.alive = b->d_alive; pm.alive_at = b->flights.meta_head() + S;
  b->pack_meta = pm;
  const int rc = bot_enqueue(b, d_dets, counts_in, d_embs, any_warp, bound, b->flights.rows_target(slot), F.d_offsets, rows_cap, F.prof ? F.ev : nullptr,
                             in ? &fd : nullptr);
  b->pack_meta = mot::lifecycle::PackMeta{};
  if (rc != MOT_OK) return rc;
  MOT_LC_HIP(b, b->flights.finish_copies(slot, st, S, rows_cap, bd));
  return MOT_OK;
}
static int bot_pop_flight(mot_bot_batch* b, mot::lifecycle::Flight** out, int* total) {
  if (b->flights.count <= 0) { b->ctx->err = "mot_bot_collect: no frame in flight"; return MOT_ERR_INVALID; }
  mot::lifecycle::Flight* F = nullptr;
  MOT_LC_HIP(b, b->flights.pop(&F));
  if (F->prof) { const int rce = bot_account_events(b, F->ev); if (rce != MOT_OK) return rce; }
  const int* maxt = b->flights.maxt_of(*F);
  b->bound_n = 0;
  for (int i = 0; i < 64; ++i) b->bound_n = (maxt[i] > b->bound_n) ? maxt[i] : b->bound_n;
  bot_set_hints(b, maxt);
  *total = F->h_meta[0];
  *out = F;
  b->d_rows_last = F->view ? F->h_rows : F->d_packed; b->d_offsets_last = F->d_offsets; b->d_counts_last = F->d_counts;
  if (F->h_meta[1]) { b->ctx->err = "mot_bot_collect: a stream exceeded cap_tracks / max_dets"; return MOT_ERR_CAPACITY; }
  if (*total > F->rows_cap) { b->ctx->err = "mot_bot_collect: more rows than rows_cap"; return MOT_ERR_CAPACITY; }  // (pack_rows skipped the streams that end past the ENQUEUE call's rows_cap)
  return MOT_OK;
}

int mot_bot_enqueue_packed(mot_bot_batch* b, const float* d_dets, const int* h_counts, const float* d_embs, const float* h_warps6,
                           const unsigned char* h_has_warp, int rows_cap) {
  if (!b || !d_dets || !h_counts || rows_cap <= 0) return MOT_ERR_INVALID;
  return bot_enqueue_flight(b, d_dets, h_counts, d_embs, h_warps6, h_has_warp, rows_cap, nullptr);
}

int mot_bot_collect_packed(mot_bot_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b || !out_counts) return MOT_ERR_INVALID;  // rows == NULL: the table stays on the device (mot_bot_device_output), only the counts come back
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = bot_pop_flight(b, &F, &total);
  if (F) std::memcpy(out_counts, b->flights.counts_of(*F), sizeof(int) * b->S);
  if (total_rows) *total_rows = total;
  if (rc != MOT_OK) return rc;
  if (!rows) return MOT_OK;
  if (total > rows_cap) { b->ctx->err = "mot_bot_collect_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  MOT_LC_HIP(b, b->flights.copy_rows(*F, rows, total));
  return MOT_OK;
}

// ---- pooled form (see mot_bt_enqueue_frame) ----
int mot_bot_enqueue_frame(mot_bot_batch* b, const mot_frame_in* in, int rows_cap) {
  if (!b || !in || !in->d_dets || !in->h_counts || !in->h_det_ld || !in->h_det_off || rows_cap <= 0) return MOT_ERR_INVALID;
  if (in->d_embs && !in->h_emb_off) return MOT_ERR_INVALID;
  return bot_enqueue_flight(b, in->d_dets, in->h_counts, in->d_embs, in->h_warps6, in->h_has_warp, rows_cap, in);
}
int mot_bot_collect_view(mot_bot_batch* b, mot_frame_view* out) {
  if (!b || !out) return MOT_ERR_INVALID;
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = bot_pop_flight(b, &F, &total);
  if (!F) return rc;
  if (!F->view) { b->ctx->err = "mot_bot_collect_view: the frame was queued with mot_bot_enqueue_packed"; return MOT_ERR_INVALID; }
  out->rows = F->h_rows; out->counts = b->flights.counts_of(*F); out->alive = b->flights.alive_of(*F, b->S); out->total = total;
  return rc;
}
int mot_bot_reset_stream(mot_bot_batch* b, int s, int fresh) {
  if (!b || s < 0 || s >= b->S) return MOT_ERR_INVALID;
  hipLaunchKernelGGL(mot::lifecycle::reset_stream_kernel<BotStream>, dim3(1), dim3(64), 0, b->ctx->stream, b->d_streams, s, b->h_streams[s], 0 * fresh /* the ids restart either way, botsort.cpp:257 */, b->d_err);
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}
namespace {
__global__ void __launch_bounds__(256) bot_move(const BotStream* from, BotStream* to, const float* mean_from, float* mean_to, const float* feat_from,
                                                float* feat_to, int cap, int E) {
  using mot::lifecycle::move_array;
  const BotStream& A = *from;
  BotStream& B = *to;
  const size_t n = static_cast<size_t>(cap);
  move_array(B.free_stack, A.free_stack, n);
  move_array(B.active[0], A.active[A.cur], n); move_array(B.lost[0], A.lost[A.cur], n);
  move_array(B.t_id, A.t_id, n); move_array(B.t_state, A.t_state, n); move_array(B.t_act, A.t_act, n); move_array(B.t_tlen, A.t_tlen, n);
  move_array(B.t_fid, A.t_fid, n); move_array(B.t_sf, A.t_sf, n); move_array(B.t_cls, A.t_cls, n); move_array(B.t_det, A.t_det, n);
  move_array(B.t_feat, A.t_feat, n); move_array(B.t_conf, A.t_conf, n);
  move_array(mean_to, mean_from, n * 72);
  if (E > 0) move_array(feat_to, feat_from, n * E);
  __syncthreads();
  if (threadIdx.x == 0) {
    B.frame_count = A.frame_count; B.next_id = A.next_id; B.next_slot = A.next_slot; B.n_free = A.n_free;
    B.n_active = A.n_active; B.n_lost = A.n_lost; B.err = A.err; B.cur = 0; B.idle = 1;
  }
}
}  // namespace
int mot_bot_move_stream(mot_bot_batch* src, int s, mot_bot_batch* dst, int s2) {
  if (!src || !dst || s < 0 || s >= src->S || s2 < 0 || s2 >= dst->S || dst->CAP < src->CAP || dst->D < src->D || dst->E != src->E) return MOT_ERR_INVALID;
  MOT_LC_HIP(src, hipStreamSynchronize(src->ctx->stream));
  hipStream_t st = dst->ctx->stream;
  const int E = src->E;
  hipLaunchKernelGGL(bot_move, dim3(1), dim3(256), 0, st, src->d_streams + s, dst->d_streams + s2, src->mean + static_cast<size_t>(s) * 72 * src->CAP,
                     dst->mean + static_cast<size_t>(s2) * 72 * dst->CAP, E ? src->feat + static_cast<size_t>(s) * src->CAP * E : nullptr,
                     E ? dst->feat + static_cast<size_t>(s2) * dst->CAP * E : nullptr, src->CAP, E);
  MOT_LC_HIP(dst, hipGetLastError());
  BotStream h;
  MOT_LC_HIP(dst, hipMemcpyAsync(&h, dst->d_streams + s2, sizeof(BotStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(dst, hipStreamSynchronize(st));
  if (h.n_active + h.n_lost > dst->bound_n) dst->bound_n = h.n_active + h.n_lost;
  return MOT_OK;
}

int mot_bot_device_output(mot_bot_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts) {
  if (!b || !b->d_rows_last || !b->d_offsets_last) return MOT_ERR_INVALID;
  if (d_rows) *d_rows = b->d_rows_last;
  if (d_offsets) *d_offsets = b->d_offsets_last;
  if (d_counts) *d_counts = b->d_counts_last;
  return MOT_OK;
}

int mot_bot_profile(mot_bot_batch* b, int enable) {
  b->profile = enable != 0;
  if (enable) {
    b->lap_ms = b->cos_ms = b->frame_ms = b->feat_ms = 0.0;
    b->frames = 0;
    MOT_LC_HIP(b, hipMemsetAsync(b->d_stats, 0, 8 * 64 * sizeof(unsigned long long), b->ctx->stream));
    MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));
  }
  return MOT_OK;
}

int mot_bot_profile_stats(mot_bot_batch* b, double* out8) {
  unsigned long long raw[8 * 64];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long h[4] = {0, 0, 0, 0};
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 4; ++k) h[k] += raw[i * 8 + k];
  out8[0] = b->lap_ms; out8[1] = b->cos_ms; out8[2] = b->frame_ms; out8[3] = static_cast<double>(b->frames);
  out8[4] = static_cast<double>(h[0]); out8[5] = static_cast<double>(h[1]); out8[6] = static_cast<double>(h[2]); out8[7] = static_cast<double>(b->E);
  return MOT_OK;
}

int mot_bot_profile_feat(mot_bot_batch* b, double* out2) {  // HIP-event ms of the three feat_kernel launches of the profiled frames; feature rows they moved
  unsigned long long raw[8 * 64];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long rows = 0;
  for (int i = 0; i < 64; ++i) rows += raw[i * 8 + 3];
  out2[0] = b->feat_ms; out2[1] = static_cast<double>(rows);
  return MOT_OK;
}

int mot_bot_dump(mot_bot_batch* b, int s, int* ids, float* mean, float* cov, float* feats, unsigned char* has_feat, int cap) {
  hipStream_t st = b->ctx->stream;
  BotStream h;
  MOT_LC_HIP(b, hipMemcpyAsync(&h, b->d_streams + s, sizeof(BotStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  const int n = h.n_active + h.n_lost;
  if (n > cap) return -n;
  const int CAP = b->CAP, E = b->E;
  std::vector<int> slots(n), tid(CAP), tf(CAP);
  if (h.n_active) MOT_LC_HIP(b, hipMemcpyAsync(slots.data(), h.active[h.cur], sizeof(int) * h.n_active, hipMemcpyDeviceToHost, st));
  if (h.n_lost) MOT_LC_HIP(b, hipMemcpyAsync(slots.data() + h.n_active, h.lost[h.cur], sizeof(int) * h.n_lost, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(tid.data(), h.t_id, sizeof(int) * CAP, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(tf.data(), h.t_feat, sizeof(int) * CAP, hipMemcpyDeviceToHost, st));
  std::vector<float> m(static_cast<size_t>(72) * CAP), fe(feats && E ? static_cast<size_t>(CAP) * E : 0);
  MOT_LC_HIP(b, hipMemcpyAsync(m.data(), b->mean + static_cast<size_t>(s) * 72 * CAP, sizeof(float) * m.size(), hipMemcpyDeviceToHost, st));
  if (!fe.empty()) MOT_LC_HIP(b, hipMemcpyAsync(fe.data(), b->feat + static_cast<size_t>(s) * CAP * E, sizeof(float) * fe.size(), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {
    const int sl = slots[i];
    ids[i] = tid[sl];
    for (int k = 0; k < 8; ++k) mean[static_cast<size_t>(i) * 8 + k] = m[static_cast<size_t>(sl) * 72 + k];
    for (int k = 0; k < 64; ++k) cov[static_cast<size_t>(i) * 64 + k] = m[static_cast<size_t>(sl) * 72 + 8 + k];
    if (has_feat) has_feat[i] = tf[sl] ? 1 : 0;
    if (!fe.empty()) std::memcpy(feats + static_cast<size_t>(i) * E, fe.data() + static_cast<size_t>(sl) * E, sizeof(float) * E);
  }
  return n;
}

}  // extern "C"
