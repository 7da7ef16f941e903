// ByteTrack with the per-stream lifecycle ON THE DEVICE (reference: src/trackers/bytetrack.cpp:166-706).
//
// The host-side stage machine (host/bytetrack.cpp) spends ~14 us of one CPU core per stream-frame on list bookkeeping
// and needs three host<->device round trips per frame; on a box whose CPU quota is 16 cores that, not the GPU, bounds
// the 256x128 configuration. Here the bookkeeping itself runs in four small kernels (one wavefront per stream) between
// the numeric kernels, which read their task descriptors from device memory anyway: a frame is a FIXED sequence of 14
// launches for all streams with no host decision in between, and one copy of the output tables at the end.
//
// Track records are indexed by their Kalman slot (a track keeps its slot for life), the active/lost lists are arrays of
// slots in the reference's list order, and every "for ... push_back" of the reference becomes an order-preserving
// wavefront compaction (ballot + popcount prefix). Sizes are fixed at creation (cap_tracks, max_dets); exceeding them
// raises the stream's error flag instead of reallocating.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lifecycle_common.hpp"
#include "cost_math.hpp"

namespace {
using mot::lifecycle::compact;
using mot::lifecycle::kW;
using mot::lifecycle::FrameDev;

enum St { New = 0, Tracked = 1, Lost = 2, Removed = 3 };

struct BtParams {
  float min_conf, track_thresh, match_thresh, det_thresh;
  int max_time_lost;
};

// Everything one stream owns, device resident. Arrays are CAP (tracks) or D (detections) long.
struct BtStream {
  // ---- persistent ----
  int frame_count, next_id, next_slot, n_free, n_active, n_lost, err;
  int* free_stack;
  int* active[2]; int* lost[2]; int cur;  // lists of slots, ping-pong buffers
  int *t_id, *t_state, *t_act, *t_tlen, *t_fid, *t_sf, *t_cls, *t_det;
  float* t_conf;
  // ---- frame input ----
  const float* dets; int ld, n;  // SoA [6][ld]
  int skip;                      // the stream sits this frame out (pooled form: counts[s] < 0)
  // ---- frame scratch ----
  int *high, *second; int n_high, n_second;
  int* pool_slot; int n_pool, n_tracked;   // pool = tracked (from active) ++ lost
  int* unconf_slot; int n_unconf;
  int *pred_src, *pred_dst; unsigned char* pred_flags;
  int *x1, *y1, *x2, *y2, *x3, *y3;
  int *upd_src, *upd_dst, *upd_meas; unsigned char* upd_flags; int n_upd;
  int* refind; int n_refind;
  int* u_track; int n_utrack;
  int* u_det; int n_udet;
  int *r_slot, *r_pool; int n_r;
  int* rem;
  int lap2_q, lap3_q;
  int *init_dst, *init_meas; int n_init;
  int* lost_new; int n_lost_new;
  int *age_a, *age_b; unsigned char *dup_a, *dup_b;
  float* abox;  // [4][CAP] (round 5: unused — the boxes of the new lists are computed from the means where they are read)
  float* lbox;
  // round 5: the box passes of a frame (kf_kernel<XYAH, boxes> twice, <XYAH, predict + boxes> once) are folded into the list kernels: a box is
  // eight loads and six operations of a lane that has the slot in hand anyway, a launch is 5 us of a single camera's frame
  const float* kmean;  // this stream's dense-form covariances: 64 floats per slot, 256-byte records — only for the tracks that left the block form (kflag)
  float* kblk;         // round 6: the covariances in block form, [slot][4][4] = {P(c,c), P(c,c+4), P(c+4,c), P(c+4,c+4)} (kf_kernels.hip: kf_update_blocks_kernel)
  unsigned char* kflag;  // [slot] 1: the track's covariance lives in its 64-float record (it met a non-finite value or a non-positive innovation variance)
  float* kdense;       // round 6: the MEANS, dense ([slot][8]): what the list kernels read boxes from and what the Kalman update reads and writes
                       // (mot_kf_task.mean_dense) — a record's mean was 32 B of a 288-byte stride, a 64-byte line fetched for every track three or four
                       // times per frame (bt_begin and bt_dups moved 100 KB per stream that way); the record keeps the covariance
  float* dmeas4;  // round 6: the frame's Kalman measurements once more as [D][4] (mot_kf_task.meas4): the block-form update reads a detection's four
                  // components with one 16-byte access per quad of lanes instead of one line per component
  float *pool_box, *rbox, *ubox;  // [4][CAP] predicted boxes of the pool / stored boxes of the second association's tracks / of the unconfirmed ones
};

// KalmanFilterXYAH state -> box (kf_kernels.hip::s8_box<MOT_KF_XYAH>, ops.hpp:110-114): the same operations in the same order
__device__ __forceinline__ float4 xyah_box4(float cx, float cy, float a, float h) {
  const float w = a * h;
  return make_float4(cx - w * 0.5f, cy - h * 0.5f, cx + w * 0.5f, cy + h * 0.5f);
}
__device__ __forceinline__ float4 stored_box(const float* kdense, int slot) {
  const float4 m = *reinterpret_cast<const float4*>(kdense + static_cast<size_t>(slot) * 8);
  return xyah_box4(m.x, m.y, m.z, m.w);
}
// the box of the PREDICTED state, nothing stored (kf_kernel<XYAH, OP_PREDICT_BOXES>): x' = F x touches the mean only
__device__ __forceinline__ float4 predicted_box(const float* kdense, int slot, bool zero_v7) {
  const float4* mp = reinterpret_cast<const float4*>(kdense + static_cast<size_t>(slot) * 8);
  const float4 a = mp[0];
  float4 v = mp[1];
  if (zero_v7) v.w = 0.0f;
  return xyah_box4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
}
__device__ __forceinline__ void store_box(float* planes, int CAP, int i, const float4& b) {
  planes[i] = b.x; planes[static_cast<size_t>(CAP) + i] = b.y; planes[static_cast<size_t>(2) * CAP + i] = b.z; planes[static_cast<size_t>(3) * CAP + i] = b.w;
}



constexpr int kAF = 256;
constexpr int kAFMax = 1024;  // a handful of streams (one camera): the list kernels run with up to 16 wavefronts per stream - the chip is empty
                              // anyway and a frame is a chain of dependent launches, so each one's latency is the frame's
struct Compact3 {  // three order-preserving appends of one round, one exchange
  int pos[3];
};
__device__ __forceinline__ Compact3 compact3_block(bool p0, bool p1, bool p2, int& b0, int& b1, int& b2, int (*cnt)[3]) {
  const int lane = static_cast<int>(threadIdx.x) & 63, w = static_cast<int>(threadIdx.x) >> 6;
  const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1), m2 = __ballot(p2);
  if (lane == 0) { cnt[w][0] = __popcll(m0); cnt[w][1] = __popcll(m1); cnt[w][2] = __popcll(m2); }
  __syncthreads();
  int before[3] = {0, 0, 0}, tot[3] = {0, 0, 0};
  for (int k = 0; k < static_cast<int>(blockDim.x >> 6); ++k)
    for (int q = 0; q < 3; ++q) { const int c = cnt[k][q]; if (k < w) before[q] += c; tot[q] += c; }
  __syncthreads();  // (the counts are rewritten by the next round)
  const unsigned long long below = (1ull << lane) - 1ull;
  Compact3 r;
  r.pos[0] = b0 + before[0] + __popcll(m0 & below);
  r.pos[1] = b1 + before[1] + __popcll(m1 & below);
  r.pos[2] = b2 + before[2] + __popcll(m2 & below);
  b0 += tot[0]; b1 += tot[1]; b2 += tot[2];
  return r;
}
// ---- K0: detection split, pools, predict + first-association tasks (bytetrack.cpp:166-265) ----
// stats[0] = assignment problems queued, stats[1] = sum of their n + m (algorithmic bytes of the solver: 24 B per row/column)
__global__ void __launch_bounds__(kAFMax) bt_begin(BtStream* streams, BtParams P, int CAP, int D, FrameDev FD, const float* dets_base,
                                                 mot_det_task* det_t, mot_kf_task* pred_t, mot_lap_task* lap1_t, unsigned long long* stats, int* maxt,
                                                 int* declined) {
  __shared__ int cnt[kAFMax / 64][3];
  BtStream& S = streams[blockIdx.x];
  const int t = static_cast<int>(threadIdx.x);
  if (blockIdx.x == 0 && t == 0) *declined = 0;  // the counter of the assignment launch behind this kernel (mot::launch_lap, prezeroed)
  const int n = FD.counts[blockIdx.x];
  if (n < 0) {  // not this stream's frame: nothing of its state moves, every task it owns is empty
    if (t == 0) {
      S.skip = 1;
      det_t[blockIdx.x].n = 0; pred_t[blockIdx.x].n = 0;
      mot_lap_task& L = lap1_t[blockIdx.x];
      L.n = 0; L.m = 0; L.geom.n = 0; L.geom.m = 0;
    }
    return;
  }
  int ldd = D;
  const float* dets = mot::lifecycle::frame_dets(FD, dets_base, blockIdx.x, D, ldd);
  const int n_active = S.n_active, n_lost = S.n_lost;
  const int* act = S.active[S.cur];
  const int* lst = S.lost[S.cur];
  __syncthreads();  // (everyone has read the scalars the first lane updates below)
  if (t == 0) {
    S.frame_count += 1;
    S.dets = dets; S.ld = ldd; S.n = n; S.skip = 0;
    if (n > D) S.err = 1;
  }
  const float* conf = dets + static_cast<size_t>(4) * ldd;
  float* dbox = det_t[blockIdx.x].box;
  float* dmeas = det_t[blockIdx.x].meas;
  float* __restrict__ dmeas4 = S.dmeas4;
  const int dldb = det_t[blockIdx.x].ldb, dldm = det_t[blockIdx.x].ldm;
  int nh = 0, ns = 0, z = 0;
  // (round 6: kU chunks per thread, the loads of a level issued together — see bt_after_first)
  constexpr int kU = 4;
  const int T = static_cast<int>(blockDim.x);
  int* __restrict__ high = S.high; int* __restrict__ second = S.second; int* __restrict__ pool_slot = S.pool_slot; int* __restrict__ unconf_slot = S.unconf_slot;
  const int* __restrict__ t_act = S.t_act; const int* __restrict__ t_state = S.t_state; const float* __restrict__ kdense = S.kdense;
  int* __restrict__ pred_src = S.pred_src; int* __restrict__ pred_dst = S.pred_dst; unsigned char* __restrict__ pred_flags = S.pred_flags;
  float* __restrict__ pool_box = S.pool_box; float* __restrict__ ubox = S.ubox;
  const bool fits = n <= D;
  for (int base = 0; base < n; base += kU * T) {
    float c[kU], bx[kU][4];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = base + u * T + t;
      c[u] = (i < n) ? conf[i] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) bx[u][q] = (i < n && fits) ? dets[static_cast<size_t>(q) * ldd + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (base + u * T >= n) break;  // (uniform)
      const int i = base + u * T + t;
      const bool hi = i < n && c[u] > P.track_thresh;
      const bool lo = i < n && c[u] > P.min_conf && c[u] < P.track_thresh;
      const Compact3 k = compact3_block(hi, lo, false, nh, ns, z, cnt);
      if (hi) high[k.pos[0]] = i;
      if (lo) second[k.pos[1]] = i;
      if (i < n && fits) {  // the detection's association box and Kalman measurement (det_kernel<MOT_DET_XYAH>, bytetrack.cpp:29-33: the same operations)
        const float x1 = bx[u][0], y1 = bx[u][1], x2 = bx[u][2], y2 = bx[u][3];
        const float w = x2 - x1, h = y2 - y1;
        const float xc = x1 + w * 0.5f, yc = y1 + h * 0.5f;
        const float tl = xc - w * 0.5f, tt = yc - h * 0.5f;
        const float zz[4] = {tl + w * 0.5f, tt + h * 0.5f, (h > 0.0f) ? (w / h) : 0.0f, h};
        const float bb[4] = {xc - w * 0.5f, yc - h * 0.5f, xc + w * 0.5f, yc + h * 0.5f};
#pragma unroll
        for (int q = 0; q < 4; ++q) { dbox[static_cast<size_t>(q) * dldb + i] = bb[q]; dmeas[static_cast<size_t>(q) * dldm + i] = zz[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) dmeas4[static_cast<size_t>(i) * 4 + q] = zz[q];  // (scalar stores: the pool's offset is 16-byte aligned only when cap_tracks is a multiple of 4)
      }
    }
  }
  int np = 0, nu = 0;
  for (int base = 0; base < n_active; base += kU * T) {
    int slot[kU], ta[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < n_active) ? act[i] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; ta[u] = (i < n_active) ? t_act[slot[u]] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (base + u * T >= n_active) break;
      const int i = base + u * T + t;
      const bool a = i < n_active && ta[u] != 0;
      const bool un = i < n_active && ta[u] == 0;
      const Compact3 k = compact3_block(a, un, false, np, nu, z, cnt);
      if (a) pool_slot[k.pos[0]] = slot[u];
      if (un) unconf_slot[k.pos[1]] = slot[u];
    }
  }
  const int n_tracked = np;
  for (int base = 0; base < n_lost; base += kU * T) {  // tracked and lost are disjoint by id at frame start (:565-580 of the previous frame)
    int sl[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; sl[u] = (i < n_lost) ? lst[i] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; if (i < n_lost) pool_slot[np + i] = sl[u]; }
  }
  np += n_lost;
  __syncthreads();
  for (int base = 0; base < np; base += kU * T) {  // the reference predicts COPIES of the pool (:251-265): here the prediction is box-only
    int slot[kU], st[kU];
    float4 ma[kU], mv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < np) ? pool_slot[i] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = base + u * T + t;
      const float4* mp = reinterpret_cast<const float4*>(kdense + static_cast<size_t>(slot[u]) * 8);
      st[u] = (i < np) ? t_state[slot[u]] : 0;
      ma[u] = (i < np) ? mp[0] : make_float4(0.f, 0.f, 0.f, 0.f);
      mv[u] = (i < np) ? mp[1] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = base + u * T + t;
      if (i < np) {
        pred_src[i] = slot[u];
        pred_dst[i] = slot[u];
        // only the predicted BOXES are needed now; a matched track is re-predicted inside its update (MOT_KF_PREDICT_FIRST)
        const bool zero_v7 = st[u] != Tracked;
        pred_flags[i] = (zero_v7 ? MOT_KF_ZERO_V7 : 0) | MOT_KF_NO_STORE;
        float4 v = mv[u];
        if (zero_v7) v.w = 0.0f;  // (predicted_box)
        store_box(pool_box, CAP, i, xyah_box4(ma[u].x + v.x, ma[u].y + v.y, ma[u].z + v.z, ma[u].w + v.w));
      }
    }
  }
  for (int i = t; i < nu; i += T) store_box(ubox, CAP, i, stored_box(kdense, unconf_slot[i]));  // (third association)
  if (t == 0) {
    S.n_high = nh; S.n_second = ns; S.n_pool = np; S.n_tracked = n_tracked; S.n_unconf = nu;
    S.n_upd = 0; S.n_refind = 0; S.n_utrack = 0; S.n_udet = 0; S.n_r = 0; S.n_init = 0; S.n_lost_new = 0; S.lap2_q = 0; S.lap3_q = 0;
    det_t[blockIdx.x].dets = dets; det_t[blockIdx.x].ld = ldd; det_t[blockIdx.x].n = (n <= D) ? n : 0;
    pred_t[blockIdx.x].n = np;
    mot_lap_task& L = lap1_t[blockIdx.x];
    const bool q = np > 0 && nh > 0;
    L.n = q ? np : 0; L.m = q ? nh : 0;
    L.geom.n = L.n; L.geom.m = L.m;
    L.geom.bconf = conf;  // score fusion reads the frame's confidences (:300-312)
    if (q) atomicMax(&maxt[64 + (blockIdx.x & 63)], np);
    if (stats && q) {  // 64 counter sets, so that thousands of streams do not serialise on one address
      unsigned long long* st = stats + (blockIdx.x & 63) * 8;
      atomicAdd(&st[0], 1ull); atomicAdd(&st[1], static_cast<unsigned long long>(np + nh)); atomicAdd(&st[4], static_cast<unsigned long long>(np));
    }
  }
}

// ---- K1: apply the first association, queue the second and the unconfirmed one (:267-455) ----
// Four wavefronts per stream (round 3; one wavefront walked the 1000-row pool in 16 dependent rounds: 215 us per launch at the
// north-star shape). The lists stay in the reference's order: an append position = entries before this round + entries of the earlier
// wavefronts of the round + the lane's rank inside its wavefront (ballot + popcount); the wavefronts' counts meet in LDS.
__global__ void __launch_bounds__(kAFMax) bt_after_first(BtStream* streams, BtParams P, int CAP, mot_kf_task* box_t, mot_lap_task* lap23_t,
                                                      unsigned long long* stats, int* maxt, int* declined) {
  __shared__ int cnt[kAFMax / 64][3];
  BtStream& S = streams[blockIdx.x];
  const int t = static_cast<int>(threadIdx.x);
  if (blockIdx.x == 0 && t == 0) *declined = 0;
  if (S.skip) {
    if (t == 0) {
      box_t[2 * blockIdx.x + 0].n = 0; box_t[2 * blockIdx.x + 1].n = 0;
      for (int k = 0; k < 2; ++k) { mot_lap_task& L = lap23_t[2 * blockIdx.x + k]; L.n = 0; L.m = 0; L.geom.n = 0; L.geom.m = 0; }
    }
    return;
  }
  const int np = S.n_pool, nd = S.n_high;
  const bool have = np > 0 && nd > 0;
  int n_upd = 0, n_ref = 0, n_ut = 0, n_ud = 0;
  // Round 6: the loops below were chains of dependent loads — list entry, then the slot's fields, then the detection's — walked one 256-entry chunk at
  // a time: 28 memory round trips per stream, each ~2 us with 6 144 streams in flight (the kernel ran 286 us per launch for 6 MB of traffic). Now a
  // thread takes kU chunks at once: the loads of one LEVEL of all its entries are issued together (the slots of a list are distinct, so reading a later
  // chunk's fields before an earlier chunk's are written changes nothing), the order-preserving compaction — ballots, LDS, barriers, no memory — then
  // runs chunk by chunk as before. The pointers are read from the stream record once (every barrier made the compiler reload them).
  constexpr int kU = 4;
  const int T = static_cast<int>(blockDim.x);
  const int* __restrict__ x1 = S.x1; const int* __restrict__ y1 = S.y1; const int* __restrict__ pool_slot = S.pool_slot; const int* __restrict__ high = S.high;
  int* __restrict__ t_state = S.t_state; int* __restrict__ t_tlen = S.t_tlen; int* __restrict__ t_fid = S.t_fid; int* __restrict__ t_act = S.t_act;
  int* __restrict__ t_cls = S.t_cls; int* __restrict__ t_det = S.t_det; float* __restrict__ t_conf = S.t_conf;
  int* __restrict__ upd_src = S.upd_src; int* __restrict__ upd_dst = S.upd_dst; int* __restrict__ upd_meas = S.upd_meas; unsigned char* __restrict__ upd_flags = S.upd_flags;
  const unsigned char* __restrict__ pred_flags = S.pred_flags; int* __restrict__ refind = S.refind; int* __restrict__ u_track = S.u_track; int* __restrict__ u_det = S.u_det;
  const float* __restrict__ dconf = S.dets + static_cast<size_t>(4) * S.ld; const float* __restrict__ dcls = S.dets + static_cast<size_t>(5) * S.ld;
  const int frame_count = S.frame_count, n_tracked = S.n_tracked;
  for (int base = 0; base < np; base += kU * T) {
    int xs[kU], slot[kU], st[kU], det[kU], tl[kU];
    unsigned char pf[kU];
    float cf[kU], cl[kU];
    bool v[kU], m[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = base + u * T + t;
      v[u] = i < np;
      xs[u] = (v[u] && have) ? x1[i] : -1;
      slot[u] = v[u] ? pool_slot[i] : 0;
      pf[u] = v[u] ? pred_flags[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      m[u] = v[u] && xs[u] >= 0;
      st[u] = m[u] ? t_state[slot[u]] : 0;
      tl[u] = m[u] ? t_tlen[slot[u]] : 0;
      det[u] = m[u] ? high[xs[u]] : 0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      cf[u] = m[u] ? dconf[det[u]] : 0.0f;
      cl[u] = m[u] ? dcls[det[u]] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (base + u * T >= np) break;  // (uniform)
      const int i = base + u * T + t;
      const bool was_tracked = m[u] && st[u] == Tracked;
      const bool rf = m[u] && !was_tracked;
      const bool um = v[u] && xs[u] < 0;
      const Compact3 c = compact3_block(m[u], rf, um, n_upd, n_ref, n_ut, cnt);
      if (m[u]) {
        const int pu = c.pos[0], sl = slot[u];
        upd_src[pu] = sl; upd_dst[pu] = sl; upd_meas[pu] = det[u];
        upd_flags[pu] = (pf[u] & MOT_KF_ZERO_V7) | MOT_KF_PREDICT_FIRST;  // the update starts from the PREDICTED copy (:251-265)
        // STrack::update :71-89 / re_activate :55-69
        t_tlen[sl] = was_tracked ? tl[u] + 1 : 0;
        t_fid[sl] = frame_count;
        t_state[sl] = Tracked; t_act[sl] = 1;
        t_conf[sl] = cf[u];
        t_cls[sl] = static_cast<int>(cl[u]);
        t_det[sl] = det[u];
      }
      if (rf) refind[c.pos[1]] = slot[u];
      if (um) u_track[c.pos[2]] = i;
    }
  }
  {
    int z1 = 0, z2 = 0;
    for (int base = 0; base < nd; base += kU * T) {
      bool un[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int j = base + u * T + t; un[u] = j < nd && (!have || y1[j] < 0); }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (base + u * T >= nd) break;
        const Compact3 c = compact3_block(un[u], false, false, n_ud, z1, z2, cnt);
        if (un[u]) u_det[c.pos[0]] = base + u * T + t;
      }
    }
  }
  __syncthreads();
  // second association: the still-Tracked, still-unmatched pool members that came from the active list (:367-442)
  int n_r = 0;
  {
    int z1 = 0, z2 = 0;
    int* __restrict__ r_slot = S.r_slot; int* __restrict__ r_pool = S.r_pool; float* __restrict__ rbox = S.rbox; const float* __restrict__ kdense = S.kdense;
    for (int base = 0; base < n_ut; base += kU * T) {
      int pi[kU], sl[kU], stt[kU];
      float4 bx[kU];
      bool in[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int k = base + u * T + t; in[u] = k < n_ut; pi[u] = in[u] ? u_track[k] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) sl[u] = in[u] ? pool_slot[pi[u]] : 0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool cand = in[u] && pi[u] < n_tracked;
        stt[u] = cand ? t_state[sl[u]] : Removed;
        bx[u] = cand ? stored_box(kdense, sl[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (base + u * T >= n_ut) break;
        const bool r = in[u] && pi[u] < n_tracked && stt[u] == Tracked;
        const Compact3 c = compact3_block(r, false, false, n_r, z1, z2, cnt);
        if (r) { r_slot[c.pos[0]] = sl[u]; r_pool[c.pos[0]] = pi[u]; store_box(rbox, CAP, c.pos[0], bx[u]); }
      }
    }
  }
  {
    int* __restrict__ rem = S.rem;
    for (int base = 0; base < n_ud; base += kU * T) {
      int j[kU], h[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int k = base + u * T + t; j[u] = (k < n_ud) ? u_det[k] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int k = base + u * T + t; h[u] = (k < n_ud) ? high[j[u]] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int k = base + u * T + t; if (k < n_ud) rem[k] = h[u]; }
    }
  }
  __syncthreads();
  if (t == 0) {
    S.n_upd = n_upd; S.n_refind = n_ref; S.n_utrack = n_ut; S.n_udet = n_ud; S.n_r = n_r;
    const bool q2 = S.n_second > 0 && n_r > 0;
    const bool q3 = S.n_unconf > 0 && n_ud > 0;
    S.lap2_q = q2; S.lap3_q = q3;
    box_t[2 * blockIdx.x + 0].n = q2 ? n_r : 0;
    box_t[2 * blockIdx.x + 1].n = q3 ? S.n_unconf : 0;
    mot_lap_task& A = lap23_t[2 * blockIdx.x + 0];
    A.n = q2 ? n_r : 0; A.m = q2 ? S.n_second : 0; A.geom.n = A.n; A.geom.m = A.m;
    mot_lap_task& B = lap23_t[2 * blockIdx.x + 1];
    B.n = q3 ? S.n_unconf : 0; B.m = q3 ? n_ud : 0; B.geom.n = B.n; B.geom.m = B.m;
    B.geom.bconf = S.dets + static_cast<size_t>(4) * S.ld;
    {
      const int mn = (A.n > B.n) ? A.n : B.n, mm = (A.m > B.m) ? A.m : B.m;
      if (mn > 0) { atomicMax(&maxt[128 + (blockIdx.x & 63)], mn); atomicMax(&maxt[192 + (blockIdx.x & 63)], mm); }
    }
    if (stats) {
      unsigned long long* st = stats + (blockIdx.x & 63) * 8;
      const int cnt = (q2 ? 1 : 0) + (q3 ? 1 : 0);
      if (cnt) {
        atomicAdd(&st[2], static_cast<unsigned long long>(cnt)); atomicAdd(&st[3], static_cast<unsigned long long>(A.n + A.m + B.n + B.m));
        atomicAdd(&st[5], static_cast<unsigned long long>(A.n + B.n));
      }
    }
  }
}

// ---- K2: apply associations 2 and 3, births, deaths, list algebra, queue the Kalman work (:442-580) ----
__global__ void __launch_bounds__(kAFMax) bt_after_second(BtStream* streams, BtParams P, int CAP, mot_kf_task* init_t, mot_kf_task* upd_t,
                                                       mot_kf_task* box2_t, mot_iou_task* dup_t, unsigned long long* stats, mot_kf_task* updf_t) {
  // Four wavefronts per stream. The first part (second association, unconfirmed tracks, births) walks lists of a few dozen entries and
  // hands out slots in order: the first wavefront does it alone; the list algebra over the ~800 active tracks is shared by all four.
  __shared__ int cnt[kAFMax / 64][3];
  __shared__ int sh[8];
  BtStream& S = streams[blockIdx.x];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip) {
    if (t == 0) {
      init_t[blockIdx.x].n = 0; upd_t[blockIdx.x].n = 0; updf_t[blockIdx.x].n = 0;
      box2_t[2 * blockIdx.x + 0].n = 0; box2_t[2 * blockIdx.x + 1].n = 0;
      dup_t[blockIdx.x].n = 0; dup_t[blockIdx.x].m = 0;
    }
    return;
  }
  auto wave_sync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };  // (one wavefront: its loads have landed before its next stores go out)
  int n_upd = S.n_upd, n_ln = 0;
  int n_init = 0;
  int free_top = S.n_free, next_slot = S.next_slot, err = 0;
  if (t < kW) {
  if (S.lap2_q) {
    for (int i0 = 0; i0 < S.n_r; i0 += kW) {
      const int i = i0 + t;
      const bool v = i < S.n_r;
      const int slot = v ? S.r_slot[i] : 0;
      const int j = v ? S.x2[i] : -1;
      const bool m = v && j >= 0;
      const int pu = compact(m, n_upd);
      if (m) {
        const int det = S.second[j];
        S.upd_src[pu] = slot; S.upd_dst[pu] = slot; S.upd_meas[pu] = det;
        S.upd_flags[pu] = (S.pred_flags[S.r_pool[i]] & MOT_KF_ZERO_V7) | MOT_KF_PREDICT_FIRST;
        S.t_fid[slot] = S.frame_count; S.t_tlen[slot] += 1;  // state is Tracked here
        S.t_state[slot] = Tracked; S.t_act[slot] = 1;
        S.t_conf[slot] = S.dets[static_cast<size_t>(4) * S.ld + det];
        S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
        S.t_det[slot] = det;
      }
      const bool l = v && j < 0 && S.t_state[slot] != Lost;
      const int pl = compact(l, n_ln);
      if (l) { S.t_state[slot] = Lost; S.lost_new[pl] = slot; }
    }
  }
  // unconfirmed tracks vs. the leftover high detections (:455-542); u_det_final = detections nobody took
  int* udf = S.y3;  // reuse: final unmatched detection list (indices into high) is written over y3 once it has been read
  int n_udf = 0;
  if (S.lap3_q) {
    // first gather the unmatched columns (read y3 completely before overwriting it: one chunk at a time, in order)
    for (int j0 = 0; j0 < S.n_udet; j0 += kW) {
      const int j = j0 + t;
      const bool u = j < S.n_udet && S.y3[j] < 0;
      const int val = (j < S.n_udet) ? S.u_det[j] : 0;
      wave_sync();
      const int p = compact(u, n_udf);  // p <= j: never overwrites an unread entry
      if (u) udf[p] = val;
      wave_sync();
    }
    for (int i0 = 0; i0 < S.n_unconf; i0 += kW) {
      const int i = i0 + t;
      const bool v = i < S.n_unconf;
      const int slot = v ? S.unconf_slot[i] : 0;
      const int j = v ? S.x3[i] : -1;
      const bool m = v && j >= 0;
      const int pu = compact(m, n_upd);
      if (m) {
        const int det = S.high[S.u_det[j]];
        S.upd_src[pu] = slot; S.upd_dst[pu] = slot; S.upd_meas[pu] = det;  // un-predicted state (:524-527)
        S.upd_flags[pu] = 0;
        if (S.t_state[slot] == Tracked) { S.t_fid[slot] = S.frame_count; S.t_tlen[slot] += 1; }
        else { S.t_tlen[slot] = 0; S.t_fid[slot] = S.frame_count; }
        S.t_state[slot] = Tracked; S.t_act[slot] = 1;
        S.t_conf[slot] = S.dets[static_cast<size_t>(4) * S.ld + det];
        S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
        S.t_det[slot] = det;
      }
      if (v && j < 0) S.t_state[slot] = Removed;
    }
  } else {
    for (int j = t; j < S.n_udet; j += kW) udf[j] = S.u_det[j];
    n_udf = S.n_udet;
  }
  wave_sync();
  // births (:546-554): ids in list order
  for (int j0 = 0; j0 < n_udf; j0 += kW) {
    const int k = j0 + t;
    const int det = (k < n_udf) ? S.high[udf[k]] : 0;
    const float c = (k < n_udf) ? S.dets[static_cast<size_t>(4) * S.ld + det] : 0.f;
    const bool b = k < n_udf && c >= P.det_thresh;
    const int base0 = n_init;
    const int p = compact(b, n_init);
    const int births = n_init - base0;
    // slots: from the free stack first, then fresh ones (which physical slot a track gets is not observable)
    int slot = -1;
    if (b) {
      const int r = p - base0;
      if (r < free_top) slot = S.free_stack[free_top - 1 - r];
      else { slot = next_slot + (r - free_top); if (slot >= CAP) { slot = CAP - 1; err = 1; } }
      S.t_id[slot] = S.next_id + p + 1;
      S.t_conf[slot] = c;
      S.t_cls[slot] = static_cast<int>(S.dets[static_cast<size_t>(5) * S.ld + det]);
      S.t_det[slot] = det;
      S.t_tlen[slot] = 0; S.t_state[slot] = Tracked;
      S.t_act[slot] = (S.frame_count == 1) ? 1 : 0;
      S.t_fid[slot] = S.frame_count; S.t_sf[slot] = S.frame_count;
      S.init_dst[p] = slot; S.init_meas[p] = det;
    }
    const int from_free = (births < free_top) ? births : free_top;
    next_slot += births - from_free;
    free_top -= from_free;
  }
  err = __any(err) ? 1 : 0;
  if (t == 0) { sh[0] = n_upd; sh[1] = n_ln; sh[2] = n_init; sh[3] = free_top; sh[4] = next_slot; sh[5] = err; }
  }
  __syncthreads();
  n_upd = sh[0]; n_ln = sh[1]; n_init = sh[2]; free_top = sh[3]; next_slot = sh[4]; err = sh[5];
  const int n_active = S.n_active, n_lost = S.n_lost, n_refind = S.n_refind;
  const int* lst = S.lost[S.cur];
  const int* act = S.active[S.cur];
  // (round 6: kU chunks per thread, the loads of a level issued together — see bt_after_first)
  constexpr int kU = 4;
  const int T = static_cast<int>(blockDim.x);
  int* __restrict__ t_state = S.t_state; const int* __restrict__ t_fid = S.t_fid; const int* __restrict__ t_sf = S.t_sf; int* __restrict__ free_stack = S.free_stack;
  const int frame_count = S.frame_count;
  for (int base = 0; base < n_lost; base += kU * T) {  // :557-562
    int slot[kU], fid[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < n_lost) ? lst[i] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; fid[u] = (i < n_lost) ? t_fid[slot[u]] : frame_count; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; if (i < n_lost && frame_count - fid[u] > P.max_time_lost) t_state[slot[u]] = Removed; }
  }
  __syncthreads();
  // list algebra (:565-580). A track lives in exactly one record, so the reference's copies are moves.
  int* __restrict__ na = S.active[S.cur ^ 1];
  int* __restrict__ nl = S.lost[S.cur ^ 1];
  int n_na = 0, n_nl = 0, z = 0;
  for (int base = 0; base < n_active; base += kU * T) {
    int slot[kU], st[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < n_active) ? act[i] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; st[u] = (i < n_active) ? t_state[slot[u]] : -1; }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (base + u * T >= n_active) break;  // (uniform)
      const bool k = st[u] == Tracked;
      const bool dead = st[u] == Removed;
      const Compact3 c = compact3_block(k, dead, false, n_na, free_top, z, cnt);
      if (k) na[c.pos[0]] = slot[u];
      if (dead) free_stack[c.pos[1]] = slot[u];
    }
  }
  // the new tracks' Kalman records (round 5: here instead of a launch of kf_kernel<XYAH, initiate> — a handful of births per stream and frame;
  // KalmanFilterXYAH::initiate as kf_kernels.hip::s8_init<MOT_KF_XYAH> writes it: mean = (z, 0), P = diag(sd^2))
  for (int i = t; i < n_init; i += static_cast<int>(blockDim.x)) {
    const int slot = S.init_dst[i], det = S.init_meas[i];
    const float* zm = init_t[blockIdx.x].meas;
    const int ldm = init_t[blockIdx.x].ldm;
    const float z0 = zm[det], z1 = zm[static_cast<size_t>(ldm) + det], z2 = zm[static_cast<size_t>(2) * ldm + det], h = zm[static_cast<size_t>(3) * ldm + det];
    constexpr float kWp = 1.0f / 20.0f, kWv = 1.0f / 160.0f;
    float sd[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sd[q] = (q < 4) ? 2.0f * kWp * h : 10.0f * kWv * h;
    sd[2] = 1e-2f; sd[6] = 1e-5f;
    float4* dn = reinterpret_cast<float4*>(S.kdense + static_cast<size_t>(slot) * 8);
    dn[0] = make_float4(z0, z1, z2, h);
    dn[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // P = diag(sd^2) in block form: block c = {sd_c^2, 0, 0, sd_(c+4)^2}
    float4* blk = reinterpret_cast<float4*>(S.kblk + static_cast<size_t>(slot) * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) blk[c] = make_float4(sd[c] * sd[c], 0.0f, 0.0f, sd[c + 4] * sd[c + 4]);
    S.kflag[slot] = 0;
  }
  if (n_na + n_init + n_refind > CAP) err = 1;
  else {
    for (int i = t; i < n_init; i += static_cast<int>(blockDim.x)) na[n_na + i] = S.init_dst[i];
    n_na += n_init;
    for (int i = t; i < n_refind; i += static_cast<int>(blockDim.x)) na[n_na + i] = S.refind[i];  // re-found lost tracks, in match order
    n_na += n_refind;
  }
  for (int base = 0; base < n_lost; base += kU * T) {
    int slot[kU], st[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < n_lost) ? lst[i] : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; st[u] = (i < n_lost) ? t_state[slot[u]] : -1; }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (base + u * T >= n_lost) break;
      const bool k = st[u] == Lost;          // Tracked = re-found (now active), Removed = aged out
      const bool dead = st[u] == Removed;
      const Compact3 c = compact3_block(k, dead, false, n_nl, free_top, z, cnt);
      if (k) nl[c.pos[0]] = slot[u];
      if (dead) free_stack[c.pos[1]] = slot[u];
    }
  }
  if (n_nl + n_ln > CAP) err = 1;
  else {
    for (int i = t; i < n_ln; i += T) nl[n_nl + i] = S.lost_new[i];
    n_nl += n_ln;
  }
  __syncthreads();
  {
    int* __restrict__ age_a = S.age_a; int* __restrict__ age_b = S.age_b; unsigned char* __restrict__ dup_a = S.dup_a; unsigned char* __restrict__ dup_b = S.dup_b;
    const int ea = (n_na < CAP) ? n_na : CAP, eb = (n_nl < CAP) ? n_nl : CAP;
    for (int base = 0; base < ea; base += kU * T) {
      int slot[kU], f[kU], sf[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < ea) ? na[i] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; f[u] = (i < ea) ? t_fid[slot[u]] : 0; sf[u] = (i < ea) ? t_sf[slot[u]] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; if (i < ea) { age_a[i] = f[u] - sf[u]; dup_a[i] = 0; } }
    }
    for (int base = 0; base < eb; base += kU * T) {
      int slot[kU], f[kU], sf[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; slot[u] = (i < eb) ? nl[i] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; f[u] = (i < eb) ? t_fid[slot[u]] : 0; sf[u] = (i < eb) ? t_sf[slot[u]] : 0; }
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int i = base + u * T + t; if (i < eb) { age_b[i] = f[u] - sf[u]; dup_b[i] = 0; } }
    }
  }
  if (t == 0) {
    S.n_upd = n_upd; S.n_init = n_init; S.n_lost_new = n_ln;
    S.next_id += n_init; S.next_slot = next_slot; S.n_free = free_top;
    S.n_active = n_na; S.n_lost = n_nl; S.cur ^= 1;
    if (err) S.err = 1;
    init_t[blockIdx.x].n = n_init;
    upd_t[blockIdx.x].n = n_upd;
    updf_t[blockIdx.x].n = 0;  // (the dense-form updates of this frame: appended by kf_update_blocks_kernel)
    if (stats) {  // stats[6] / stats[7]: Kalman updates / initiations queued (profile leg: bytes moved by those launches)
      unsigned long long* st = stats + (blockIdx.x & 63) * 8;
      atomicAdd(&st[6], static_cast<unsigned long long>(n_upd)); atomicAdd(&st[7], static_cast<unsigned long long>(n_init));
    }
    mot_kf_task& BA = box2_t[2 * blockIdx.x + 0];
    BA.n = n_na; BA.src = na;
    mot_kf_task& BL = box2_t[2 * blockIdx.x + 1];
    BL.n = (n_na > 0 && n_nl > 0) ? n_nl : 0; BL.src = nl;
    mot_iou_task& U = dup_t[blockIdx.x];
    U.n = (n_na > 0 && n_nl > 0) ? n_na : 0; U.m = (n_na > 0 && n_nl > 0) ? n_nl : 0;
  }
}

// ---- duplicate marking (remove_duplicate_stracks :659-706): iou_distance(active, lost) < 0.15 -> the younger one goes.
// Same pair arithmetic as the N x M cost kernel (cost_math.hpp), one workgroup per stream over its own na x nl pairs
// (a launch of the tiled cost kernel over the capacity bound spends 0.5 ms on tiles that exit at once).
// A thread owns an active track (box in registers). Duplicates need IoU > 0.85, so a pair is first put through iou_pair's own
// intersection and union without the division: only pairs with inter > 0.8 * union reach the exact arithmetic.
// SORTED (the lost boxes fit in LDS): evaluating all na x nl prefilters was 10 % of a north-star frame's kernel time, pure
// VALU. inter > 0.8 * union with positive widths forces |x1_a - x1_b| < 0.25 * w_a (inter <= iw * min(h) and
// union >= max(area) give iw > 0.8 * max(w_a, w_b), and iw <= w - |x1_a - x1_b| on the side that starts later), so the
// lost boxes are ranked by x1 in LDS and an active box looks only at the ranks inside x1_a -/+ 0.3 * w_a (binary search;
// the margin dwarfs any rounding). Lost boxes with a non-finite coordinate are ranked first and always visited; an active
// box whose width is not a positive finite number visits everything. `verify` (MOT_BT_DUPS_VERIFY=1, tests): the full scan
// runs as well and any pair it would mark outside the window raises the stream's error flag 3.
template <int MODE>  // 0: lost boxes read from global, all pairs; 1: staged in LDS, sorted window; 2: staged in LDS, all pairs
__device__ __forceinline__ void bt_dups_body(BtStream& S, int CAP, int verify, int lds_items) {
  extern __shared__ __attribute__((aligned(16))) float sbox[];  // [nl] float4 boxes, [nl] sorted x1 keys, [nl] sorted indices, [nl] keys
  const int na = S.n_active, nl = S.n_lost, T = static_cast<int>(blockDim.x);
  if (S.skip || na <= 0 || nl <= 0) return;
  const int* act = S.active[S.cur];  // (bt_after_second has made the new lists current; the Kalman updates and initiations of the frame are done)
  const int* lst = S.lost[S.cur];
  // the launch reserves LDS for lds_items lost boxes (far more than a stream usually has); a stream with more reads them from
  // global memory and tests every pair
  const bool staged = MODE != 0 && nl <= lds_items;
  float4* wl = reinterpret_cast<float4*>(sbox);
  float* sx = sbox + static_cast<size_t>(4) * nl;
  int* si = reinterpret_cast<int*>(sx + nl);
  __shared__ int n_irr;
  if (MODE == 2 && staged) {
    for (int j = threadIdx.x; j < nl; j += T)
      wl[j] = stored_box(S.kdense, lst[j]);
    __syncthreads();
  }
  if (MODE == 1 && staged) {
    if (threadIdx.x == 0) n_irr = 0;
    for (int base = 0; base < nl; base += 4 * T) {
      int sl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = base + u * T + static_cast<int>(threadIdx.x); sl[u] = (j < nl) ? lst[j] : 0; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = base + u * T + static_cast<int>(threadIdx.x); if (j < nl) wl[j] = stored_box(S.kdense, sl[u]); }
    }
    __syncthreads();
    // rank by (key, index) with key = x1, or -inf for a box with a non-finite coordinate: nl is a few hundred at most,
    // counting against the key array (one broadcast LDS read and three VALU operations per comparison) beats a sorting network
    float* key = reinterpret_cast<float*>(si + nl);
    for (int j = threadIdx.x; j < nl; j += T) {
      const float4 b = wl[j];
      const bool irr = !(fabsf(b.x) < 3.0e38f && fabsf(b.y) < 3.0e38f && fabsf(b.z) < 3.0e38f && fabsf(b.w) < 3.0e38f);
      key[j] = irr ? -3.4e38f : b.x;
      if (irr) atomicAdd(&n_irr, 1);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nl; j += T) {
      const float kj = key[j];
      int rank = 0;
#pragma unroll 8
      for (int k = 0; k < nl; ++k) {
        const float kk = key[k];
        rank += (kk < kj || (kk == kj && k < j)) ? 1 : 0;
      }
      sx[rank] = kj;
      si[rank] = j;
    }
    __syncthreads();
  }
  // (round 6: the boxes and ages of kU of a thread's active tracks are fetched together — list entry, then the slot's mean: two dependent loads that
  // four chunks used to pay one after the other)
  constexpr int kU = 4;
  for (int base = 0; base < na; base += kU * T) {
  int slot_u[kU], age_u[kU];
  float4 box_u[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) { const int i = base + u * T + static_cast<int>(threadIdx.x); slot_u[u] = (i < na) ? act[i] : 0; age_u[u] = (i < na) ? S.age_a[i] : 0; }
#pragma unroll
  for (int u = 0; u < kU; ++u) { const int i = base + u * T + static_cast<int>(threadIdx.x); box_u[u] = (i < na) ? stored_box(S.kdense, slot_u[u]) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int i = base + u * T + static_cast<int>(threadIdx.x);
    if (i >= na) continue;
    const float4 ab = box_u[u];
    const float a[4] = {ab.x, ab.y, ab.z, ab.w};
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    const int age_a = age_u[u];
    bool dup_me = false;
    auto test = [&](int j, bool mark) {
      float4 bb;
      if (staged) bb = wl[j];
      else bb = stored_box(S.kdense, lst[j]);
      const float iw = mot::smax(0.0f, mot::smin(a[2], bb.z) - mot::smax(a[0], bb.x));
      const float ih = mot::smax(0.0f, mot::smin(a[3], bb.w) - mot::smax(a[1], bb.y));
      const float inter = iw * ih;
      const float area_b = (bb.z - bb.x) * (bb.w - bb.y);
      const float uni = area_a + area_b - inter;
      if (inter > 0.8f * uni) {
        const float iou = (uni > 0.0f) ? (inter / uni) : 0.0f;  // iou_pair's value (cost_math.hpp), inter and uni as it computes them
        if (1.0f - iou < 0.15f) {
          if (!mark) return true;
          if (age_a > S.age_b[j]) S.dup_b[j] = 1;
          else dup_me = true;
          return true;
        }
      }
      return false;
    };
    if (MODE == 1 && staged) {
      const float wa = a[2] - a[0];
      const bool regular = wa > 0.0f && wa < 3.0e38f && fabsf(a[0]) < 3.0e38f;
      int lo = 0, hi = nl;
      if (regular) {
        const float xl = a[0] - 0.3f * wa, xh = a[0] + 0.3f * wa;
        const int ni = n_irr;
        for (int j = 0; j < ni; ++j) test(si[j], true);  // lost boxes with non-finite coordinates: no window applies
        int l = ni, h = nl;
        while (l < h) { const int m = (l + h) >> 1; if (sx[m] < xl) l = m + 1; else h = m; }
        lo = l; h = nl;
        while (l < h) { const int m = (l + h) >> 1; if (sx[m] <= xh) l = m + 1; else h = m; }
        hi = l;
      }
      for (int r = lo; r < hi; ++r) test(si[r], true);
      if (verify && regular) {  // the full scan must not find a pair outside the window
        for (int r = n_irr; r < nl; ++r)
          if ((r < lo || r >= hi) && test(si[r], false)) S.err = 3;
      }
    } else {
      for (int j = 0; j < nl; ++j) test(j, true);
    }
    if (dup_me) S.dup_a[i] = 1;
  }
  }
}
template <int MODE>
__global__ void __launch_bounds__(kAFMax) bt_dups(BtStream* streams, int CAP, int verify, int lds_items) {
  bt_dups_body<MODE>(streams[blockIdx.x], CAP, verify, lds_items);
}

// ---- K3: duplicate removal and the output table (:582-621, :659-706) ----
// Four wavefronts per stream (as bt_after_first): the lists are ~800 entries of dependent loads (slot, then the slot's fields), which one
// wavefront walks in 13 rounds.
__device__ __forceinline__ void bt_finish_body(BtStream& S, int CAP, float* out, int* out_counts, int cap_out, int* max_tracks, int* alive, int* err) {
  __shared__ int cnt[kAFMax / 64][3];
  const int t = static_cast<int>(threadIdx.x);
  if (S.skip) {  // its tracks still bound the next frame's launches
    if (t == 0) {
      out_counts[blockIdx.x] = 0; alive[blockIdx.x] = S.err ? -S.err : S.n_active + S.n_lost; atomicMax(&max_tracks[blockIdx.x & 63], S.n_active + S.n_lost);
      if (S.err) atomicMax(err, S.err);
    }
    return;
  }
  int* act = S.active[S.cur];
  int* lst = S.lost[S.cur];
  float* rows = out + static_cast<size_t>(blockIdx.x) * cap_out * 8;
  const int n_active = S.n_active, n_lost = S.n_lost;
  const bool dups = n_active > 0 && n_lost > 0;
  int free_top = S.n_free;
  int n_keep = 0, n_rows = 0;
  // (round 6: kU chunks per thread, the loads of a level issued together — see bt_after_first. The lists are compacted in place: an entry lands at
  // p <= i, so the entries a thread has read ahead are never ones an earlier chunk overwrites)
  constexpr int kU = 4;
  const int T = static_cast<int>(blockDim.x);
  const unsigned char* __restrict__ dup_a = S.dup_a; const unsigned char* __restrict__ dup_b = S.dup_b;
  const int* __restrict__ t_act = S.t_act; const int* __restrict__ t_id = S.t_id; const int* __restrict__ t_cls = S.t_cls; const int* __restrict__ t_det = S.t_det;
  const float* __restrict__ t_conf = S.t_conf; const float* __restrict__ kdense = S.kdense; int* __restrict__ free_stack = S.free_stack;
  for (int base = 0; base < n_active; base += kU * T) {
    int slot[kU], ta[kU], rid[kU], rcls[kU], rdet[kU];
    float rconf[kU];
    float4 ob[kU];
    bool v[kU], dup[kU], emit[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = base + u * T + t;
      v[u] = i < n_active;
      slot[u] = v[u] ? act[i] : 0;
      dup[u] = v[u] && dups && dup_a[i] != 0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) ta[u] = (v[u] && !dup[u]) ? t_act[slot[u]] : 0;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      emit[u] = v[u] && !dup[u] && ta[u] != 0;
      ob[u] = emit[u] ? stored_box(kdense, slot[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
      rid[u] = emit[u] ? t_id[slot[u]] : 0; rconf[u] = emit[u] ? t_conf[slot[u]] : 0.f;
      rcls[u] = emit[u] ? t_cls[slot[u]] : 0; rdet[u] = emit[u] ? t_det[slot[u]] : 0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (base + u * T >= n_active) break;  // (uniform)
      const bool keep = v[u] && !dup[u];
      const Compact3 c = compact3_block(keep, emit[u], dup[u], n_keep, n_rows, free_top, cnt);
      if (keep) act[c.pos[0]] = slot[u];
      if (emit[u] && c.pos[1] < cap_out) {
        float4* r = reinterpret_cast<float4*>(rows + static_cast<size_t>(c.pos[1]) * 8);
        r[0] = ob[u];
        r[1] = make_float4(static_cast<float>(rid[u]), rconf[u], static_cast<float>(rcls[u]), static_cast<float>(rdet[u]));
      }
      if (dup[u]) free_stack[c.pos[2]] = slot[u];
    }
  }
  int n_keep_l = 0;
  {
    int z = 0;
    for (int base = 0; base < n_lost; base += kU * T) {
      int slot[kU];
      bool v[kU], dup[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = base + u * T + t;
        v[u] = i < n_lost;
        slot[u] = v[u] ? lst[i] : 0;
        dup[u] = v[u] && dups && dup_b[i] != 0;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (base + u * T >= n_lost) break;
        const bool keep = v[u] && !dup[u];
        const Compact3 c = compact3_block(keep, dup[u], false, n_keep_l, free_top, z, cnt);
        if (keep) lst[c.pos[0]] = slot[u];
        if (dup[u]) free_stack[c.pos[1]] = slot[u];
      }
    }
  }
  if (t == 0) {
    S.n_active = n_keep; S.n_lost = n_keep_l; S.n_free = free_top;
    if (n_rows > cap_out) S.err = 2;
    out_counts[blockIdx.x] = (n_rows <= cap_out) ? n_rows : -n_rows;
    // tracks alive after this frame: an exact upper bound of every problem side of the next frame (64 slots: no hot address)
    atomicMax(&max_tracks[blockIdx.x & 63], n_keep + n_keep_l);
    const int e = S.err;
    alive[blockIdx.x] = e ? -e : n_keep + n_keep_l;  // (a stream in error reports -(error code): its caller alone gets the error)  // the batch's error word (round 4: gathered here; a kernel of its own before — one launch of a frame's critical path)
    if (e) atomicMax(err, e);
  }
}
__global__ void __launch_bounds__(kAFMax) bt_finish(BtStream* streams, int CAP, float* out, int* out_counts, int cap_out, int* max_tracks, int* alive, int* err) {
  bt_finish_body(streams[blockIdx.x], CAP, out, out_counts, cap_out, max_tracks, alive, err);
}
// a handful of streams (one camera): duplicate marking and the output table in ONE launch — the same workgroup owns the stream in both, the flags
// it has just written are its own (round 5: a launch is 4-5 us of a 0.2 ms frame; with thousands of streams the two keep their own thread counts)
__global__ void __launch_bounds__(kAFMax) bt_dups_finish(BtStream* streams, int CAP, int verify, int lds_items, float* out, int* out_counts, int cap_out,
                                                          int* max_tracks, int* alive, int* err) {
  BtStream& S = streams[blockIdx.x];
  bt_dups_body<1>(S, CAP, verify, lds_items);
  __syncthreads();
  bt_finish_body(S, CAP, out, out_counts, cap_out, max_tracks, alive, err);
}

}  // namespace

// ---- host side: allocation and the per-frame launch sequence -----------------------------------------------------------
struct mot_bt_batch {
  mot_ctx* ctx = nullptr;
  int S = 0, CAP = 0, D = 0;
  BtParams prm{};
  mot::lifecycle::Allocs mem;
  BtStream* d_streams = nullptr;
  std::vector<BtStream> h_streams;  // host mirror of the pointers (scalars are only valid on the device)
  int* d_counts = nullptr;
  int* d_err = nullptr;
  int* d_decl = nullptr;  // [2] problems the sparse solver declined in the frame's two assignment launches (cleared by the kernel in front of each)
  int* d_maxt = nullptr;  // [4][64] per-frame maxima: tracks alive (bt_finish), pool rows of the first association (bt_begin), rows and
                          // columns of the second / unconfirmed associations (bt_after_first) — bounds and LDS hints of the next frame
  int hint1_n = 0, hint23_n = 0, hint23_m = 0;
  int bound_n = 0;        // upper bound of tracked + lost per stream for the NEXT frame (0 right after creation / reset)
  float* d_out = nullptr; int* d_out_counts = nullptr; int out_cap = 0;
  float* d_packed = nullptr; int* d_offsets = nullptr; int packed_cap = 0;  // mot_bt_step_packed
  // frames in flight (mot_bt_enqueue_packed / mot_bt_collect_packed, mot_bt_enqueue_frame / mot_bt_collect_view): lifecycle_common.hpp
  mot::lifecycle::Flights flights;
  int* d_alive = nullptr;  // [S] live tracks per stream after the frame
  const int* d_counts_last = nullptr;  // the frame mot_bt_device_output describes: per-stream row counts, rows, offsets
  const float* d_rows_last = nullptr; const int* d_offsets_last = nullptr;
  mot_det_task* det_t = nullptr;
  mot_kf_task *pred_t = nullptr, *box_t = nullptr, *init_t = nullptr, *upd_t = nullptr, *box2_t = nullptr;
  mot_lap_task *lap1_t = nullptr, *lap23_t = nullptr;
  mot_iou_task* dup_t = nullptr;
  float* mean = nullptr;  // [S][CAP] covariance records of 64 floats (mot_kf_task's slab when mean_dense is set)
  float* mean_dense = nullptr;  // [S][CAP][8] the means (BtStream::kdense)
  float* cov_blocks = nullptr;  // [S][CAP][16] block-form covariances (BtStream::kblk)
  unsigned char* dense_flag = nullptr;  // [S][CAP]
  mot_kf_task* updf_t = nullptr;  // the dense-form updates of a frame (tracks that left the block form), filled on the device
  int* fb_i = nullptr; unsigned char* fb_f = nullptr;  // their lists: [S][3][CAP] slots / slots / measurement columns, [S][CAP] flags
  // profiling (bench.py's roofline leg): HIP events around the two assignment launches and the whole frame
  bool profile = false;
  unsigned long long* d_stats = nullptr;
  hipEvent_t ev[12] = {};
  double lap_sparse1_ms = 0.0;
  double lap_ms[2] = {0.0, 0.0}, frame_ms = 0.0, kf_ms[3] = {0.0, 0.0, 0.0};  // kf_ms: predict(boxes), initiate, update
  long frames = 0;
  template <class T>
  T* dalloc(size_t n) { return mem.get<T>(n); }
};


extern "C" {

void mot_bt_destroy(mot_bt_batch* b) {
  if (!b) return;
  b->flights.release();
  b->mem.release();
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  delete b;
}

int mot_bt_reset(mot_bt_batch* b) {  // ByteTrack::reset: the lists go, the id counter keeps counting (clear_count() is empty, bytetrack.hpp:38-40)
  // scalars back to zero; the arrays need no clearing (everything is rebuilt from the empty lists)
  MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));  // (frames still in flight have finished by now: they are dropped with the tracks)
  std::vector<BtStream> cur(b->S);
  MOT_LC_HIP(b, hipMemcpy(cur.data(), b->d_streams, sizeof(BtStream) * b->S, hipMemcpyDeviceToHost));
  std::vector<BtStream> h = b->h_streams;
  for (int s = 0; s < b->S; ++s) h[s].next_id = cur[s].next_id;
  MOT_LC_HIP(b, hipMemcpy(b->d_streams, h.data(), sizeof(BtStream) * b->S, hipMemcpyHostToDevice));
  MOT_LC_HIP(b, hipMemset(b->d_err, 0, sizeof(int)));
  b->flights.drop_all();
  b->bound_n = 0; b->hint1_n = b->hint23_n = b->hint23_m = 0;
  return MOT_OK;
}

int mot_bt_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, const float* p5, mot_bt_batch** out) {
  if (!ctx || !out || nstreams <= 0 || cap_tracks <= 0 || max_dets <= 0) return MOT_ERR_INVALID;
  auto* b = new mot_bt_batch();
  b->ctx = ctx; b->S = nstreams; b->CAP = cap_tracks; b->D = max_dets;
  const float min_conf = p5 ? p5[0] : 0.1f, track_thresh = p5 ? p5[1] : 0.45f, match_thresh = p5 ? p5[2] : 0.8f;
  const int track_buffer = p5 ? static_cast<int>(p5[3]) : 25, frame_rate = p5 ? static_cast<int>(p5[4]) : 30;
  b->prm.min_conf = min_conf; b->prm.track_thresh = track_thresh; b->prm.match_thresh = match_thresh;
  b->prm.det_thresh = track_thresh;                                                    // bytetrack.cpp:145
  b->prm.max_time_lost = static_cast<int>(frame_rate / 30.0f * track_buffer);           // :141-142
  const int S = nstreams, CAP = cap_tracks, D = max_dets, C2 = cap_tracks;  // (no scratch slots: predictions are box-only)
  const size_t ints_per = static_cast<size_t>(CAP) * 28 + static_cast<size_t>(D) * 9;
  int* ip = b->dalloc<int>(ints_per * S);
  float* fp = b->dalloc<float>((static_cast<size_t>(CAP) * (1 + 4 * 5) + static_cast<size_t>(D) * 12) * S);
  unsigned char* bp = b->dalloc<unsigned char>(static_cast<size_t>(CAP) * 4 * S);
  b->mean = b->dalloc<float>(static_cast<size_t>(S) * 64 * C2);  // (covariance records of 256 bytes; hipMalloc aligns to 256)
  b->mean_dense = b->dalloc<float>(static_cast<size_t>(S) * 8 * C2);
  b->cov_blocks = b->dalloc<float>(static_cast<size_t>(S) * 16 * C2);
  b->dense_flag = b->dalloc<unsigned char>(static_cast<size_t>(S) * C2);
  b->updf_t = b->dalloc<mot_kf_task>(S);
  b->fb_i = b->dalloc<int>(static_cast<size_t>(S) * 3 * C2);
  b->fb_f = b->dalloc<unsigned char>(static_cast<size_t>(S) * C2);
  b->d_streams = b->dalloc<BtStream>(S);
  b->d_counts = b->dalloc<int>(S);
  b->d_err = b->dalloc<int>(1);
  b->d_decl = b->dalloc<int>(2);
  b->d_maxt = b->dalloc<int>(256);
  b->d_alive = b->dalloc<int>(S);
  b->flights.n_maxt = 256; b->flights.with_alive = true;
  b->d_stats = b->dalloc<unsigned long long>(8 * 64);
  if (b->d_stats) (void)hipMemset(b->d_stats, 0, 8 * 64 * sizeof(unsigned long long));
  for (auto& e : b->ev) (void)hipEventCreate(&e);
  b->det_t = b->dalloc<mot_det_task>(S);
  b->pred_t = b->dalloc<mot_kf_task>(S); b->box_t = b->dalloc<mot_kf_task>(2 * S); b->init_t = b->dalloc<mot_kf_task>(S);
  b->upd_t = b->dalloc<mot_kf_task>(S); b->box2_t = b->dalloc<mot_kf_task>(2 * S);
  b->lap1_t = b->dalloc<mot_lap_task>(S); b->lap23_t = b->dalloc<mot_lap_task>(2 * S);
  b->dup_t = b->dalloc<mot_iou_task>(S);
  const size_t wb1 = (mot::lap_scratch_bytes(CAP, D) + 255) & ~size_t(255);
  char* work = b->dalloc<char>(wb1 * 3 * S);
  int* info = b->dalloc<int>(static_cast<size_t>(4) * 3 * S);
  if (!ip || !fp || !bp || !b->mean || !b->mean_dense || !b->cov_blocks || !b->dense_flag || !b->updf_t || !b->fb_i || !b->fb_f || !b->d_streams || !b->d_counts || !b->d_err || !b->d_decl || !b->d_alive || !b->d_maxt || !b->det_t || !b->pred_t || !b->box_t ||
      !b->init_t || !b->upd_t || !b->box2_t || !b->lap1_t || !b->lap23_t || !b->dup_t || !work || !info) {
    mot_bt_destroy(b);
    return MOT_ERR_NOMEM;
  }
  std::vector<BtStream> hs(S);
  std::vector<mot_det_task> det(S);
  std::vector<mot_kf_task> pred(S), box(2 * S), init(S), upd(S), box2(2 * S), updf(S);
  std::vector<mot_lap_task> lap1(S), lap23(2 * S);
  std::vector<mot_iou_task> dup(S);
  for (int s = 0; s < S; ++s) {
    BtStream& T = hs[s];
    std::memset(&T, 0, sizeof(T));
    int* i = ip + ints_per * s;
    auto I = [&](int n) { int* r = i; i += n; return r; };
    T.free_stack = I(CAP); T.active[0] = I(CAP); T.active[1] = I(CAP); T.lost[0] = I(CAP); T.lost[1] = I(CAP);
    T.t_id = I(CAP); T.t_state = I(CAP); T.t_act = I(CAP); T.t_tlen = I(CAP); T.t_fid = I(CAP); T.t_sf = I(CAP); T.t_cls = I(CAP); T.t_det = I(CAP);
    T.pool_slot = I(CAP); T.unconf_slot = I(CAP); T.pred_src = I(CAP); T.pred_dst = I(CAP);
    T.x1 = I(CAP); T.x2 = I(CAP); T.x3 = I(CAP); T.upd_src = I(CAP); T.upd_dst = I(CAP); T.upd_meas = I(CAP);
    T.refind = I(CAP); T.u_track = I(CAP); T.r_slot = I(CAP); T.r_pool = I(CAP); T.lost_new = I(CAP);  // 28 CAP-sized arrays
    T.high = I(D); T.second = I(D); T.y1 = I(D); T.y2 = I(D); T.y3 = I(D); T.u_det = I(D); T.rem = I(D); T.init_dst = I(D); T.init_meas = I(D);
    float* f = fp + (static_cast<size_t>(CAP) * 21 + static_cast<size_t>(D) * 12) * s;
    auto F = [&](int n) { float* r = f; f += n; return r; };
    T.t_conf = F(CAP);
    float* pool_box = F(4 * CAP); float* rbox = F(4 * CAP); float* ubox = F(4 * CAP); T.abox = F(4 * CAP); float* lbox = F(4 * CAP);
    T.lbox = lbox;
    T.pool_box = pool_box; T.rbox = rbox; T.ubox = ubox;
    float* d_box = F(4 * D); float* d_meas = F(4 * D);
    T.dmeas4 = F(4 * D);
    unsigned char* u = bp + static_cast<size_t>(CAP) * 4 * s;
    T.pred_flags = u; T.dup_a = u + CAP; T.dup_b = u + 2 * CAP; T.upd_flags = u + 3 * CAP;
    float* mean = b->mean + static_cast<size_t>(s) * 64 * C2;
    float* cov = mean + 8;
    T.kmean = mean;
    T.kdense = b->mean_dense + static_cast<size_t>(s) * 8 * C2;
    T.kblk = b->cov_blocks + static_cast<size_t>(s) * 16 * C2;
    T.kflag = b->dense_flag + static_cast<size_t>(s) * C2;
    // ---- static parts of the task descriptors ----
    std::memset(&det[s], 0, sizeof(mot_det_task));
    det[s].box = d_box; det[s].ldb = D; det[s].meas = d_meas; det[s].ldm = D;
    auto kf = [&](mot_kf_task& k) { std::memset(&k, 0, sizeof(k)); k.mean = mean; k.cov = cov; k.cap = C2; };
    kf(pred[s]); pred[s].src = T.pred_src; pred[s].dst = T.pred_dst; pred[s].flags = T.pred_flags; pred[s].boxes = pool_box; pred[s].ldb = CAP;
    kf(box[2 * s]); box[2 * s].src = T.r_slot; box[2 * s].boxes = rbox; box[2 * s].ldb = CAP;
    kf(box[2 * s + 1]); box[2 * s + 1].src = T.unconf_slot; box[2 * s + 1].boxes = ubox; box[2 * s + 1].ldb = CAP;
    kf(init[s]); init[s].src = T.init_dst; init[s].dst = T.init_dst; init[s].meas = d_meas; init[s].ldm = D; init[s].midx = T.init_meas;
    kf(upd[s]); upd[s].src = T.upd_src; upd[s].dst = T.upd_dst; upd[s].meas = d_meas; upd[s].ldm = D; upd[s].midx = T.upd_meas;
    upd[s].flags = T.upd_flags; upd[s].mean_dense = T.kdense; upd[s].cov_blocks = T.kblk; upd[s].dense_flag = T.kflag; upd[s].meas4 = T.dmeas4;
    updf[s] = upd[s];  // the same filter on the 64-float records, over the lists the block-form kernel fills
    { int* fi = b->fb_i + static_cast<size_t>(s) * 3 * C2; updf[s].src = fi; updf[s].dst = fi + C2; updf[s].midx = fi + 2 * C2; updf[s].flags = b->fb_f + static_cast<size_t>(s) * C2; updf[s].n = 0; }
    kf(box2[2 * s]); box2[2 * s].boxes = T.abox; box2[2 * s].ldb = CAP;
    kf(box2[2 * s + 1]); box2[2 * s + 1].boxes = lbox; box2[2 * s + 1].ldb = CAP;
    auto lap = [&](mot_lap_task& L, int k, int* x, int* y, const float* a, const int* bidx, const float* bconf, int mode, float thresh) {
      std::memset(&L, 0, sizeof(L));
      L.x = x; L.y = y; L.thresh = thresh; L.mode = MOT_LAP_PLAIN; L.info = info + (static_cast<size_t>(s) * 3 + k) * 4;
      L.work = work + (static_cast<size_t>(s) * 3 + k) * wb1;
      L.geom.a = a; L.geom.lda = CAP; L.geom.b = d_box; L.geom.ldb = D; L.geom.bidx = bidx; L.geom.bconf = bconf; L.geom.mode = mode;
    };
    lap(lap1[s], 0, T.x1, T.y1, pool_box, T.high, nullptr, MOT_COST_IOU_DIST_FUSE, match_thresh);
    lap(lap23[2 * s], 1, T.x2, T.y2, rbox, T.second, nullptr, MOT_COST_IOU_DIST, 0.5f);
    lap(lap23[2 * s + 1], 2, T.x3, T.y3, ubox, T.rem, nullptr, MOT_COST_IOU_DIST_FUSE, 0.7f);
    std::memset(&dup[s], 0, sizeof(mot_iou_task));
    dup[s].a = T.abox; dup[s].lda = CAP; dup[s].b = lbox; dup[s].ldb = CAP; dup[s].mode = MOT_COST_IOU_DIST; dup[s].pair_thresh = 0.15f;
    dup[s].dup_a = T.dup_a; dup[s].dup_b = T.dup_b;
  }
  // age arrays: carve from a separate allocation (kept out of the int pool arithmetic above)
  int* ages = b->dalloc<int>(static_cast<size_t>(2) * CAP * S);
  if (!ages) { mot_bt_destroy(b); return MOT_ERR_NOMEM; }
  for (int s = 0; s < S; ++s) {
    hs[s].age_a = ages + static_cast<size_t>(2) * CAP * s; hs[s].age_b = hs[s].age_a + CAP;
    dup[s].age_a = hs[s].age_a; dup[s].age_b = hs[s].age_b;
  }
  b->h_streams = hs;
  hipStream_t st = ctx->stream;
#define BT_UP(dst, vec) MOT_LC_HIP(b, hipMemcpyAsync(dst, vec.data(), sizeof(vec[0]) * vec.size(), hipMemcpyHostToDevice, st))
  BT_UP(b->d_streams, hs); BT_UP(b->det_t, det); BT_UP(b->pred_t, pred); BT_UP(b->box_t, box); BT_UP(b->init_t, init); BT_UP(b->upd_t, upd); BT_UP(b->updf_t, updf);
  BT_UP(b->box2_t, box2); BT_UP(b->lap1_t, lap1); BT_UP(b->lap23_t, lap23); BT_UP(b->dup_t, dup);
#undef BT_UP
  MOT_LC_HIP(b, hipMemsetAsync(b->d_err, 0, sizeof(int), st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  *out = b;
  return MOT_OK;
}

// enqueues the frame's launches; the per-stream tables land in b->d_out ([S][cap_out][8]) and b->d_out_counts. fd == nullptr: the classic
// form (h_counts is copied to the device here, every stream takes part, detections at s * 6 * max_dets); else the pooled input block
// (already on its way to the device) and h_counts only sizes the launches
static int bt_enqueue_frame(mot_bt_batch* b, const float* d_dets, const int* h_counts, int cap_out, hipEvent_t* ev = nullptr,
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
    MOT_LC_HIP(b, hipMemcpyAsync(b->d_counts, h_counts, sizeof(int) * S, hipMemcpyHostToDevice, st));
    FD.counts = b->d_counts;
  }
  if (!b->flights.maxt_clean) MOT_LC_HIP(b, hipMemsetAsync(b->d_maxt, 0, 256 * sizeof(int), st));  // (else: the last frame's pack_offsets cleared them)
  b->flights.maxt_clean = false;
  // Launch bounds (grid sizes, the solver's LDS layout and variant) from exact upper bounds instead of the capacities:
  // no side of any problem of this frame exceeds the tracks alive after the previous frame (bn) / this frame's detections (bd)
  int bd = 1;
  for (int s = 0; s < S; ++s) bd = (h_counts[s] > bd) ? h_counts[s] : bd;
  if (bd > D) bd = D;
  const int bn = (b->bound_n < 1) ? 1 : (b->bound_n > CAP ? CAP : b->bound_n);
  const int bn2 = (bn + bd > CAP) ? CAP : bn + bd;  // lists after this frame's births
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[0], st));
  // four wavefronts per stream once the lists are long enough to share (short lists: the extra wavefronts only add barriers; 256 x 128: 8.2 M against 8.8 M frames/s)
  const int longest = bn > bd ? bn : bd;
  int active = 0;  // streams with a frame (the pooled form runs the whole segment's grid; a single camera is one of 512)
  for (int s = 0; s < S; ++s) active += (h_counts[s] >= 0) ? 1 : 0;
  const bool few = active <= 16;  // latency over throughput
  const int bt_threads = few ? (longest > 512 ? kAFMax : (longest > 256 ? 512 : (longest > 64 ? kAF : kW))) : ((longest > 384) ? kAF : kW);
  hipLaunchKernelGGL(bt_begin, dim3(S), dim3(bt_threads), 0, st, b->d_streams, b->prm, CAP, D, FD, d_dets, b->det_t, b->pred_t, b->lap1_t, prof ? b->d_stats : nullptr, b->d_maxt, b->d_decl);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[6], st));
  // (round 5: the predicted boxes of the pool come out of bt_begin, the boxes of the other associations' tracks out of bt_after_first, those of
  // the new lists are computed by bt_dups / bt_finish from the updated means: three launches fewer per frame)
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[1], st));
  MOT_LC_HIP(b, mot::launch_lap(b->lap1_t, S, bn, bd, true, false, true, st, b->hint1_n, 0, true, nullptr, prof ? ev[10] : nullptr, b->d_decl, active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[2], st));
  hipLaunchKernelGGL(bt_after_first, dim3(S), dim3(bt_threads), 0, st, b->d_streams, b->prm, CAP, b->box_t, b->lap23_t, prof ? b->d_stats : nullptr, b->d_maxt, b->d_decl + 1);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[3], st));
  MOT_LC_HIP(b, mot::launch_lap(b->lap23_t, 2 * S, bn, bd, true, false, true, st, b->hint23_n, b->hint23_m, true, nullptr, nullptr, b->d_decl + 1, 2 * active));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[4], st));
  hipLaunchKernelGGL(bt_after_second, dim3(S), dim3(bt_threads), 0, st, b->d_streams, b->prm, CAP, b->init_t, b->upd_t, b->box2_t, b->dup_t, prof ? b->d_stats : nullptr, b->updf_t);
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[7], st));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[8], st));
  MOT_LC_HIP(b, mot::launch_kf_update_blocks(b->upd_t, b->updf_t, S, bn, st));
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[9], st));
  {
    static const int verify = std::getenv("MOT_BT_DUPS_VERIFY") != nullptr ? 1 : 0;  // tests: cross-check the sorted window
    static const bool full = std::getenv("MOT_BT_DUPS_FULL") != nullptr;  // measurement aid: every pair, boxes staged in LDS
    // LDS for up to 1024 lost boxes per stream (28 KB: five workgroups per CU); the rare stream with more takes the global path
    const int items = (bn2 < 1024) ? bn2 : 1024;
    static const bool merge_ok = !(std::getenv("MOT_BT_MERGE_FINISH") && std::getenv("MOT_BT_MERGE_FINISH")[0] == '0');  // (A/B measurements)
    if (few && !full && merge_ok) {
      const int td = (bn2 > 256) ? kAFMax : 256;
      hipLaunchKernelGGL(bt_dups_finish, dim3(S), dim3(td > bt_threads ? td : bt_threads), static_cast<size_t>(28) * items, st, b->d_streams, CAP, verify, items,
                         b->d_out, b->d_out_counts, cap_out, b->d_maxt, b->d_alive, b->d_err);
    } else {
      if (!full) hipLaunchKernelGGL(bt_dups<1>, dim3(S), dim3(few && bn2 > 256 ? kAFMax : 256), static_cast<size_t>(28) * items, st, b->d_streams, CAP, verify, items);
      else hipLaunchKernelGGL(bt_dups<2>, dim3(S), dim3(256), static_cast<size_t>(16) * items, st, b->d_streams, CAP, 0, items);
      hipLaunchKernelGGL(bt_finish, dim3(S), dim3(bt_threads), 0, st, b->d_streams, CAP, b->d_out, b->d_out_counts, cap_out, b->d_maxt, b->d_alive, b->d_err);
    }
  }
  if (prof) MOT_LC_HIP(b, hipEventRecord(ev[5], st));
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}

// after the frame's kernels: error flag, launch bounds of the next frame, event times (synchronises the stream)
// LDS hints of the next frame's assignment launches from this frame's largest problems (a quarter more, and some)
static void bt_set_hints(mot_bt_batch* b, const int* maxt) {
  int m1 = 0, m2 = 0, m3 = 0;
  for (int i = 0; i < 64; ++i) { m1 = (maxt[64 + i] > m1) ? maxt[64 + i] : m1; m2 = (maxt[128 + i] > m2) ? maxt[128 + i] : m2; m3 = (maxt[192 + i] > m3) ? maxt[192 + i] : m3; }
  b->hint1_n = m1 > 0 ? m1 + m1 / 8 + 48 : 0;
  b->hint23_n = m2 > 0 ? m2 + m2 / 4 + 32 : 0;
  b->hint23_m = m3 > 0 ? m3 + m3 / 4 + 32 : 0;
}

// adds one frame's event times to the profile sums (the events have completed)
static int bt_account_events(mot_bt_batch* b, hipEvent_t* ev) {
  float ms = 0.f;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[1], ev[2])); b->lap_ms[0] += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[1], ev[10])); b->lap_sparse1_ms += ms;  // the first association's sparse solver alone
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[3], ev[4])); b->lap_ms[1] += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[0], ev[5])); b->frame_ms += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[6], ev[1])); b->kf_ms[0] += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[7], ev[8])); b->kf_ms[1] += ms;
  MOT_LC_HIP(b, hipEventElapsedTime(&ms, ev[8], ev[9])); b->kf_ms[2] += ms;
  b->frames += 1;
  return MOT_OK;
}

static int bt_finish_frame(mot_bt_batch* b) {
  hipStream_t st = b->ctx->stream;
  int err = 0;
  int maxt[256];
  MOT_LC_HIP(b, hipMemcpyAsync(&err, b->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(maxt, b->d_maxt, sizeof(maxt), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  b->bound_n = 0;
  for (int i = 0; i < 64; ++i) b->bound_n = (maxt[i] > b->bound_n) ? maxt[i] : b->bound_n;
  bt_set_hints(b, maxt);
  if (b->profile) {
    const int rce = bt_account_events(b, b->ev);
    if (rce != MOT_OK) return rce;
  }
  if (err) { b->ctx->err = "mot_bt_step: a stream exceeded cap_tracks / max_dets / cap_out"; return MOT_ERR_CAPACITY; }
  return MOT_OK;
}

int mot_bt_step(mot_bt_batch* b, const float* d_dets, const int* h_counts, float* out, int* out_counts, int cap_out) {
  hipStream_t st = b->ctx->stream;
  const int S = b->S;
  if (b->flights.count > 0) { b->ctx->err = "mot_bt_step: frames are in flight (collect them first)"; return MOT_ERR_INVALID; }
  const int rc = bt_enqueue_frame(b, d_dets, h_counts, cap_out, b->profile ? b->ev : nullptr);
  if (rc != MOT_OK) return rc;
  MOT_LC_HIP(b, hipMemcpyAsync(out, b->d_out, sizeof(float) * static_cast<size_t>(S) * cap_out * 8, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(out_counts, b->d_out_counts, sizeof(int) * S, hipMemcpyDeviceToHost, st));
  return bt_finish_frame(b);
}

int mot_bt_step_packed(mot_bt_batch* b, const float* d_dets, const int* h_counts, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  hipStream_t st = b->ctx->stream;
  const int S = b->S;
  // staging holds a stream's whole track list (no per-stream row limit short of cap_tracks); the packed buffer rows_cap rows
  if (b->flights.count > 0) { b->ctx->err = "mot_bt_step_packed: frames are in flight (collect them first)"; return MOT_ERR_INVALID; }
  const int rc = bt_enqueue_frame(b, d_dets, h_counts, b->CAP, b->profile ? b->ev : nullptr);
  if (rc != MOT_OK) return rc;
  if (!b->d_offsets) b->d_offsets = b->dalloc<int>(static_cast<size_t>(S) + 1);
  if (rows_cap > b->packed_cap) { b->d_packed = b->dalloc<float>(static_cast<size_t>(rows_cap) * 8); b->packed_cap = b->d_packed ? rows_cap : 0; }
  if (!b->d_offsets || !b->d_packed) return MOT_ERR_NOMEM;
  hipLaunchKernelGGL(mot::lifecycle::pack_offsets, dim3(1), dim3(1024), 0, st, b->d_out_counts, S, b->d_offsets, mot::lifecycle::PackMeta{});
  hipLaunchKernelGGL(mot::lifecycle::pack_rows, dim3(S), dim3(256), 0, st, b->d_out, b->CAP, b->d_out_counts, b->d_offsets, b->d_packed, rows_cap);
  MOT_LC_HIP(b, hipGetLastError());
  b->d_rows_last = b->d_packed; b->d_offsets_last = b->d_offsets; b->d_counts_last = b->d_out_counts;
  int total = 0;
  MOT_LC_HIP(b, hipMemcpyAsync(out_counts, b->d_out_counts, sizeof(int) * S, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(&total, b->d_offsets + S, sizeof(int), hipMemcpyDeviceToHost, st));
  const int rc2 = bt_finish_frame(b);  // synchronises: total is known
  if (total_rows) *total_rows = total;
  if (rc2 != MOT_OK) return rc2;
  if (total > rows_cap) { b->ctx->err = "mot_bt_step_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  if (total > 0) {
    MOT_LC_HIP(b, hipMemcpyAsync(rows, b->d_packed, sizeof(float) * static_cast<size_t>(total) * 8, hipMemcpyDeviceToHost, st));
    MOT_LC_HIP(b, hipStreamSynchronize(st));
  }
  return MOT_OK;
}

int mot_bt_device_output(mot_bt_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts) {
  if (!b || !b->d_rows_last || !b->d_offsets_last) return MOT_ERR_INVALID;
  if (d_rows) *d_rows = b->d_rows_last;
  if (d_offsets) *d_offsets = b->d_offsets_last;
  if (d_counts) *d_counts = b->d_counts_last;
  return MOT_OK;
}

// ---- frames in flight ------------------------------------------------------------------------------------------------
// mot_bt_step_packed waits for its frame: the host sits idle while the GPU works and the GPU sits idle while the rows
// cross PCIe. Split in two, a single host thread keeps two frames in flight: enqueue(f + 1) returns as soon as the launches
// are queued, collect(f) then waits for frame f only (an event), and copies its rows on a second stream while frame f + 1
// runs. The launch bounds of a frame come from the tracks alive after the last COLLECTED frame plus the detections of the
// frames enqueued since (a stream gains at most one track per detection).
static int bt_enqueue_flight(mot_bt_batch* b, const float* d_dets, const int* h_counts, int rows_cap, const mot_frame_in* in) {
  if (b->flights.count >= 2) { b->ctx->err = "mot_bt_enqueue: two frames are already in flight (collect one first)"; return MOT_ERR_INVALID; }
  hipStream_t st = b->ctx->stream;
  const int S = b->S;
  const int slot = b->flights.slot_for_enqueue();
  int* counts_in = nullptr;
  int bd = 0;
  MOT_LC_HIP(b, b->flights.prepare(b->mem, slot, S, rows_cap, h_counts, &counts_in, &bd, in != nullptr, b->profile));
  mot::lifecycle::Flight& F = b->flights.fl[slot];
  mot::lifecycle::FrameDev fd;
  if (in) MOT_LC_HIP(b, b->flights.upload_block(b->mem, slot, S, in->h_counts, in->h_det_ld, in->h_det_off, nullptr, st, &fd));
  const int saved = b->bound_n;
  b->bound_n = saved + b->flights.pending_bd();  // tracks the frame still in flight may have added
  if (b->bound_n > b->CAP) b->bound_n = b->CAP;
  const int rc = bt_enqueue_frame(b, d_dets, counts_in, b->CAP, F.prof ? F.ev : nullptr, in ? &fd : nullptr);
  b->bound_n = saved;
  if (rc != MOT_OK) return rc;
  MOT_LC_HIP(b, b->flights.finish(slot, st, b->d_out, b->CAP, b->d_out_counts, S, b->d_err, b->d_maxt, nullptr, rows_cap, bd, nullptr, nullptr, b->d_alive));
  return MOT_OK;
}
// the oldest frame in flight: waits for it, books its maxima and event times; the caller takes the rows
static int bt_pop_flight(mot_bt_batch* b, mot::lifecycle::Flight** out, int* total) {
  if (b->flights.count <= 0) { b->ctx->err = "mot_bt_collect: no frame in flight"; return MOT_ERR_INVALID; }
  mot::lifecycle::Flight* F = nullptr;
  MOT_LC_HIP(b, b->flights.pop(&F));
  if (F->prof) { const int rce = bt_account_events(b, F->ev); if (rce != MOT_OK) return rce; }
  const int* maxt = b->flights.maxt_of(*F);
  b->bound_n = 0;
  for (int i = 0; i < 64; ++i) b->bound_n = (maxt[i] > b->bound_n) ? maxt[i] : b->bound_n;
  bt_set_hints(b, maxt);
  *total = F->h_meta[0];
  *out = F;
  b->d_rows_last = F->view ? F->h_rows : F->d_packed;  // mot_bt_device_output: the frame just collected
  b->d_offsets_last = F->d_offsets; b->d_counts_last = F->d_counts;
  if (F->h_meta[1]) { b->ctx->err = "mot_bt_collect: a stream exceeded cap_tracks / max_dets"; return MOT_ERR_CAPACITY; }
  if (*total > F->rows_cap) { b->ctx->err = "mot_bt_collect: more rows than rows_cap"; return MOT_ERR_CAPACITY; }  // (pack_rows skipped the streams that end past the ENQUEUE call's rows_cap)
  return MOT_OK;
}

int mot_bt_enqueue_packed(mot_bt_batch* b, const float* d_dets, const int* h_counts, int rows_cap) {
  if (!b || !d_dets || !h_counts || rows_cap <= 0) return MOT_ERR_INVALID;
  return bt_enqueue_flight(b, d_dets, h_counts, rows_cap, nullptr);
}

int mot_bt_collect_packed(mot_bt_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows) {
  if (!b || !out_counts) return MOT_ERR_INVALID;  // rows == NULL: the table stays on the device (mot_bt_device_output), only the counts come back
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = bt_pop_flight(b, &F, &total);
  if (F) std::memcpy(out_counts, b->flights.counts_of(*F), sizeof(int) * b->S);
  if (total_rows) *total_rows = total;
  if (rc != MOT_OK) return rc;
  if (!rows) return MOT_OK;
  if (total > rows_cap) { b->ctx->err = "mot_bt_collect_packed: more rows than rows_cap"; return MOT_ERR_CAPACITY; }
  MOT_LC_HIP(b, b->flights.copy_rows(*F, rows, total));
  return MOT_OK;
}

// ---- pooled form (round 4): the streams of the batch are independent tracker objects, a frame carries the ones that have work ----
int mot_bt_enqueue_frame(mot_bt_batch* b, const mot_frame_in* in, int rows_cap) {
  if (!b || !in || !in->d_dets || !in->h_counts || !in->h_det_ld || !in->h_det_off || rows_cap <= 0) return MOT_ERR_INVALID;
  return bt_enqueue_flight(b, in->d_dets, in->h_counts, rows_cap, in);
}
int mot_bt_collect_view(mot_bt_batch* b, mot_frame_view* out) {
  if (!b || !out) return MOT_ERR_INVALID;
  mot::lifecycle::Flight* F = nullptr;
  int total = 0;
  const int rc = bt_pop_flight(b, &F, &total);
  if (!F) return rc;
  if (!F->view) { b->ctx->err = "mot_bt_collect_view: the frame was queued with mot_bt_enqueue_packed"; return MOT_ERR_INVALID; }
  out->rows = F->h_rows; out->counts = b->flights.counts_of(*F); out->alive = b->flights.alive_of(*F, b->S); out->total = total;
  return rc;
}
int mot_bt_reset_stream(mot_bt_batch* b, int s, int fresh) {
  if (!b || s < 0 || s >= b->S) return MOT_ERR_INVALID;
  hipLaunchKernelGGL(mot::lifecycle::reset_stream_kernel<BtStream>, dim3(1), dim3(64), 0, b->ctx->stream, b->d_streams, s, b->h_streams[s], fresh ? 0 : 1, b->d_err);
  MOT_LC_HIP(b, hipGetLastError());
  return MOT_OK;
}
namespace {
__global__ void __launch_bounds__(256) bt_move(const BtStream* from, BtStream* to, const float* mean_from, float* mean_to, const float* dense_from, float* dense_to, const float* blk_from, float* blk_to, const unsigned char* flag_from, unsigned char* flag_to, int cap) {
  using mot::lifecycle::move_array;
  const BtStream& A = *from;
  BtStream& B = *to;
  const size_t n = static_cast<size_t>(cap);
  move_array(B.free_stack, A.free_stack, n);
  move_array(B.active[0], A.active[A.cur], n); move_array(B.lost[0], A.lost[A.cur], n);
  move_array(B.t_id, A.t_id, n); move_array(B.t_state, A.t_state, n); move_array(B.t_act, A.t_act, n); move_array(B.t_tlen, A.t_tlen, n);
  move_array(B.t_fid, A.t_fid, n); move_array(B.t_sf, A.t_sf, n); move_array(B.t_cls, A.t_cls, n); move_array(B.t_det, A.t_det, n);
  move_array(B.t_conf, A.t_conf, n);
  move_array(mean_to, mean_from, n * 64);
  move_array(dense_to, dense_from, n * 8);
  move_array(blk_to, blk_from, n * 16);
  move_array(flag_to, flag_from, n);
  __syncthreads();
  if (threadIdx.x == 0) {
    B.frame_count = A.frame_count; B.next_id = A.next_id; B.next_slot = A.next_slot; B.n_free = A.n_free;
    B.n_active = A.n_active; B.n_lost = A.n_lost; B.err = A.err; B.cur = 0; B.skip = 1;
  }
}
}  // namespace
// moves stream s of `src` into stream s2 of `dst` (a batch with the same parameters and capacities at least as large, on the same
// device): lists, per-track records and Kalman states; the slot indices stay valid, fresh slots continue behind src's. Synchronous.
int mot_bt_move_stream(mot_bt_batch* src, int s, mot_bt_batch* dst, int s2) {
  if (!src || !dst || s < 0 || s >= src->S || s2 < 0 || s2 >= dst->S || dst->CAP < src->CAP || dst->D < src->D) return MOT_ERR_INVALID;
  MOT_LC_HIP(src, hipStreamSynchronize(src->ctx->stream));
  hipStream_t st = dst->ctx->stream;
  hipLaunchKernelGGL(bt_move, dim3(1), dim3(256), 0, st, src->d_streams + s, dst->d_streams + s2, src->mean + static_cast<size_t>(s) * 64 * src->CAP,
                     dst->mean + static_cast<size_t>(s2) * 64 * dst->CAP, src->mean_dense + static_cast<size_t>(s) * 8 * src->CAP,
                     dst->mean_dense + static_cast<size_t>(s2) * 8 * dst->CAP, src->cov_blocks + static_cast<size_t>(s) * 16 * src->CAP,
                     dst->cov_blocks + static_cast<size_t>(s2) * 16 * dst->CAP, src->dense_flag + static_cast<size_t>(s) * src->CAP,
                     dst->dense_flag + static_cast<size_t>(s2) * dst->CAP, src->CAP);
  MOT_LC_HIP(dst, hipGetLastError());
  BtStream h;
  MOT_LC_HIP(dst, hipMemcpyAsync(&h, dst->d_streams + s2, sizeof(BtStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(dst, hipStreamSynchronize(st));
  const int alive = h.n_active + h.n_lost;
  if (alive > dst->bound_n) dst->bound_n = alive;  // the next frame's launches cover the newcomer's lists
  return MOT_OK;
}

int mot_bt_profile(mot_bt_batch* b, int enable) {
  b->profile = enable != 0;
  if (enable) {
    b->lap_ms[0] = b->lap_ms[1] = b->frame_ms = 0.0;
    b->lap_sparse1_ms = 0.0;
    b->kf_ms[0] = b->kf_ms[1] = b->kf_ms[2] = 0.0;
    b->frames = 0;
    MOT_LC_HIP(b, hipMemsetAsync(b->d_stats, 0, 8 * 64 * sizeof(unsigned long long), b->ctx->stream));
    MOT_LC_HIP(b, hipStreamSynchronize(b->ctx->stream));
  }
  return MOT_OK;
}

int mot_bt_profile_stats(mot_bt_batch* b, double* out8) {
  unsigned long long raw[8 * 64];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long h[4] = {0, 0, 0, 0};
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 4; ++k) h[k] += raw[i * 8 + k];
  out8[0] = b->lap_ms[0]; out8[1] = b->lap_ms[1]; out8[2] = b->frame_ms; out8[3] = static_cast<double>(b->frames);
  out8[4] = static_cast<double>(h[0]); out8[5] = static_cast<double>(h[1]); out8[6] = static_cast<double>(h[2]); out8[7] = static_cast<double>(h[3]);
  return MOT_OK;
}

int mot_bt_profile_lap_sparse(mot_bt_batch* b, double* out2) {  // HIP-event ms of the first association's sparse-solver kernel alone, launches
  out2[0] = b->lap_sparse1_ms; out2[1] = static_cast<double>(b->frames);
  return MOT_OK;
}

int mot_bt_profile_dims(mot_bt_batch* b, double* out4) {
  unsigned long long raw[8 * 64];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 8; ++k) h[k] += raw[i * 8 + k];
  out4[0] = static_cast<double>(h[4]); out4[1] = static_cast<double>(h[1] - h[4]);
  out4[2] = static_cast<double>(h[5]); out4[3] = static_cast<double>(h[3] - h[5]);
  return MOT_OK;
}

int mot_bt_profile_kalman(mot_bt_batch* b, double* out6) {
  unsigned long long raw[8 * 64];
  MOT_LC_HIP(b, hipMemcpy(raw, b->d_stats, sizeof(raw), hipMemcpyDeviceToHost));
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 8; ++k) h[k] += raw[i * 8 + k];
  out6[0] = b->kf_ms[0]; out6[1] = static_cast<double>(h[4]);  // predicted boxes: pool tracks of the first association
  out6[2] = b->kf_ms[1]; out6[3] = static_cast<double>(h[7]);  // initiations
  out6[4] = b->kf_ms[2]; out6[5] = static_cast<double>(h[6]);  // updates
  return MOT_OK;
}

int mot_bt_dump(mot_bt_batch* b, int s, int* ids, float* mean, float* cov, int cap) {
  hipStream_t st = b->ctx->stream;
  BtStream h;
  MOT_LC_HIP(b, hipMemcpyAsync(&h, b->d_streams + s, sizeof(BtStream), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  const int n = h.n_active + h.n_lost;
  if (n > cap) return -n;
  std::vector<int> slots(n), tid(b->CAP);
  if (h.n_active) MOT_LC_HIP(b, hipMemcpyAsync(slots.data(), h.active[h.cur], sizeof(int) * h.n_active, hipMemcpyDeviceToHost, st));
  if (h.n_lost) MOT_LC_HIP(b, hipMemcpyAsync(slots.data() + h.n_active, h.lost[h.cur], sizeof(int) * h.n_lost, hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(tid.data(), h.t_id, sizeof(int) * b->CAP, hipMemcpyDeviceToHost, st));
  const int C2 = b->CAP;
  std::vector<float> m(static_cast<size_t>(64) * C2);
  MOT_LC_HIP(b, hipMemcpyAsync(m.data(), b->mean + static_cast<size_t>(s) * 64 * C2, sizeof(float) * m.size(), hipMemcpyDeviceToHost, st));
  std::vector<float> md(static_cast<size_t>(8) * C2);  // (the means live in the dense array, BtStream::kdense)
  MOT_LC_HIP(b, hipMemcpyAsync(md.data(), b->mean_dense + static_cast<size_t>(s) * 8 * C2, sizeof(float) * md.size(), hipMemcpyDeviceToHost, st));
  std::vector<float> mb(static_cast<size_t>(16) * C2);
  std::vector<unsigned char> mf(C2);
  MOT_LC_HIP(b, hipMemcpyAsync(mb.data(), b->cov_blocks + static_cast<size_t>(s) * 16 * C2, sizeof(float) * mb.size(), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipMemcpyAsync(mf.data(), b->dense_flag + static_cast<size_t>(s) * C2, mf.size(), hipMemcpyDeviceToHost, st));
  MOT_LC_HIP(b, hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {
    const int sl = slots[i];
    ids[i] = tid[sl];
    for (int k = 0; k < 8; ++k) mean[static_cast<size_t>(i) * 8 + k] = md[static_cast<size_t>(sl) * 8 + k];
    float* C = cov + static_cast<size_t>(i) * 64;
    if (mf[sl]) { for (int k = 0; k < 64; ++k) C[k] = m[static_cast<size_t>(sl) * 64 + k]; }
    else {  // block form -> 8 x 8
      for (int k = 0; k < 64; ++k) C[k] = 0.0f;
      for (int c = 0; c < 4; ++c) {
        const float* q = &mb[static_cast<size_t>(sl) * 16 + 4 * c];
        C[c * 8 + c] = q[0]; C[c * 8 + c + 4] = q[1]; C[(c + 4) * 8 + c] = q[2]; C[(c + 4) * 8 + c + 4] = q[3];
      }
    }
  }
  return n;
}

}  // extern "C"
