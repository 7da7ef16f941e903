// Fast path of the linear assignment on gfx950: lap_sparse.hpp (viable pairs only, shortest augmenting paths, uniqueness
// certificate), ONE WAVEFRONT per problem. It runs in front of lap_kernel (the exact lapjv emulation) over the same task
// array: a problem whose optimum it certifies as unique is finished here and its status word (last 16 bytes of the task's
// scratch) says so; every other problem is left untouched for lap_kernel, which skips the finished ones. Solver state
// (duals, assignments, search slots, x1 buckets: 16 B per row + 12 B per column + 4.4 KB) sits in LDS when it fits, the
// viable-pair lists and the bucket-ordered row boxes in the task's global scratch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/motcpp_amd.h"
#include "lap_cost.hpp"
#include "lap_sparse.hpp"

namespace {

constexpr int kScratch = 1024;  // DevGroup reduction scratch

// Outcome histogram of the fast path since the last reset (diagnostics, one atomic per problem): [0] finished here,
// [1] enumeration declined (tie with the threshold, NaN/inf, list overflow, viable non-intersecting pairs), [2] a path search
// reached too many rows, [3] certificate arithmetic, [4] too many tight pairs, [5] optimum not unique, [6] not attempted
// (diagnostics requested / cost flavour), [7] empty problems
__device__ unsigned long long g_fast_hist[64 * 32];  // 64 sets (block & 63) so that thousands of problems do not serialise on one line;  // [8..] cycles: enumeration, matching init, path searches, certificate; [12] searches, [13] column scans
__device__ int g_sp_hist_wide_only;  // diagnostics: 1 = only the four-wavefront kernel feeds the cycle counters (a frame's first association alone)
__device__ unsigned long long g_sp_timeline[8192][2];  // diagnostics: residence (100 MHz clock) of the first 8192 workgroups of the last launch
__device__ __forceinline__ unsigned long long* hist_set() { return g_fast_hist + (blockIdx.x & 63) * 32; }
__device__ __forceinline__ void count_outcome(int k) { if (threadIdx.x == 0) atomicAdd(hist_set() + k, 1ull); }

__device__ __forceinline__ int* status_word(const mot_lap_task& T, size_t scratch_bytes) {
  return reinterpret_cast<int*>(static_cast<char*>(T.work) + scratch_bytes - 16);
}

// kThreads = 64: one wavefront does everything. kThreads = 256: four wavefronts enumerate the viable pairs, build the initial
// matching and check the certificate together (those loops run over rows / columns / pairs); the serial path searches in
// between are done by the first wavefront alone while the others wait at the next barrier.
template <bool PLAIN, int HS, int kThreads>
__global__ void __launch_bounds__(kThreads) lap_sparse_kernel(const mot_lap_task* __restrict__ tasks, int lds_ecap, int* declined, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const long long kc0 = MOT_CLOCK();           // (diagnostics: the workgroup's whole residence, in shader cycles and in the constant 100 MHz clock —
  const long long kw0 = wall_clock64();        //  their ratio is the clock the kernel really ran at, their sum over a launch the mean residency)
  const mot_lap_task T = tasks[blockIdx.x];
  const int nr = T.n, nc = T.m, t = threadIdx.x;
  int* status = status_word(T, mot::lap_task_scratch_bytes(nr, nc));
  if (nr <= 0 || nc <= 0) {
    for (int i = t; i < nr; i += kThreads) { T.x[i] = -1; if (T.xval) T.xval[i] = 0.f; }
    for (int j = t; j < nc; j += kThreads) T.y[j] = -1;
    if (t == 0) { if (T.info) T.info[0] = 2; *status = 1; }
    count_outcome(7);
    return;
  }
  const bool geom = T.geom.a != nullptr;
  // left to the exact path: diagnostics requested, the gated appearance cost in a launch compiled without it, or a
  // proximity gate that lets pairs that do not intersect through (their cost then depends on the row)
  bool skip = T.prof != nullptr;
  if (geom && T.geom.mode == MOT_COST_BOTSORT) {
    if (PLAIN) skip = true;
    else if (!(1.0f > T.geom.prox_thresh)) skip = true;
  }
  if (geom && T.geom.mode == MOT_COST_FUSE_IOU) skip = true;  // cost depends on a per-pair ReID term whatever the overlap
  // the launch's LDS was sized from a hint of the problem sizes (tighter than the hard bounds): a problem that does not fit is left
  // to the exact path
  if (HS == mot::kMemLds && static_cast<size_t>(kScratch) + mot::sparse_hot_bytes(nr, nc, lds_ecap) > static_cast<size_t>(lds_bytes)) skip = true;
  if (skip) { if (t == 0) { *status = 0; atomicAdd(declined, 1); } count_outcome(6); return; }

  mot::DevGroup g(smem);
  mot::SparseWorkT<HS> w;
  char* cold = static_cast<char*>(T.work);
  const size_t cold_b = (mot::sparse_cold_bytes(nr, nc) + 15) & ~size_t(15);
  mot::sparse_carve_cold(w, cold, nr, nc);
  if constexpr (HS == mot::kMemLds) mot::sparse_carve_hot(w, smem + kScratch, nr, nc, lds_ecap);
  else mot::sparse_carve_hot(w, cold + cold_b, nr, nc, mot::sparse_global_ecap(nc));

  int path = 0;
  if (T.mode == MOT_LAP_OCSORT) {
    // a = (iou > gate); trivial one-to-one case iff max row sum == 1 and max col sum == 1 (ocsort.cpp:684-696)
    int max_row = 0, max_col = 0;
    for (int i = t; i < nr; i += kThreads) { w.x[i] = -1; w.slot[i] = 0; }
    g.sync();
    for (int j = t; j < nc; j += kThreads) {
      int c = 0, last = -1;
      for (int i = 0; i < nr; ++i)
        if (mot::gld(T.iou, static_cast<size_t>(i) * T.ldi + j) > T.gate) {
          ++c; last = i;
          mot::DevGroup::atomic_add(w.slot.raw(i), 1);
          mot::DevGroup::atomic_max(w.x.raw(i), j);
        }
      w.y[j] = (c == 1) ? last : -1;
      if (c > max_col) max_col = c;
    }
    g.sync();
    for (int i = t; i < nr; i += kThreads) {
      const int c = w.slot[i];
      if (c != 1) w.x[i] = -1;
      if (c > max_row) max_row = c;
    }
    max_row = g.reduce_max(max_row);
    max_col = g.reduce_max(max_col);
    g.sync();
    if (max_row == 1 && max_col == 1) path = 1;
  }
  int solved = 1, reason = 0;
  const long long ck0 = MOT_CLOCK();
  if (path == 0) {
    mot::SparseEnum e;
    if (geom) {
      const mot_iou_task& G = T.geom;
      using Cost = mot::IouCostT<0, mot::kMemGlobal, false, PLAIN>;
      Cost C;
      C.prm = mot::CostParams{G.mode, G.prox_thresh, G.app_thresh, G.fuse, G.emb != nullptr, G.emb == nullptr && G.lde < 0, MOT_ASSOC_IOU, 1.0f};
      C.emb = G.emb;
      C.lde = G.lde;
      C.conf = nullptr;
      auto eval = [&](int i, const float ra[4], float raa, const float cb[4], float cba, float cf, int j) {
        typename Cost::Row r;
        r.i = i;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.a[k] = ra[k];
        r.area = raa;
        return C.eval_f(r, cb, cba, cf, j);
      };
      auto zc = [&](float cf) { return mot::cost_from_iou<!PLAIN>(C.prm, 0.0f, cf, []() { return 0.0f; }); };
      // IoU a pair of a column with confidence cf needs before it can be viable (costs fall with the IoU in these modes),
      // 1 % taken off; 0 = no bound (then every intersecting pair is evaluated)
      const bool bound_ok = T.mode == MOT_LAP_PLAIN;
      const int cmode = G.mode, cfuse = G.fuse;
      const float th_f = T.thresh, prox = G.prox_thresh;
      auto min_iou = [=](float cf) {
        if (!bound_ok) return 0.0f;
        float need = 0.0f;
        if (cmode == MOT_COST_IOU_DIST) need = 1.0f - th_f;
        else if (cmode == MOT_COST_NEG_IOU) need = -th_f;
        else if (cmode == MOT_COST_IOU_DIST_FUSE) need = (cf > 0.0f) ? (1.0f - th_f) / cf : 2.0f;  // conf <= 0: never below 1
        else if (cmode == MOT_COST_BOTSORT) {
          const float plain = cfuse ? ((cf > 0.0f) ? (1.0f - th_f) / cf : 2.0f) : 1.0f - th_f;
          need = mot::smin(plain, 1.0f - prox);  // the appearance term needs 1 - iou <= prox
        }
        if (!(need > 1.0e-3f)) return 0.0f;
        return (need < 1.0f) ? 0.99f * need : 0.99f;
      };
      e = mot::sparse_enumerate_boxes<(kThreads == 1024) ? 2 : ((kThreads == 256) ? 6 : mot::kSpRC), (kThreads == 1024) ? 2 : mot::kSpQ>(g, w, nr, nc, mot::SparseBoxes{G.a, G.lda, G.aidx}, mot::SparseBoxes{G.b, G.ldb, G.bidx},
                                      G.bconf, G.bidx, T.thresh, eval, zc, min_iou);
    } else {
      e = mot::sparse_enumerate_matrix(g, w, nr, nc, T.cost, T.ldc, T.thresh);
    }
    if (e.ok <= 0) { solved = 0; reason = (e.ok == -10) ? 14 : ((e.ok == -11) ? 15 : ((e.ok == -12) ? 21 : ((e.ok == -13) ? 22 : ((e.ok == -14) ? 23 : 1)))); }
    else if (T.mode == MOT_LAP_GATE_MIN && !(e.mincost < static_cast<double>(T.gate))) path = 2;
    else {
      const long long ck1 = MOT_CLOCK();
      mot::SparseProf pf;
      int r;
      if constexpr (kThreads == 64) r = mot::sparse_solve(g, w, nr, nc, T.thresh, &pf);
      else {
        const int nfree0 = mot::sparse_init(g, w, nr, nc, T.thresh);
        const int nfree = mot::sparse_short_searches(g, w, nfree0, T.thresh);
        if (t == 0) atomicAdd(hist_set() + 28, static_cast<unsigned long long>(nfree0 - nfree) << 32);  // (high half of [28]: searches finished by the lanes' short form)
        const long long ck2 = MOT_CLOCK();
        // the path searches: every wavefront takes every fourth free column (lap_sparse.hpp, SHARED); the few that ran into each other's
        // rows are then redone by the first wavefront alone
        static_assert(kThreads == 256 || kThreads == 1024, "one search per wavefront, four at a time");
        if (t < 256) {
          mot::DevWave gw;
          int scans = 0;
          long long seg[4] = {0, 0, 0, 0};
          const int rs = mot::sparse_search_impl<true>(gw, w, nfree, T.thresh, t >> 6, 4, false, &scans, seg);
          if (rs != 1 && (t & 63) == 0) w.ctr.atomic_min(3, rs);
          if (t == 0) {
            unsigned long long* h = hist_set();  // where the searches spend their cycles (first wavefront)
            atomicAdd(h + 24, static_cast<unsigned long long>(seg[0])); atomicAdd(h + 25, static_cast<unsigned long long>(seg[1]));
            atomicAdd(h + 26, static_cast<unsigned long long>(seg[2])); atomicAdd(h + 27, static_cast<unsigned long long>(seg[3]));
          }
          pf.n_scan = scans;
        }
        g.sync();
        const int nretry = w.ctr[1];
        if (nretry > 0 && static_cast<int>(w.ctr[3]) == 1 && t < 64) {
          mot::DevWave gw;
          const int rs = mot::sparse_search_impl<false>(gw, w, nretry, T.thresh, 0, 1, true, nullptr, nullptr);
          if (t == 0) { w.ctr[3] = rs; atomicAdd(hist_set() + 28, static_cast<unsigned long long>(nretry)); }
        }
        if (nretry > 0) g.sync();
        const long long ck3 = MOT_CLOCK();
        r = w.ctr[3];
        g.sync();
        if (r == 1) r = mot::sparse_certify(g, w, nr, nc, T.thresh);
        pf.c_init = ck2 - ck1; pf.c_search = ck3 - ck2; pf.c_cert = MOT_CLOCK() - ck3; pf.n_search = nfree;
      }
      solved = (r == 1) ? 1 : 0;
      reason = (r >= -5 && r <= -2) ? -r : 1;
      if (t == 0 && !(g_sp_hist_wide_only && kThreads == 64)) {
        unsigned long long* h = hist_set();
        atomicAdd(h + 8, static_cast<unsigned long long>(ck1 - ck0)); atomicAdd(h + 9, static_cast<unsigned long long>(pf.c_init));
        atomicAdd(h + 10, static_cast<unsigned long long>(pf.c_search)); atomicAdd(h + 11, static_cast<unsigned long long>(pf.c_cert));
        atomicAdd(h + 12, static_cast<unsigned long long>(pf.n_search)); atomicAdd(h + 13, static_cast<unsigned long long>(pf.n_scan));
        atomicAdd(h + 16, static_cast<unsigned long long>(e.c_stage)); atomicAdd(h + 17, static_cast<unsigned long long>(e.c_cand));
        atomicAdd(h + 18, static_cast<unsigned long long>(e.c_csr)); atomicAdd(h + 19, static_cast<unsigned long long>(e.n_cand));
        atomicAdd(h + 20, static_cast<unsigned long long>(e.n_hit));
      }
    }
  }
  if (!solved) { if (t == 0) { *status = 0; atomicAdd(declined, 1); } count_outcome(reason); return; }
  if (path == 2) {
    for (int i = t; i < nr; i += kThreads) w.x[i] = -1;
    for (int j = t; j < nc; j += kThreads) w.y[j] = -1;
  }
  g.sync();
  for (int i = t; i < nr; i += kThreads) {
    const int xi = w.x[i];
    T.x[i] = xi;
    if (T.xval) {
      float v = 0.f;
      if (xi >= 0) {
        if (T.iou) v = mot::gld(T.iou, static_cast<size_t>(i) * T.ldi + xi);
        else if (!geom) v = mot::gld(T.cost, static_cast<size_t>(i) * T.ldc + xi);
        else {  // the pair's cost as enumerated (bit-identical to the cost kernel's value)
          const int ee = w.eoff[xi], e0 = mot::sp_e0(ee);
          for (int k = e0; k < e0 + mot::sp_deg(ee); ++k)
            if (static_cast<int>(w.erow[k]) == i) { v = w.ecost[k]; break; }
        }
      }
      T.xval[i] = v;
    }
  }
  for (int j = t; j < nc; j += kThreads) T.y[j] = w.y[j];
  if (t == 0) { if (T.info) T.info[0] = path; *status = 1; }
  count_outcome(0);
  if (t == 0 && !(g_sp_hist_wide_only && kThreads == 64)) {
    unsigned long long* h = hist_set();
    atomicAdd(h + 29, static_cast<unsigned long long>(MOT_CLOCK() - kc0)); atomicAdd(h + 30, static_cast<unsigned long long>(wall_clock64() - kw0));
    atomicAdd(h + 31, static_cast<unsigned long long>(ck0 - kc0));
    if (kThreads == 256 && blockIdx.x < 8192) { g_sp_timeline[blockIdx.x][0] = static_cast<unsigned long long>(kw0); g_sp_timeline[blockIdx.x][1] = static_cast<unsigned long long>(wall_clock64()); }
  }
}

}  // namespace

namespace mot {
// Launches the fast path over the task array. Hot state in LDS when it fits in 64 KB (the CSR list's capacity gives way first:
// 5 pairs per column by default, never fewer than 3), else in the task's global scratch.
hipError_t launch_lap_sparse(const mot_lap_task* tasks, int ntasks, int max_n, int max_m, bool plain_costs, int* declined, hipStream_t st,
                             int hint_n, int hint_m, int active_tasks) {
  if (ntasks <= 0) return hipSuccess;
  // LDS per problem (which decides how many problems a CU holds) from the caller's hint of the sizes when it has one
  int n = max_n > 0 ? max_n : 1, m = max_m > 0 ? max_m : 1;
  if (hint_n > 0 && hint_n < n) n = hint_n;
  if (hint_m > 0 && hint_m < m) m = hint_m;
  const bool active_few = ((active_tasks > 0 && active_tasks < ntasks) ? active_tasks : ntasks) <= 32;
  constexpr size_t kBudget = 64 * 1024 - 64;
  int ecap = sparse_default_ecap(m);
  size_t hot = kScratch + sparse_hot_bytes(n, m, ecap);
  if (hot > kBudget) {
    const size_t fixed = kScratch + sparse_hot_bytes(n, m, 0);
    ecap = (fixed < kBudget) ? static_cast<int>((kBudget - fixed) / 8) : 0;
    hot = kScratch + sparse_hot_bytes(n, m, ecap);
  }
  const bool lds = ecap >= 3 * m + 16;
  // LDS per problem decides how many problems a CU holds (160 KB: 5 at 32 KB, 4 at 40 KB, 3 at 53 KB): the pair list takes what is left of
  // the step the launch lands on anyway — a problem whose list overflows goes to the exact solver, milliseconds instead of microseconds
  if (lds) {
    // (a handful of problems: nobody shares the CU — the list gets the whole budget)
    for (int k = (active_few ? 2 : 32); k >= 1; --k) {  // the finest step the launch already fits: k problems per CU
      const size_t cap = ((static_cast<size_t>(160) * 1024 / k) < kBudget ? (static_cast<size_t>(160) * 1024 / k) : kBudget) & ~size_t(63);
      if (hot <= cap) {
        const int more = static_cast<int>((cap - hot) / 6) - 4;  // (6 bytes per entry; the two arrays round up to 16 bytes each)
        if (more > 0) { ecap += more; hot = kScratch + sparse_hot_bytes(n, m, ecap); }
        break;
      }
    }
  }
  // four wavefronts per problem once there is enough to enumerate (the pairs are listed four times faster; the hot state's
  // LDS, which bounds the problems resident per CU, is the same)
  // a handful of problems (one camera): the launch's latency is the frame's — four wavefronts from small problems on, sixteen for the
  // large ones (one column per lane in the enumeration; the searches still run on four)
  const bool few = active_few;
  const bool wide = lds && (static_cast<long>(n) * m >= (few ? 4 * 1024 : 64 * 1024));
  static const bool allow16 = std::getenv("MOT_LAP_SPARSE_NO16") == nullptr;
  const bool wide16 = wide && few && allow16 && (static_cast<long>(n) * m >= 128 * 1024);
#define MOT_SP_LAUNCH(P, H, TH, BYTES) hipLaunchKernelGGL((lap_sparse_kernel<P, H, TH>), dim3(ntasks), dim3(TH), BYTES, st, tasks, ecap, declined, static_cast<int>(BYTES))
  if (lds) {
    if (wide16) { if (plain_costs) MOT_SP_LAUNCH(true, kMemLds, 1024, hot); else MOT_SP_LAUNCH(false, kMemLds, 1024, hot); }
    else if (wide) { if (plain_costs) MOT_SP_LAUNCH(true, kMemLds, 256, hot); else MOT_SP_LAUNCH(false, kMemLds, 256, hot); }
    else { if (plain_costs) MOT_SP_LAUNCH(true, kMemLds, 64, hot); else MOT_SP_LAUNCH(false, kMemLds, 64, hot); }
  } else {
    ecap = 0;
    if (plain_costs) MOT_SP_LAUNCH(true, kMemGlobal, 64, kScratch); else MOT_SP_LAUNCH(false, kMemGlobal, 64, kScratch);
  }
#undef MOT_SP_LAUNCH
  return hipGetLastError();
}
hipError_t lap_sparse_timeline(unsigned long long* out, int n, hipStream_t st) {
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (n < 0) { const int v = (n == -1) ? 1 : 0; return hipMemcpyToSymbol(HIP_SYMBOL(g_sp_hist_wide_only), &v, sizeof(int)); }  // (-1 / -2: cycle counters from the wide kernel only / from all)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_timeline), sizeof(unsigned long long) * 2 * static_cast<size_t>(n < 8192 ? n : 8192));
}
hipError_t lap_fast_stats(unsigned long long* out16 /* [32] */, bool reset, hipStream_t st) {
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (out16) {
    unsigned long long raw[64 * 32];
    e = hipMemcpyFromSymbol(raw, HIP_SYMBOL(g_fast_hist), sizeof(raw));
    if (e != hipSuccess) return e;
    for (int k = 0; k < 32; ++k) { out16[k] = 0; for (int s = 0; s < 64; ++s) out16[k] += raw[s * 32 + k]; }
  }
  if (reset) { static const unsigned long long z[64 * 32] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_fast_hist), z, sizeof(z)); }
  return e;
}
}  // namespace mot
