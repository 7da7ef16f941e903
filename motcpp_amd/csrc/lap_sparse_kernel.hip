// Fast path of the linear assignment on gfx950: lap_sparse.hpp (viable pairs only, shortest augmenting paths, uniqueness
// certificate), ONE WAVEFRONT per problem. It runs in front of lap_kernel (the exact lapjv emulation) over the same task
// array: a problem whose optimum it certifies as unique is finished here and its status word (last 16 bytes of the task's
// scratch) says so; every other problem is left untouched for lap_kernel, which skips the finished ones. Solver state
// (duals, assignments, search slots, x1 buckets: 16 B per row + 12 B per column + 4.4 KB) sits in LDS when it fits, the
// viable-pair lists and the bucket-ordered row boxes in the task's global scratch.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"
#include "lap_cost.hpp"
#include "lap_sparse.hpp"

namespace {

constexpr int kScratch = 1024;  // DevGroup reduction scratch

__device__ __forceinline__ int* status_word(const mot_lap_task& T, size_t scratch_bytes) {
  return reinterpret_cast<int*>(static_cast<char*>(T.work) + scratch_bytes - 16);
}

template <bool PLAIN, int HS>
__global__ void __launch_bounds__(64) lap_sparse_kernel(const mot_lap_task* __restrict__ tasks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mot_lap_task T = tasks[blockIdx.x];
  const int nr = T.n, nc = T.m, t = threadIdx.x;
  int* status = status_word(T, mot::lap_task_scratch_bytes(nr, nc));
  if (nr <= 0 || nc <= 0) {
    for (int i = t; i < nr; i += 64) { T.x[i] = -1; if (T.xval) T.xval[i] = 0.f; }
    for (int j = t; j < nc; j += 64) T.y[j] = -1;
    if (t == 0) { if (T.info) T.info[0] = 2; *status = 1; }
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
  if (skip) { if (t == 0) *status = 0; return; }

  mot::DevGroup g(smem);
  mot::SparseWorkT<HS> w;
  char* cold = static_cast<char*>(T.work);
  const size_t cold_b = (mot::sparse_cold_bytes(nr, nc) + 15) & ~size_t(15);
  mot::sparse_carve_cold(w, cold, nr, nc);
  if constexpr (HS == mot::kMemLds) mot::sparse_carve_hot(w, smem + kScratch, nr, nc);
  else mot::sparse_carve_hot(w, cold + cold_b, nr, nc);

  int path = 0;
  if (T.mode == MOT_LAP_OCSORT) {
    // a = (iou > gate); trivial one-to-one case iff max row sum == 1 and max col sum == 1 (ocsort.cpp:684-696)
    int max_row = 0, max_col = 0;
    for (int i = t; i < nr; i += 64) { w.x[i] = -1; w.slot[i] = 0; }
    g.sync();
    for (int j = t; j < nc; j += 64) {
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
    for (int i = t; i < nr; i += 64) {
      const int c = w.slot[i];
      if (c != 1) w.x[i] = -1;
      if (c > max_row) max_row = c;
    }
    max_row = g.reduce_max(max_row);
    max_col = g.reduce_max(max_col);
    g.sync();
    if (max_row == 1 && max_col == 1) path = 1;
  }
  int solved = 1;
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
      e = mot::sparse_enumerate_boxes(g, w, nr, nc, mot::SparseBoxes{G.a, G.lda, G.aidx}, mot::SparseBoxes{G.b, G.ldb, G.bidx},
                                      G.bconf, G.bidx, T.thresh, eval, zc);
    } else {
      e = mot::sparse_enumerate_matrix(g, w, nr, nc, T.cost, T.ldc, T.thresh);
    }
    if (!e.ok) solved = 0;
    else if (T.mode == MOT_LAP_GATE_MIN && !(e.mincost < static_cast<double>(T.gate))) path = 2;
    else solved = (mot::sparse_solve(g, w, nr, nc, T.thresh) == 1) ? 1 : 0;
  }
  if (!solved) { if (t == 0) *status = 0; return; }
  if (path == 2) {
    for (int i = t; i < nr; i += 64) w.x[i] = -1;
    for (int j = t; j < nc; j += 64) w.y[j] = -1;
  }
  g.sync();
  for (int i = t; i < nr; i += 64) {
    const int xi = w.x[i];
    T.x[i] = xi;
    if (T.xval) {
      float v = 0.f;
      if (xi >= 0) {
        if (T.iou) v = mot::gld(T.iou, static_cast<size_t>(i) * T.ldi + xi);
        else if (!geom) v = mot::gld(T.cost, static_cast<size_t>(i) * T.ldc + xi);
        else {  // the pair's cost as enumerated (bit-identical to the cost kernel's value)
          for (int k = 0; k < mot::kSpK; ++k) {
            const int r = w.erow[static_cast<size_t>(xi) * mot::kSpK + k];
            if (r < 0) break;
            if (r == i) { v = w.ecost[static_cast<size_t>(xi) * mot::kSpK + k]; break; }
          }
        }
      }
      T.xval[i] = v;
    }
  }
  for (int j = t; j < nc; j += 64) T.y[j] = w.y[j];
  if (t == 0) { if (T.info) T.info[0] = path; *status = 1; }
}

}  // namespace

namespace mot {
// Launches the fast path over the task array. Hot state in LDS when it fits next to 7 other problems' (<= 20 KB) or at
// least alone in 60 KB; else in the task's global scratch.
hipError_t launch_lap_sparse(const mot_lap_task* tasks, int ntasks, int max_n, int max_m, bool plain_costs, hipStream_t st) {
  if (ntasks <= 0) return hipSuccess;
  const size_t hot = kScratch + sparse_hot_bytes(max_n > 0 ? max_n : 1, max_m > 0 ? max_m : 1) + 16;
  const bool lds = hot <= 60 * 1024;
  if (lds) {
    if (plain_costs) hipLaunchKernelGGL((lap_sparse_kernel<true, kMemLds>), dim3(ntasks), dim3(64), hot, st, tasks);
    else hipLaunchKernelGGL((lap_sparse_kernel<false, kMemLds>), dim3(ntasks), dim3(64), hot, st, tasks);
  } else {
    if (plain_costs) hipLaunchKernelGGL((lap_sparse_kernel<true, kMemGlobal>), dim3(ntasks), dim3(64), kScratch, st, tasks);
    else hipLaunchKernelGGL((lap_sparse_kernel<false, kMemGlobal>), dim3(ntasks), dim3(64), kScratch, st, tasks);
  }
  return hipGetLastError();
}
}  // namespace mot
