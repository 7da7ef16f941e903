// Linear-assignment kernel for gfx950: one 256-lane workgroup (4 wavefronts) per problem runs the
// exact lapjv restatement of lap_core.hpp. All solver state (duals, assignments, free list, path
// arrays: 44 bytes per extended row) lives in LDS when n+m <= mot_lap_lds_limit() — the cost matrix
// is the only thing read from HBM/L2, rows by coalesced loads — otherwise in a caller-provided
// global scratch. Throughput comes from many problems in flight (grid = problems: streams x stages),
// not from one problem; a problem is latency/dependency-bound by construction (sequential rows).
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"
#include "lap_core.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kScratch = 1024;
constexpr int kLdsBudget = 160 * 1024;

template <bool kLds>
__global__ void __launch_bounds__(kThreads) lap_kernel(const mot_lap_task* __restrict__ tasks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mot_lap_task T = tasks[blockIdx.x];
  const int nr = T.n, nc = T.m, n = nr + nc;
  const int t = threadIdx.x;
  if (nr <= 0 || nc <= 0) {
    for (int i = t; i < nr; i += kThreads) { T.x[i] = -1; if (T.xval) T.xval[i] = 0.f; }
    for (int j = t; j < nc; j += kThreads) T.y[j] = -1;
    if (T.info && t == 0) T.info[0] = 2;
    return;
  }
  mot::DevGroup g(smem);
  void* base = kLds ? static_cast<void*>(smem + kScratch) : T.work;
  const mot::LapWork W = mot::lap_carve(base, n);
  int path = 0;

  if (T.mode == MOT_LAP_GATE_MIN) {
    double mn = 1e300;
    for (int e = t; e < nr * nc; e += kThreads) {
      const double c = static_cast<double>(T.cost[static_cast<size_t>(e / nc) * T.ldc + (e % nc)]);
      if (c < mn) mn = c;
    }
    mn = g.reduce_min(mn);
    if (!(mn < static_cast<double>(T.gate))) path = 2;
  } else if (T.mode == MOT_LAP_OCSORT) {
    // a = (iou > gate); trivial one-to-one case iff max row sum == 1 and max col sum == 1 (ocsort.cpp:684-696)
    int max_row = 0, max_col = 0;
    for (int i = t; i < nr; i += kThreads) {
      int c = 0, last = -1;
      const float* r = T.iou + static_cast<size_t>(i) * T.ldi;
      for (int j = 0; j < nc; ++j)
        if (r[j] > T.gate) { ++c; last = j; }
      W.x[i] = (c == 1) ? last : -1;
      if (c > max_row) max_row = c;
    }
    for (int j = t; j < nc; j += kThreads) {
      int c = 0, last = -1;
      for (int i = 0; i < nr; ++i)
        if (T.iou[static_cast<size_t>(i) * T.ldi + j] > T.gate) { ++c; last = i; }
      W.y[j] = (c == 1) ? last : -1;
      if (c > max_col) max_col = c;
    }
    max_row = g.reduce_max(max_row);
    max_col = g.reduce_max(max_col);
    g.sync();
    if (max_row == 1 && max_col == 1) path = 1;
  }

  if (path == 0) {
    mot::LapProblem P{T.cost, T.ldc, nr, nc, static_cast<double>(T.thresh) / 2.0};
    mot::lap_solve(g, P, W);
    g.sync();
    for (int i = t; i < nr; i += kThreads) { const int v = W.x[i]; W.x[i] = (v >= nc) ? -1 : v; }
    for (int j = t; j < nc; j += kThreads) { const int v = W.y[j]; W.y[j] = (v >= nr) ? -1 : v; }
  } else if (path == 2) {
    for (int i = t; i < nr; i += kThreads) W.x[i] = -1;
    for (int j = t; j < nc; j += kThreads) W.y[j] = -1;
  }
  g.sync();
  for (int i = t; i < nr; i += kThreads) {
    const int xi = W.x[i];
    T.x[i] = xi;
    if (T.xval) {
      float v = 0.f;
      if (xi >= 0) v = T.iou ? T.iou[static_cast<size_t>(i) * T.ldi + xi] : T.cost[static_cast<size_t>(i) * T.ldc + xi];
      T.xval[i] = v;
    }
  }
  for (int j = t; j < nc; j += kThreads) T.y[j] = W.y[j];
  if (T.info && t == 0) T.info[0] = path;
}

}  // namespace

namespace mot {
int lap_lds_limit() { return static_cast<int>((kLdsBudget - kScratch - 64) / lap_work_bytes(1)); }

hipError_t launch_lap(const mot_lap_task* tasks, int ntasks, int max_nm, hipStream_t st) {
  if (ntasks <= 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (max_nm <= lap_lds_limit()) {
    const size_t lds = kScratch + ((lap_work_bytes(max_nm > 0 ? max_nm : 1) + 15) & ~size_t(15));
    hipLaunchKernelGGL(lap_kernel<true>, dim3(ntasks), dim3(kThreads), lds, st, tasks);
  } else {
    hipLaunchKernelGGL(lap_kernel<false>, dim3(ntasks), dim3(kThreads), kScratch, st, tasks);
  }
  return hipGetLastError();
}
}  // namespace mot
