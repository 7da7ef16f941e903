// StrongSORT's two cost matrices on gfx950 (reference: src/trackers/strongsort.cpp).
//   ss_nn_kernel   NearestNeighborDistanceMetric::distance (:239-275, cosine :306-334): per track the MINIMUM over its stored samples of
//                  1 - <sample, feature>; the inner products come from the fp32 MFMA kernel (cosine_mfma.hip, raw dot products of the
//                  re-normalised rows), this pass reduces them. A track without samples costs 1e5 (:271).
//   ss_iou_kernel  iou_matching::iou_cost (:500-583): 1 - IoU on (top-left, width, height) boxes with the reference's own arithmetic —
//                  the far corner is top-left + size, the areas are width x height, union <= 1e-6 gives IoU 0 — a track that was not
//                  updated in the previous frame costs 1e5 for every detection (:563-566), and min_cost_matching's clamp (:376-379:
//                  anything above max_distance becomes max_distance + 1e-5) applied on the way out.
// Both are plain HBM-bound passes: one output element per lane, coalesced along the detections.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"

namespace {
constexpr int kT = 256;

__global__ void __launch_bounds__(kT) ss_nn_kernel(const mot_ss_nn_task* __restrict__ tasks) {
  const mot_ss_nn_task T = tasks[blockIdx.z];
  const int i = blockIdx.y;
  const int j = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n || j >= T.m) return;
  const int s0 = T.soff[i], s1 = T.soff[i + 1];
  float c = 1e5f;
  if (s1 > s0) {
    c = 1.0f - T.dots[static_cast<size_t>(s0) * T.ldd + j];
    for (int s = s0 + 1; s < s1; ++s) {
      const float d = 1.0f - T.dots[static_cast<size_t>(s) * T.ldd + j];
      c = (d < c) ? d : c;  // colwise().minCoeff()
    }
  }
  T.cost[static_cast<size_t>(i) * T.ldc + j] = c;
}

__global__ void __launch_bounds__(kT) ss_iou_kernel(const mot_ss_iou_task* __restrict__ tasks) {
  const mot_ss_iou_task T = tasks[blockIdx.z];
  const int i = blockIdx.y;
  const int j = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n || j >= T.m) return;
  float cost;
  if (T.stale && T.stale[i]) cost = 1e5f;
  else {
    const int slot = T.src ? T.src[i] : i;
    const float* mean = T.mean + static_cast<size_t>(slot) * 72;  // 8-state record: mean then covariance
    // Track::to_tlwh :94-100
    const float bh = mean[3];
    const float bw = mean[2] * bh;
    const float bx = mean[0] - bw / 2.0f, by = mean[1] - bh / 2.0f;
    const float bx2 = bx + bw, by2 = by + bh, ab = bw * bh;
    const int dj = T.didx ? T.didx[j] : j;
    const float cx = T.dtlwh[dj], cy = T.dtlwh[static_cast<size_t>(T.ldd) + dj];
    const float cw = T.dtlwh[static_cast<size_t>(2) * T.ldd + dj], ch = T.dtlwh[static_cast<size_t>(3) * T.ldd + dj];
    const float cx2 = cx + cw, cy2 = cy + ch;
    const float tlx = (bx < cx) ? cx : bx, tly = (by < cy) ? cy : by;          // std::max
    const float brx = (cx2 < bx2) ? cx2 : bx2, bry = (cy2 < by2) ? cy2 : by2;  // std::min
    const float dw = brx - tlx, dh = bry - tly;
    const float w = (0.0f < dw) ? dw : 0.0f, h = (0.0f < dh) ? dh : 0.0f;      // std::max(0.0f, .)
    const float ai = w * h, ac = cw * ch;
    const float au = ab + ac - ai;
    const float iou = (au > 1e-6f) ? (ai / au) : 0.0f;
    cost = 1.0f - iou;
  }
  if (cost > T.max_dist) cost = T.max_dist + 1e-5f;
  T.cost[static_cast<size_t>(i) * T.ldc + j] = cost;
}
}  // namespace

namespace mot {
hipError_t launch_ss_nn(const mot_ss_nn_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  if (max_n > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ss_nn_kernel, dim3((max_m + kT - 1) / kT, max_n, ntasks), dim3(kT), 0, st, tasks);
  return hipGetLastError();
}
hipError_t launch_ss_iou(const mot_ss_iou_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  if (max_n > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ss_iou_kernel, dim3((max_m + kT - 1) / kT, max_n, ntasks), dim3(kT), 0, st, tasks);
  return hipGetLastError();
}
}  // namespace mot
