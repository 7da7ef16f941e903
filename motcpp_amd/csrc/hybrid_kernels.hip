// HybridSORT's Kalman filter and pairwise costs on gfx950 (reference: src/trackers/hybridsort.cpp). Nine-state filter [u, v, s, c, r] +
// velocities of the first four with constant noise (HybridKalmanFilter :26-88) in records of 90 floats; the association costs the
// reference actually uses (:529-577, :619-716, :1052-1179): IoU with its 1e-6 guard, optionally times the height overlap (HMIoU), minus a
// weighted score difference. One lane per track / pair; float operations in the order of the reference's expressions (predict: two-term
// sums; gain, state and covariance updates: k-ordered chains; S^-1: the partial-pivot LU inverse Eigen uses for a dynamic 5 x 5).
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"

namespace {
constexpr int kT = 128;
constexpr int kRec = 90;

__device__ __forceinline__ float fmaxs(float a, float b) { return (a < b) ? b : a; }  // std::max
__device__ __forceinline__ float fmins(float a, float b) { return (b < a) ? b : a; }  // std::min

__device__ __forceinline__ void inv_lu5(float lu[5][5], float inv[5][5]) {  // (lu: S on entry, destroyed)
  int perm[5] = {0, 1, 2, 3, 4};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int p = k;
    float best = fabsf(lu[k][k]);
#pragma unroll
    for (int i = k + 1; i < 5; ++i) { const float v = fabsf(lu[i][k]); if (v > best) { best = v; p = i; } }
#pragma unroll
    for (int i = k + 1; i < 5; ++i) {
      if (p == i) {
#pragma unroll
        for (int j = 0; j < 5; ++j) { const float t = lu[k][j]; lu[k][j] = lu[i][j]; lu[i][j] = t; }
        const int tp = perm[k]; perm[k] = perm[i]; perm[i] = tp;
      }
    }
#pragma unroll
    for (int i = k + 1; i < 5; ++i) lu[i][k] /= lu[k][k];
#pragma unroll
    for (int i = k + 1; i < 5; ++i)
#pragma unroll
      for (int j = k + 1; j < 5; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    float b[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) b[i] = (perm[i] == c) ? 1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = i + 1; r < 5; ++r) b[r] -= b[i] * lu[r][i];
#pragma unroll
    for (int i = 4; i >= 0; --i) {
      b[i] /= lu[i][i];
#pragma unroll
      for (int r = 0; r < i; ++r) b[r] -= b[i] * lu[r][i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) inv[i][c] = b[i];
  }
}

__device__ __forceinline__ void det_to_z(const mot_hyb_task& T, int c, float z[5]) {  // convert_bbox_to_z :181-193
  const size_t ld = static_cast<size_t>(T.ldd);
  const float x1 = T.dets[c], y1 = T.dets[ld + c], x2 = T.dets[2 * ld + c], y2 = T.dets[3 * ld + c], cf = T.dets[4 * ld + c];
  const float w = x2 - x1, h = y2 - y1;
  z[0] = x1 + w / 2.0f; z[1] = y1 + h / 2.0f; z[2] = w * h; z[3] = cf; z[4] = (h > 1e-6f) ? w / h : 0.0f;
}

__global__ void __launch_bounds__(kT) hyb_predict_kernel(const mot_hyb_task* __restrict__ tasks) {
  const mot_hyb_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  float* P = rec + 9;
  if (rec[7] + rec[2] <= 0) rec[7] = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) rec[k] = rec[k] + rec[k + 5];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 9; ++c) P[r * 9 + c] = P[r * 9 + c] + P[(r + 5) * 9 + c];  // F P
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 4; ++c) P[r * 9 + c] = P[r * 9 + c] + P[r * 9 + c + 5];    // (F P) F^T
#pragma unroll
  for (int k = 0; k < 5; ++k) P[k * 10] = P[k * 10] + 0.1f;
#pragma unroll
  for (int k = 5; k < 9; ++k) P[k * 10] = P[k * 10] + 0.01f;
}

__global__ void __launch_bounds__(kT) hyb_update_kernel(const mot_hyb_task* __restrict__ tasks) {
  const mot_hyb_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  float* P = rec + 9;
  float z[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (T.didx[i] >= 0) det_to_z(T, T.didx[i], z);
  const float Rd[5] = {1.0f, 1.0f, 10.0f, 0.01f, 1.0f};
  float S[5][5], lu[5][5], Si[5][5];
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) { S[r][c] = P[r * 9 + c] + ((r == c) ? Rd[r] : 0.0f); lu[r][c] = S[r][c]; }
  inv_lu5(lu, Si);
  float K[9][5];
  for (int r = 0; r < 9; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      float a = P[r * 9 + 0] * Si[0][c];
#pragma unroll
      for (int k = 1; k < 5; ++k) a += P[r * 9 + k] * Si[k][c];
      K[r][c] = a;
    }
  float inn[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) inn[k] = z[k] - rec[k];
  for (int r = 0; r < 9; ++r) {
    float a = K[r][0] * inn[0];
#pragma unroll
    for (int k = 1; k < 5; ++k) a += K[r][k] * inn[k];
    rec[r] = rec[r] + a;
  }
  // P = (I - K H) P, column by column (a column of the new P needs the same column of the old one only)
  for (int c = 0; c < 9; ++c) {
    float col[9], out[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) col[k] = P[k * 9 + c];
    for (int r = 0; r < 9; ++r) {
      float a = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float A = ((r == k) ? 1.0f : 0.0f) - ((k < 5) ? K[r][k] : 0.0f);
        const float t = A * col[k];
        a = (k == 0) ? t : a + t;
      }
      out[r] = a;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) P[k * 9 + c] = out[k];
  }
}

__global__ void __launch_bounds__(kT) hyb_init_kernel(const mot_hyb_task* __restrict__ tasks) {
  const mot_hyb_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  float z[5];
  det_to_z(T, T.didx[i], z);
#pragma unroll
  for (int k = 0; k < 5; ++k) rec[k] = z[k];
#pragma unroll
  for (int k = 5; k < 9; ++k) rec[k] = 0.0f;
  for (int k = 0; k < 81; ++k) rec[9 + k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 5; ++k) rec[9 + k * 10] = 10.0f;
#pragma unroll
  for (int k = 5; k < 9; ++k) rec[9 + k * 10] = 10.0f * 1000.0f;
}

__global__ void __launch_bounds__(kT) hyb_boxes_kernel(const mot_hyb_task* __restrict__ tasks) {
  const mot_hyb_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  const float* x = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  const float u = x[0], v = x[1], s = x[2], r = x[4];  // convert_x_to_bbox :195-201
  const float w = sqrtf(s * r);
  const float h = s / w;
  float* b = T.boxes + static_cast<size_t>(i) * 4;
  b[0] = u - w / 2; b[1] = v - h / 2; b[2] = u + w / 2; b[3] = v + h / 2;
}

__global__ void __launch_bounds__(kT) hyb_pair_kernel(const mot_hyb_task* __restrict__ tasks) {
  const mot_hyb_task& T = tasks[blockIdx.z];
  const int i = blockIdx.y;
  const int j = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n || j >= T.m) return;
  const float* a = T.a + static_cast<size_t>(i) * 4;
  const float* b = T.b + static_cast<size_t>(j) * 4;
  const float xx1 = fmaxs(a[0], b[0]), yy1 = fmaxs(a[1], b[1]), xx2 = fmins(a[2], b[2]), yy2 = fmins(a[3], b[3]);
  const float w = fmaxs(0.0f, xx2 - xx1), h = fmaxs(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float a1 = (a[2] - a[0]) * (a[3] - a[1]), a2 = (b[2] - b[0]) * (b[3] - b[1]);
  const float uni = a1 + a2 - inter;
  float v = (uni > 1e-6f) ? inter / uni : 0.0f;
  if (T.hmiou) {
    const float yy3 = fmins(a[1], b[1]), yy4 = fmaxs(a[3], b[3]);
    const float ho = fmaxs(0.0f, yy2 - yy1) / (yy4 - yy3 + 1e-6f);
    v *= ho;
  }
  if (T.score_w != 0.0f) v -= fabsf(T.b_score[j] - T.a_score[i]) * T.score_w;
  T.sim[static_cast<size_t>(i) * T.ldc + j] = v;
  float c = 1.0f - v;
  if (T.scale_first) c = c * 1.0f;
  if (T.add_const != 0.0f) c += T.add_const;
  T.cost[static_cast<size_t>(i) * T.ldc + j] = c;
}
}  // namespace

namespace mot {
hipError_t launch_hyb(int op, const mot_hyb_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  if (ntasks > 65535) return hipErrorInvalidValue;
  const dim3 lin((max_n + kT - 1) / kT, ntasks);
  switch (op) {
    case MOT_HYB_PREDICT: hipLaunchKernelGGL(hyb_predict_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_HYB_UPDATE: hipLaunchKernelGGL(hyb_update_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_HYB_INIT: hipLaunchKernelGGL(hyb_init_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_HYB_BOXES: hipLaunchKernelGGL(hyb_boxes_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_HYB_PAIR:
      if (max_m <= 0) return hipSuccess;
      if (max_n > 65535) return hipErrorInvalidValue;
      hipLaunchKernelGGL(hyb_pair_kernel, dim3((max_m + kT - 1) / kT, max_n, ntasks), dim3(kT), 0, st, tasks);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace mot
