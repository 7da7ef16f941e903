// BoostTrack's Kalman filter and cost matrices on gfx950 (reference: src/trackers/boosttrack.cpp). State [cx, cy, h, r] + velocities with
// CONSTANT process / measurement noise (BoostKalmanFilter :22-75) — unlike the box-size-dependent noise of the filters in kf_kernels.hip —
// in the same 72-float records (mean[8], covariance[64]). Passes (mot_boost_task, include/motcpp_amd.h): predict (+ the predicted boxes),
// the detection-confidence boost's row maxima of the IoU (dlo_confidence_boost :361-426), the association cost (1 - IoU minus the
// weighted Mahalanobis similarity, :567-611), update, births, output boxes. One lane per track / pair; float operations in the
// order of the reference's expressions (F, H, Q, R are sparse: predict and the projection have at most two non-zero terms per sum; the
// gain, state and covariance updates are k-ordered chains; S^-1 = the partial-pivot LU inverse Eigen uses for a dynamic 4 x 4).
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"
#include "kf_small.hpp"

namespace {
constexpr int kT = 128;
constexpr int kRec = 72;

__device__ __forceinline__ float fmaxs(float a, float b) { return (a < b) ? b : a; }  // std::max
__device__ __forceinline__ float fmins(float a, float b) { return (b < a) ? b : a; }  // std::min

__device__ __forceinline__ void state_box(const float* x, float b[4]) {  // get_state :107-115
  const float cx = x[0], cy = x[1], h = x[2], r = x[3];
  const float w = r * h;
  b[0] = cx - w / 2; b[1] = cy - h / 2; b[2] = cx + w / 2; b[3] = cy + h / 2;
}
__device__ __forceinline__ void to_z(const float b[4], float z[4]) {  // convert_bbox_to_z :126-134
  const float w = b[2] - b[0], h = b[3] - b[1];
  z[0] = b[0] + w / 2.0f; z[1] = b[1] + h / 2.0f; z[2] = h; z[3] = (h > 1e-6f) ? w / h : 0.0f;
}
__device__ __forceinline__ void load_det(const mot_boost_task& T, int c, float b[4]) {
  const size_t ld = static_cast<size_t>(T.ldd);
  b[0] = T.dets[c]; b[1] = T.dets[ld + c]; b[2] = T.dets[2 * ld + c]; b[3] = T.dets[3 * ld + c];
}

__global__ void __launch_bounds__(kT) boost_predict_kernel(const mot_boost_task* __restrict__ tasks) {
  const mot_boost_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  float x[8], P[8][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = rec[k];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) P[r][c] = rec[8 + r * 8 + c];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = x[k] + x[k + 4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) P[r][c] = P[r][c] + P[r + 4][c];  // F P
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) P[r][c] = P[r][c] + P[r][c + 4];  // (F P) F^T
#pragma unroll
  for (int k = 0; k < 4; ++k) { P[k][k] = P[k][k] + 10.0f; P[k + 4][k + 4] = P[k + 4][k + 4] + 0.01f; }
#pragma unroll
  for (int k = 0; k < 8; ++k) rec[k] = x[k];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) rec[8 + r * 8 + c] = P[r][c];
  if (T.boxes) {
    float b[4];
    state_box(x, b);
#pragma unroll
    for (int k = 0; k < 4; ++k) T.boxes[static_cast<size_t>(i) * 4 + k] = b[k];
  }
}

__global__ void __launch_bounds__(kT) boost_boxes_kernel(const mot_boost_task* __restrict__ tasks) {
  const mot_boost_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  const float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  const float x[4] = {rec[0], rec[1], rec[2], rec[3]};
  float b[4];
  state_box(x, b);
#pragma unroll
  for (int k = 0; k < 4; ++k) T.boxes[static_cast<size_t>(i) * 4 + k] = b[k];
}

// dlo_confidence_boost's per-detection reductions over the tracks: one lane per detection (tracks are a few hundred at most)
__global__ void __launch_bounds__(kT) boost_dlo_kernel(const mot_boost_task* __restrict__ tasks) {
  const mot_boost_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float a[4];
  load_det(T, i, a);
  const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
  float max_s = 0.0f;
  int vt = 0;
  for (int j = 0; j < T.m; ++j) {
    const float* b = T.boxes + static_cast<size_t>(j) * 4;
    const float a2 = (b[2] - b[0]) * (b[3] - b[1]);  // utils::iou_batch (iou.hpp:63-100)
    const float w = fmaxs(0.0f, fmins(a[2], b[2]) - fmaxs(a[0], b[0])), h = fmaxs(0.0f, fmins(a[3], b[3]) - fmaxs(a[1], b[1]));
    const float inter = w * h, uni = a1 + a2 - inter;
    const float s = (uni > 0.0f) ? (inter / uni) : 0.0f;
    if (j == 0 || s > max_s) max_s = s;
    const float th = fmaxs(0.95f - static_cast<float>(T.tsu[j] - 1), 0.8f);
    if (s > th) vt = 1;
  }
  T.max_s[i] = max_s;
  T.vt[i] = vt;
}

__global__ void __launch_bounds__(kT) boost_cost_kernel(const mot_boost_task* __restrict__ tasks) {
  const mot_boost_task& T = tasks[blockIdx.z];
  const int i = blockIdx.y;
  const int j = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n || j >= T.m) return;
  float b[4], z[4];
  load_det(T, T.didx ? T.didx[i] : i, b);
  to_z(b, z);
  const float* rec = T.slab + static_cast<size_t>(T.slots[j]) * kRec;
  const float x[4] = {rec[0], rec[1], rec[2], rec[3]};
  float t[4];
  state_box(x, t);
  // get_iou_matrix :297-329
  const float x1 = fmaxs(b[0], t[0]), y1 = fmaxs(b[1], t[1]), x2 = fmins(b[2], t[2]), y2 = fmins(b[3], t[3]);
  const float inter = fmaxs(0.0f, x2 - x1) * fmaxs(0.0f, y2 - y1);
  const float da = (b[2] - b[0]) * (b[3] - b[1]), ta = (t[2] - t[0]) * (t[3] - t[1]);
  const float uni = da + ta - inter;
  const float iou = (uni > 1e-6f) ? inter / uni : 0.0f;
  float c = 1.0f - iou;
  // get_mh_dist_matrix :331-359 and the similarity :598-611
  const float limit = 13.2767f;
  float mh = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float df = z[k] - x[k];
    const float term = df * df * (1.0f / rec[8 + k * 9]);
    mh = (k == 0) ? term : mh + term;
  }
  if (mh > limit) mh = limit;
  const float sim = (limit - mh) / limit;
  c = c - T.lambda_mhd * sim;
  if (T.emb) c = c - T.lambda_emb * ((T.emb[static_cast<size_t>(i) * T.lde + j] + 1.0f) / 2.0f);  // :613-618
  T.cost[static_cast<size_t>(i) * T.ldc + j] = c;
}

// BoostKalmanFilter::update :61-75
__global__ void __launch_bounds__(kT) boost_update_kernel(const mot_boost_task* __restrict__ tasks) {
  const mot_boost_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  float b[4], z[4];
  load_det(T, T.didx[i], b);
  to_z(b, z);
  const float Rd[4] = {1.0f, 1.0f, 10.0f, 0.01f};
  float S[4][4], Si[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) S[r][c] = rec[8 + r * 8 + c] + ((r == c) ? Rd[r] : 0.0f);
  mot::kfs::inv_lu4(S, Si);
  float K[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = rec[8 + r * 8 + 0] * Si[0][c];
#pragma unroll
      for (int k = 1; k < 4; ++k) a += rec[8 + r * 8 + k] * Si[k][c];
      K[r][c] = a;
    }
  float inn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) inn[k] = z[k] - rec[k];
  float KS[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = K[r][0] * S[0][c];
#pragma unroll
      for (int k = 1; k < 4; ++k) a += K[r][k] * S[k][c];
      KS[r][c] = a;
    }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float a = K[r][0] * inn[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) a += K[r][k] * inn[k];
    rec[r] = rec[r] + a;
  }
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = KS[r][0] * K[c][0];
#pragma unroll
      for (int k = 1; k < 4; ++k) a += KS[r][k] * K[c][k];
      rec[8 + r * 8 + c] = rec[8 + r * 8 + c] - a;
    }
}

__global__ void __launch_bounds__(kT) boost_init_kernel(const mot_boost_task* __restrict__ tasks) {
  const mot_boost_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  float* rec = T.slab + static_cast<size_t>(T.slots[i]) * kRec;
  float b[4], z[4];
  load_det(T, T.didx[i], b);
  to_z(b, z);
#pragma unroll
  for (int k = 0; k < 4; ++k) { rec[k] = z[k]; rec[k + 4] = 0.0f; }
#pragma unroll
  for (int k = 0; k < 64; ++k) rec[8 + k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { rec[8 + k * 9] = 10.0f; rec[8 + (k + 4) * 9] = 10.0f * 1000.0f; }
}
}  // namespace

namespace mot {
hipError_t launch_boost(int op, const mot_boost_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  if (ntasks > 65535) return hipErrorInvalidValue;
  const dim3 lin((max_n + kT - 1) / kT, ntasks);
  switch (op) {
    case MOT_BOOST_PREDICT: hipLaunchKernelGGL(boost_predict_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_BOOST_DLO: hipLaunchKernelGGL(boost_dlo_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_BOOST_COST:
      if (max_m <= 0) return hipSuccess;
      if (max_n > 65535) return hipErrorInvalidValue;
      hipLaunchKernelGGL(boost_cost_kernel, dim3((max_m + kT - 1) / kT, max_n, ntasks), dim3(kT), 0, st, tasks);
      break;
    case MOT_BOOST_UPDATE: hipLaunchKernelGGL(boost_update_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_BOOST_INIT: hipLaunchKernelGGL(boost_init_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_BOOST_BOXES: hipLaunchKernelGGL(boost_boxes_kernel, lin, dim3(kT), 0, st, tasks); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace mot
