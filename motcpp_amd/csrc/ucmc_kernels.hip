// UCMCTrack's ground-plane Kalman filter on gfx950 (reference: src/trackers/ucmc.cpp). State [x, vx, y, vy] and a 4 x 4 covariance per
// track, both in DOUBLE precision like the reference's Eigen::Vector4d / Matrix4d; a detection enters as its foot point mapped to the
// ground plane with a 2 x 2 covariance. Five passes (mot_ucmc_task, include/motcpp_amd.h): map the detections, predict, the cost
// matrix (Mahalanobis distance + log det S, cast to float for the assignment), the Joseph-form update, births.
//
// Arithmetic contract: every operation is an IEEE double operation in the order the reference's expressions evaluate them
// (-ffp-contract=off: no fused multiply-add). F and H are sparse, so most of the reference's inner products have at most two non-zero
// terms and no order can change them; the three- and four-term sums (the camera mapping's 3 x 3 product, (I - K H) P (I - K H)^T) are
// added in index order. The one value that is not correctly rounded is log(det S) (the device's double-precision log).
//
// These are tiny passes (tens of tracks per stream): one lane per track / pair, HBM traffic a few hundred bytes per track; they exist
// so that UCMCTrack's states never leave the device, not because they are hot.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"

namespace {
constexpr int kT = 128;

__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }  // std::max
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }  // std::min

struct Innov { double S[4], SI[4], det; };
// S = H P H^T + R (H selects rows / columns 0 and 2 of P), S^-1 = adj(S) * (1 / det) (Eigen's 2 x 2 inverse)
__device__ __forceinline__ Innov innovation(const double* P, const double* R) {
  Innov v;
  v.S[0] = P[0] + R[0]; v.S[1] = P[2] + R[1]; v.S[2] = P[8] + R[2]; v.S[3] = P[10] + R[3];
  v.det = v.S[0] * v.S[3] - v.S[2] * v.S[1];
  const double invdet = 1.0 / v.det;
  v.SI[0] = v.S[3] * invdet; v.SI[2] = -v.S[2] * invdet; v.SI[1] = -v.S[1] * invdet; v.SI[3] = v.S[0] * invdet;
  return v;
}

// CameraMapper::mapToGroundPlane / mapToImageSpace (:114-146), uv2xy (:92-112)
__global__ void __launch_bounds__(kT) ucmc_map_kernel(const mot_ucmc_task* __restrict__ tasks) {
  const mot_ucmc_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  const int c = T.didx ? T.didx[i] : i;
  const size_t ld = static_cast<size_t>(T.ld);
  const float x1 = T.dets[c], y1 = T.dets[ld + c], x2 = T.dets[2 * ld + c], y2 = T.dets[3 * ld + c];
  const float w = x2 - x1, h = y2 - y1;
  const float cx = (x1 + x2) / 2.0f, bottom = y2;
  double* y = T.y + static_cast<size_t>(i) * 2;
  double* R = T.R + static_cast<size_t>(i) * 4;
  if (!T.mapped) {
    const double scale = 0.01;
    y[0] = cx * scale; y[1] = bottom * scale;
    const double ex = dmax(0.02, dmin(0.13, 0.0005 * w)), ey = dmax(0.02, dmin(0.10, 0.0005 * h));
    R[0] = ex * ex; R[1] = 0.0; R[2] = 0.0; R[3] = ey * ey;
    return;
  }
  const double eu = dmax(2.0, dmin(13.0, 0.05 * w)), ev = dmax(2.0, dmin(10.0, 0.05 * h));  // uvError :85-90
  const double su[4] = {eu * eu, 0.0, 0.0, ev * ev};
  const double* A = T.invA;
  const double u = static_cast<double>(cx), v = static_cast<double>(bottom);
  double b[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) b[r] = (A[r * 3 + 0] * u + A[r * 3 + 1] * v) + A[r * 3 + 2] * 1.0;
  const double gamma = 1.0 / b[2];
  double C[4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) C[r * 2 + q] = gamma * A[r * 3 + q] - ((gamma * gamma) * b[r]) * A[2 * 3 + q];
  y[0] = b[0] * gamma; y[1] = b[1] * gamma;
  double Cs[4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) Cs[r * 2 + q] = C[r * 2 + 0] * su[0 * 2 + q] + C[r * 2 + 1] * su[1 * 2 + q];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) R[r * 2 + q] = Cs[r * 2 + 0] * C[q * 2 + 0] + Cs[r * 2 + 1] * C[q * 2 + 1];
}

// UCMCKalmanFilter::predict :28-31: x = F x, P = (F P) F^T + Q
__global__ void __launch_bounds__(kT) ucmc_predict_kernel(const mot_ucmc_task* __restrict__ tasks) {
  const mot_ucmc_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  const int s = T.slots[i];
  double* xs = T.x + static_cast<size_t>(s) * 4;
  double* Ps = T.P + static_cast<size_t>(s) * 16;
  const double dt = T.dt;
  const double F[4][4] = {{1.0, dt, 0.0, 0.0}, {0.0, 1.0, 0.0, 0.0}, {0.0, 0.0, 1.0, dt}, {0.0, 0.0, 0.0, 1.0}};
  double x[4], P[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = xs[k];
#pragma unroll
  for (int k = 0; k < 16; ++k) P[k] = Ps[k];
  double nx[4], FP[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) a += F[r][k] * x[k];
    nx[r] = a;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += F[r][k] * P[k * 4 + q];
      FP[r * 4 + q] = a;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += FP[r * 4 + k] * F[q][k];
      Ps[r * 4 + q] = a + T.Q[r * 4 + q];
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) xs[k] = nx[k];
}

// UCMCSingleTrack::distance :213-223, then the cast to float the tracker applies to the whole matrix (:398)
__global__ void __launch_bounds__(kT) ucmc_cost_kernel(const mot_ucmc_task* __restrict__ tasks) {
  const mot_ucmc_task& T = tasks[blockIdx.z];
  const int i = blockIdx.y;
  const int j = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n || j >= T.m) return;
  const int s = T.slots[i], d = T.didx ? T.didx[j] : j;
  const double* x = T.x + static_cast<size_t>(s) * 4;
  const double* P = T.P + static_cast<size_t>(s) * 16;
  const double* y = T.y + static_cast<size_t>(d) * 2;
  const double* R = T.R + static_cast<size_t>(d) * 4;
  const double d0 = y[0] - x[0], d1 = y[1] - x[2];
  const Innov v = innovation(P, R);
  const double r0 = d0 * v.SI[0] + d1 * v.SI[2], r1 = d0 * v.SI[1] + d1 * v.SI[3];
  const double maha = r0 * d0 + r1 * d1;
  T.cost[static_cast<size_t>(i) * T.ldc + j] = static_cast<float>(maha + log(v.det));
}

// UCMCKalmanFilter::update :33-49
__global__ void __launch_bounds__(kT) ucmc_update_kernel(const mot_ucmc_task* __restrict__ tasks) {
  const mot_ucmc_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  const int s = T.slots[i], d = T.didx[i];
  double* xs = T.x + static_cast<size_t>(s) * 4;
  double* Ps = T.P + static_cast<size_t>(s) * 16;
  const double* z = T.y + static_cast<size_t>(d) * 2;
  const double* R = T.R + static_cast<size_t>(d) * 4;
  double x[4], P[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = xs[k];
#pragma unroll
  for (int k = 0; k < 16; ++k) P[k] = Ps[k];
  const double y0 = z[0] - x[0], y1 = z[1] - x[2];
  const Innov v = innovation(P, R);
  double K[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double p0 = P[r * 4 + 0], p1 = P[r * 4 + 2];  // P H^T
    K[r][0] = p0 * v.SI[0] + p1 * v.SI[2];
    K[r][1] = p0 * v.SI[1] + p1 * v.SI[3];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) xs[r] = x[r] + (K[r][0] * y0 + K[r][1] * y1);
  double A[4][4];  // I - K H
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double kh = (q == 0) ? K[r][0] : ((q == 2) ? K[r][1] : 0.0);
      A[r][q] = ((r == q) ? 1.0 : 0.0) - kh;
    }
  double AP[4][4], KR[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += A[r][k] * P[k * 4 + q];
      AP[r][q] = a;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) KR[r][q] = K[r][0] * R[0 * 2 + q] + K[r][1] * R[1 * 2 + q];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += AP[r][k] * A[q][k];
      Ps[r * 4 + q] = a + (KR[r][0] * K[q][0] + KR[r][1] * K[q][1]);
    }
}

// UCMCSingleTrack ctor :152-201
__global__ void __launch_bounds__(kT) ucmc_init_kernel(const mot_ucmc_task* __restrict__ tasks) {
  const mot_ucmc_task& T = tasks[blockIdx.y];
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= T.n) return;
  const int s = T.slots[i], d = T.didx[i];
  double* xs = T.x + static_cast<size_t>(s) * 4;
  double* Ps = T.P + static_cast<size_t>(s) * 16;
  const double* y = T.y + static_cast<size_t>(d) * 2;
  xs[0] = y[0]; xs[1] = 0.0; xs[2] = y[1]; xs[3] = 0.0;
  const double pv = T.vmax * T.vmax / 3.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) Ps[k] = 0.0;
  Ps[0] = 1.0; Ps[5] = pv; Ps[10] = 1.0; Ps[15] = pv;
}
}  // namespace

namespace mot {
hipError_t launch_ucmc(int op, const mot_ucmc_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  if (ntasks > 65535) return hipErrorInvalidValue;
  const dim3 lin((max_n + kT - 1) / kT, ntasks);
  switch (op) {
    case MOT_UCMC_MAP: hipLaunchKernelGGL(ucmc_map_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_UCMC_PREDICT: hipLaunchKernelGGL(ucmc_predict_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_UCMC_COST:
      if (max_m <= 0) return hipSuccess;
      if (max_n > 65535) return hipErrorInvalidValue;
      hipLaunchKernelGGL(ucmc_cost_kernel, dim3((max_m + kT - 1) / kT, max_n, ntasks), dim3(kT), 0, st, tasks);
      break;
    case MOT_UCMC_UPDATE: hipLaunchKernelGGL(ucmc_update_kernel, lin, dim3(kT), 0, st, tasks); break;
    case MOT_UCMC_INIT: hipLaunchKernelGGL(ucmc_init_kernel, lin, dim3(kT), 0, st, tasks); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace mot
