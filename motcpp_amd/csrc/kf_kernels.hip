// Batched Kalman filters and detection preparation for gfx950 (wave64).
//
// Layout: track state lives in HBM as an array of records — slot s holds mean[D] then the covariance row-major at
// mean + s*(D + D*D) (288 B for the 8-state filters, 224 B for XYSR) — because the items of a launch are a GATHER
// (the matched tracks of a frame, a pool in list order): with a struct-of-arrays slab every one of a wavefront's D + D*D
// loads then touches up to 64 different cache lines (measured: the update kernel at 3.5 % of the HBM roof), while a
// record is one contiguous run whatever the slot is.
// One wavefront per 64 items, one lane per track for the arithmetic (7x7 / 8x8 tile in VGPRs, fully unrolled constant
// indexing, no scratch). The records travel through an LDS tile: the wavefront fetches them with 16-byte loads, three
// (8-state) or four (7-state) whole records per instruction — every byte of every line is used — drops them into the tile
// (row stride D + D*D + 4 floats: the lane-per-track ds_read_b128 that follow are bank-conflict free), computes, and
// scatters the new records back the same way.
//
// Arithmetic: fp32, built with -ffp-contract=off, correctly rounded / and sqrt; every inner
// product is accumulated in k order exactly like the CPU restatement so states are bit-identical
// to it (F = I + shift and H = [I 0] are applied structurally: the skipped terms are exact zeros).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/motcpp_amd.h"
#include "kf_small.hpp"

namespace {

constexpr int kThreads = 64;

template <int D>
struct St {
  float m[D];
  float P[D][D];
};

template <int D> constexpr int rec_floats() { return D + D * D; }
template <int D> constexpr int tile_stride() { return D + D * D + 4; }

// lane-per-track view of the tile: record `lane` <-> registers
template <int D>
__device__ __forceinline__ void tile_to_state(St<D>& s, const float* tile, int lane) {
  constexpr int Q = rec_floats<D>() / 4;
  const float4* row = reinterpret_cast<const float4*>(tile + lane * tile_stride<D>());
  float flat[rec_floats<D>()];
#pragma unroll
  for (int q = 0; q < Q; ++q) { const float4 v = row[q]; flat[4 * q] = v.x; flat[4 * q + 1] = v.y; flat[4 * q + 2] = v.z; flat[4 * q + 3] = v.w; }
#pragma unroll
  for (int k = 0; k < D; ++k) s.m[k] = flat[k];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) s.P[r][c] = flat[D + r * D + c];
}
template <int D>
__device__ __forceinline__ void state_to_tile(const St<D>& s, float* tile, int lane) {
  constexpr int Q = rec_floats<D>() / 4;
  float flat[rec_floats<D>()];
#pragma unroll
  for (int k = 0; k < D; ++k) flat[k] = s.m[k];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) flat[D + r * D + c] = s.P[r][c];
  float4* row = reinterpret_cast<float4*>(tile + lane * tile_stride<D>());
#pragma unroll
  for (int q = 0; q < Q; ++q) row[q] = make_float4(flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]);
}
// The wavefront moves the records of its 64 items between the slab and the tile: lane l handles 16-byte piece l % Q of
// record (l / Q) of each group of R = 64 / Q records; the record's slot comes from the lane that owns the item (-1: none).
template <int D, bool TO_TILE>
__device__ __forceinline__ void move_records(float* slab, float* tile, int my_slot, int lane) {
  constexpr int Q = rec_floats<D>() / 4, R = 64 / Q;
  const int rr = lane / Q, q = lane - rr * Q;
#pragma unroll
  for (int g = 0; g < (64 + R - 1) / R; ++g) {
    const int rec = g * R + rr;
    const int slot = __shfl(my_slot, (rec < 64) ? rec : 0, 64);
    if (rr < R && rec < 64 && slot >= 0) {
      float4* gp = reinterpret_cast<float4*>(slab + static_cast<size_t>(slot) * rec_floats<D>()) + q;
      float4* tp = reinterpret_cast<float4*>(tile + rec * tile_stride<D>()) + q;
      if (TO_TILE) *tp = *gp; else *gp = *tp;
    }
  }
}

// x' = F x, P' = F P F^T (+Q by the caller). NV = number of position components that carry a velocity.
template <int D, int NV>
__device__ __forceinline__ void motion(St<D>& s) {
  constexpr int O = D - NV;  // velocity of component i sits at i + O (7-state: 4, 8-state: 4)
#pragma unroll
  for (int i = 0; i < NV; ++i) s.m[i] = s.m[i] + s.m[i + O];
  float A[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = (i < NV) ? (s.P[i][j] + s.P[i + O][j]) : s.P[i][j];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) s.P[i][j] = (j < NV) ? (A[i][j] + A[i][j + O]) : A[i][j];
}

// Unblocked lower Cholesky of a 4x4 (diagonal, then the column below it; sums first, one subtraction).
__device__ __forceinline__ bool chol4(float A[4][4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x = A[k][k];
    if (k > 0) {
      float s = A[k][0] * A[k][0];
#pragma unroll
      for (int j = 1; j < k; ++j) s += A[k][j] * A[k][j];
      x -= s;
    }
    if (!(x > 0.0f)) return false;
    x = sqrtf(x);
    A[k][k] = x;
#pragma unroll
    for (int i = k + 1; i < 4; ++i) {
      float t = A[i][k];
      if (k > 0) {
        float s = A[i][0] * A[k][0];
#pragma unroll
        for (int j = 1; j < k; ++j) s += A[i][j] * A[k][j];
        t -= s;
      }
      A[i][k] = t / x;
    }
  }
  return true;
}
// (L L^T) z = b in place: forward column-axpy, backward row-dot.
__device__ __forceinline__ void chol4_solve(const float L[4][4], float b[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] /= L[i][i];
#pragma unroll
    for (int r = i + 1; r < 4; ++r) b[r] -= b[i] * L[r][i];
  }
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    if (i < 3) {
      float s = L[i + 1][i] * b[i + 1];
#pragma unroll
      for (int j = i + 2; j < 4; ++j) s += L[j][i] * b[j];
      b[i] -= s;
    }
    b[i] /= L[i][i];
  }
}
using mot::kfs::inv_lu4;  // kf_small.hpp: partial-pivot LU inverse of a 4x4 (XYWH's S.inverse(), xywh_kf.hpp:124)

__device__ __forceinline__ float dot4(const float a[4], const float b0, const float b1, const float b2, const float b3) {
  float s = a[0] * b0;
  s += a[1] * b1;
  s += a[2] * b2;
  s += a[3] * b3;
  return s;
}

// ---- XYSR (xysr_kf.cpp) ------------------------------------------------------------------
__device__ __forceinline__ void xysr_init(St<7>& s, const float z[4]) {
#pragma unroll
  for (int i = 0; i < 7; ++i) s.m[i] = (i < 4) ? z[i] : 0.0f;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) s.P[i][j] = (i == j) ? ((i < 4) ? 10.0f : 10.0f * 100.0f) : 0.0f;
}
__device__ __forceinline__ void xysr_predict(St<7>& s, const float q[3]) {
  motion<7, 3>(s);
#pragma unroll
  for (int i = 0; i < 7; ++i) s.P[i][i] = s.P[i][i] + ((i < 4) ? 1.0f : q[i - 4]);
}
__device__ __forceinline__ void xysr_update(St<7>& s, const float z[4]) {
  const float Rd[4] = {1.0f, 1.0f, 10.0f, 10.0f};
  float y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = z[i] - s.m[i];
  float S[4][4], L[4][4], Sinv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { S[i][j] = s.P[i][j] + ((i == j) ? Rd[i] : 0.0f); L[i][j] = S[i][j]; }
  if (chol4(L)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      b[c] = 1.0f;
      chol4_solve(L, b);
#pragma unroll
      for (int i = 0; i < 4; ++i) Sinv[i][c] = b[i];
    }
  } else {
    inv_lu4(S, Sinv);
  }
  float K[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const float pr[4] = {s.P[i][0], s.P[i][1], s.P[i][2], s.P[i][3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) K[i][j] = dot4(pr, Sinv[0][j], Sinv[1][j], Sinv[2][j], Sinv[3][j]);
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) s.m[i] = s.m[i] + dot4(K[i], y[0], y[1], y[2], y[3]);
  // Joseph form P = (I-KH) P (I-KH)^T + K R K^T (xysr_kf.cpp:110-111)
  float G[7][4];  // first four columns of I - K H (the remaining columns are those of I)
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) G[i][j] = ((i == j) ? 1.0f : 0.0f) - K[i][j];
  float M1[7][7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      float v = dot4(G[i], s.P[0][j], s.P[1][j], s.P[2][j], s.P[3][j]);
      if (i >= 4) v += s.P[i][j];
      M1[i][j] = v;
    }
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      float first = dot4(M1[i], G[j][0], G[j][1], G[j][2], G[j][3]);
      if (j >= 4) first += M1[i][j];
      const float kr[4] = {K[i][0] * Rd[0], K[i][1] * Rd[1], K[i][2] * Rd[2], K[i][3] * Rd[3]};
      const float second = dot4(kr, K[j][0], K[j][1], K[j][2], K[j][3]);
      s.P[i][j] = first + second;
    }
}
__device__ __forceinline__ void xysr_box(const St<7>& s, float b[4]) {  // ops.hpp:202-211
  const float w = sqrtf(s.m[2] * s.m[3]);
  const float h = s.m[2] / w;
  b[0] = s.m[0] - w * 0.5f; b[1] = s.m[1] - h * 0.5f; b[2] = s.m[0] + w * 0.5f; b[3] = s.m[1] + h * 0.5f;
}

// ---- XYAH / XYWH (kalman_filter.cpp, xyah_kf.cpp, xywh_kf.hpp) ------------------------------
constexpr float kWp = 1.0f / 20.0f;
constexpr float kWv = 1.0f / 160.0f;

template <int KIND>
__device__ __forceinline__ void s8_init(St<8>& s, const float z[4]) {
  const float h = z[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s.m[i] = (i < 4) ? z[i] : 0.0f;
  float sd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sd[i] = (i < 4) ? 2.0f * kWp * h : 10.0f * kWv * h;
  if (KIND == MOT_KF_XYAH) { sd[2] = 1e-2f; sd[6] = 1e-5f; }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s.P[i][j] = (i == j) ? sd[i] * sd[i] : 0.0f;
}
template <int KIND>
__device__ __forceinline__ void s8_predict(St<8>& s) {
  const float h = s.m[3];
  float sd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sd[i] = (i < 4) ? kWp * h : kWv * h;
  if (KIND == MOT_KF_XYAH) { sd[2] = 1e-2f; sd[6] = 1e-5f; }
  motion<8, 4>(s);
#pragma unroll
  for (int i = 0; i < 8; ++i) s.P[i][i] = s.P[i][i] + sd[i] * sd[i];
}
template <int KIND>
__device__ __forceinline__ void s8_update(St<8>& s, const float z[4], float conf = 0.0f) {
  const float h = s.m[3];
  float sd[4] = {kWp * h, kWp * h, kWp * h, kWp * h};
  if (KIND == MOT_KF_XYAH) {
    sd[2] = 1e-1f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sd[i] = sd[i] * (1.0f - conf);  // NSA Kalman: R_k = ((1 - c_k) std)^2 (kalman_filter.cpp:67)
  }
  float S[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) S[i][j] = s.P[i][j] + ((i == j) ? sd[i] * sd[i] : 0.0f);
  float K[8][4];
  if (KIND == MOT_KF_XYAH) {
    float L[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) L[i][j] = S[i][j];
    if (chol4(L)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float b[4] = {s.P[i][0], s.P[i][1], s.P[i][2], s.P[i][3]};
        chol4_solve(L, b);
#pragma unroll
        for (int c = 0; c < 4; ++c) K[i][c] = b[c];
      }
    } else {
      float Sinv[4][4];
      inv_lu4(S, Sinv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pr[4] = {s.P[i][0], s.P[i][1], s.P[i][2], s.P[i][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) K[i][j] = dot4(pr, Sinv[0][j], Sinv[1][j], Sinv[2][j], Sinv[3][j]);
      }
    }
  } else {
    float Sinv[4][4];
    inv_lu4(S, Sinv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float pr[4] = {s.P[i][0], s.P[i][1], s.P[i][2], s.P[i][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) K[i][j] = dot4(pr, Sinv[0][j], Sinv[1][j], Sinv[2][j], Sinv[3][j]);
    }
  }
  float inn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) inn[i] = z[i] - s.m[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s.m[i] = s.m[i] + dot4(K[i], inn[0], inn[1], inn[2], inn[3]);
  float KS[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) KS[i][j] = dot4(K[i], S[0][j], S[1][j], S[2][j], S[3][j]);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s.P[i][j] = s.P[i][j] - dot4(KS[i], K[j][0], K[j][1], K[j][2], K[j][3]);
}
template <int KIND>
__device__ __forceinline__ void s8_box(const St<8>& s, float b[4]) {
  float w = s.m[2];
  const float h = s.m[3];
  if (KIND == MOT_KF_XYAH) w = s.m[2] * s.m[3];  // xyah2xywh, ops.hpp:110-114
  b[0] = s.m[0] - w * 0.5f; b[1] = s.m[1] - h * 0.5f; b[2] = s.m[0] + w * 0.5f; b[3] = s.m[1] + h * 0.5f;
}

// ---- camera-motion compensation with a caller-supplied warp W (3x3 row-major) --------------------
// BotSTrack::multi_gmc, botsort.cpp:60-91: the corners of the state's box through W, then back to cx,cy,w,h.
__device__ __forceinline__ void s8_warp_xywh(St<8>& s, const float* W) {
  float b[4];
  s8_box<MOT_KF_XYWH>(s, b);
  float p[2][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) p[c][r] = W[r * 3] * b[2 * c] + W[r * 3 + 1] * b[2 * c + 1] + W[r * 3 + 2] * 1.0f;
  const float x1 = p[0][0] / p[0][2], y1 = p[0][1] / p[0][2];
  const float x2 = p[1][0] / p[1][2], y2 = p[1][1] / p[1][2];
  const float w = x2 - x1, h = y2 - y1;
  s.m[0] = x1 + w / 2.0f; s.m[1] = y1 + h / 2.0f; s.m[2] = w; s.m[3] = h;
}
// KalmanFilterXYSR::apply_affine_correction, xysr_kf.cpp:114-141: m = W[0:2,0:2], t = W[0:2,2].
__device__ __forceinline__ void warp_block2(St<7>& s, const float m[2][2], int r0, int c0) {  // P[r0:,c0:] = m * P[r0:,c0:] * m^T
  float A[2][2], T[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) A[i][j] = s.P[r0 + i][c0 + j];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) T[i][j] = m[i][0] * A[0][j] + m[i][1] * A[1][j];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s.P[r0 + i][c0 + j] = T[i][0] * m[j][0] + T[i][1] * m[j][1];
}
__device__ __forceinline__ void xysr_warp(St<7>& s, const float* W) {
  const float m[2][2] = {{W[0], W[1]}, {W[3], W[4]}};
  const float cx = s.m[0], cy = s.m[1], vx = s.m[4], vy = s.m[5];
  s.m[0] = (m[0][0] * cx + m[0][1] * cy) + W[2];
  s.m[1] = (m[1][0] * cx + m[1][1] * cy) + W[5];
  s.m[4] = m[0][0] * vx + m[0][1] * vy;
  s.m[5] = m[1][0] * vx + m[1][1] * vy;
  warp_block2(s, m, 0, 0);
  warp_block2(s, m, 4, 4);
  warp_block2(s, m, 0, 4);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s.P[4 + i][j] = s.P[j][4 + i];
}
template <int KIND, int D>
__device__ __forceinline__ void state_warp(St<D>& s, const float* W) {
  if constexpr (KIND == MOT_KF_XYSR) xysr_warp(s, W);
  else if constexpr (KIND == MOT_KF_XYWH) s8_warp_xywh(s, W);
}

template <int KIND> struct Dim { static constexpr int D = 8; };
template <> struct Dim<MOT_KF_XYSR> { static constexpr int D = 7; };

enum { OP_INIT = 0, OP_PREDICT = 1, OP_UPDATE = 2, OP_BOXES = 3, OP_WARP = 4, OP_PREDICT_WARP = 5, OP_PREDICT_BOXES = 6 };

template <int KIND, int OP>
__global__ void __launch_bounds__(kThreads) kf_kernel(const mot_kf_task* __restrict__ tasks) {
  constexpr int D = Dim<KIND>::D;
  __shared__ __attribute__((aligned(16))) float tile[64 * tile_stride<D>()];
  const mot_kf_task T = tasks[blockIdx.y];
  // (a launch is sized from a bound on the items of a task; where the bound is loose - initiations: a few new tracks against the frame's
  // detection count - the grid is capped and a workgroup strides over the chunks, see kf_update8_kernel)
  for (int chunk = blockIdx.x; chunk * kThreads < T.n; chunk += gridDim.x) {
  int lane = threadIdx.x;
  asm volatile("" : "+v"(lane));  // (nothing derived from the lane index stays in registers across the loop)
  const int i = chunk * kThreads + lane;
  const bool active = i < T.n;
  const int src = active ? (T.src ? T.src[i] : i) : -1;
  const int dst = active ? (T.dst ? T.dst[i] : src) : -1;
  St<D> s;
  bool no_store = false;
  float z[4] = {0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float zc = 0.0f;
  if ((OP == OP_INIT || OP == OP_UPDATE) && active) {
    const int c = T.midx ? T.midx[i] : i;
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = T.meas[static_cast<size_t>(k) * T.ldm + c];
    if (OP == OP_UPDATE && KIND == MOT_KF_XYAH && T.conf) zc = T.conf[c];
  }
  if (OP == OP_BOXES) {
    if (active) {  // only the first four mean components are needed: one 16-byte load per track
      const float4 m4 = *reinterpret_cast<const float4*>(T.mean + static_cast<size_t>(src) * rec_floats<D>());
      s.m[0] = m4.x; s.m[1] = m4.y; s.m[2] = m4.z; s.m[3] = m4.w;
    }
  } else if (OP == OP_PREDICT_BOXES) {
    // Boxes of the predicted states, nothing stored: the box is a function of the predicted MEAN alone (x' = F x, the same
    // additions as motion<>), so only the first 32 bytes of a record are read — not the covariance.
    if (active) {
      const float4* mp = reinterpret_cast<const float4*>(T.mean + static_cast<size_t>(src) * rec_floats<D>());
      const float4 a = mp[0], b = mp[1];
      float m[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const unsigned f = T.flags ? T.flags[i] : 0u;
      if constexpr (KIND == MOT_KF_XYSR) {
        if ((f & MOT_KF_OCSORT_CLAMP) && (m[6] + m[2]) <= 0.0f) m[6] = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) m[k] = m[k] + m[k + 4];
      } else {
        if (f & MOT_KF_ZERO_V7) m[7] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = m[k] + m[k + 4];
      }
#pragma unroll
      for (int k = 0; k < D; ++k) s.m[k] = m[k];
    }
  } else {
    if (OP != OP_INIT) {
      move_records<D, true>(T.mean, tile, src, lane);
      __syncthreads();
      if (active) tile_to_state<D>(s, tile, lane);
      __syncthreads();
    }
    if (active) {
      if (OP == OP_INIT) {
        if constexpr (KIND == MOT_KF_XYSR) xysr_init(s, z); else s8_init<KIND>(s, z);
      } else if (OP == OP_WARP) {
        state_warp<KIND, D>(s, T.warp);
      } else {
        const unsigned f = T.flags ? T.flags[i] : 0u;
        if (OP == OP_PREDICT || OP == OP_PREDICT_WARP) {
          if constexpr (KIND == MOT_KF_XYSR) {
            if ((f & MOT_KF_OCSORT_CLAMP) && (s.m[6] + s.m[2]) <= 0.0f) s.m[6] = 0.0f;
            xysr_predict(s, T.q);
          } else {
            if (f & MOT_KF_ZERO_V7) s.m[7] = 0.0f;
            s8_predict<KIND>(s);
          }
          if (OP == OP_PREDICT_WARP) state_warp<KIND, D>(s, T.warp);
        } else {
          if (f & MOT_KF_PREDICT_FIRST) {
            if constexpr (KIND == MOT_KF_XYSR) {
              if ((f & MOT_KF_OCSORT_CLAMP) && (s.m[6] + s.m[2]) <= 0.0f) s.m[6] = 0.0f;
              xysr_predict(s, T.q);
            } else {
              if (f & MOT_KF_ZERO_V7) s.m[7] = 0.0f;
              s8_predict<KIND>(s);
            }
          }
          if constexpr (KIND == MOT_KF_XYSR) xysr_update(s, z); else s8_update<KIND>(s, z, zc);
        }
        no_store = (OP == OP_PREDICT || OP == OP_PREDICT_WARP) && (f & MOT_KF_NO_STORE);
      }
    }
    const int out_slot = (active && !no_store) ? dst : -1;
    if (__builtin_amdgcn_ballot_w64(out_slot >= 0) != 0) {  // (box-only predictions write nothing back)
      if (out_slot >= 0) state_to_tile<D>(s, tile, lane);
      __syncthreads();
      move_records<D, false>(T.mean, tile, out_slot, lane);
    }
  }
  if (T.boxes && active) {
    float b[4];
    if constexpr (KIND == MOT_KF_XYSR) xysr_box(s, b); else s8_box<KIND>(s, b);
    if constexpr (KIND == MOT_KF_XYAH) {
      if (T.reserved == MOT_KF_BOX_TLWH_SUM) {  // StrongSORT's Track::to_tlbr (strongsort.cpp:94-111): the far corner is top-left + size
        const float w = s.m[2] * s.m[3];
        b[2] = b[0] + w; b[3] = b[1] + s.m[3];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) T.boxes[static_cast<size_t>(k) * T.ldb + i] = b[k];
  }
  if ((chunk + static_cast<int>(gridDim.x)) * kThreads < T.n) __syncthreads();  // (another chunk follows: it reuses the tile)
  }
}

// ---- detections ------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(kThreads) det_kernel(const mot_det_task* __restrict__ tasks) {
  const mot_det_task T = tasks[blockIdx.y];
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= T.n) return;
  const float x1 = T.dets[i], y1 = T.dets[static_cast<size_t>(T.ld) + i];
  const float x2 = T.dets[static_cast<size_t>(2) * T.ld + i], y2 = T.dets[static_cast<size_t>(3) * T.ld + i];
  float box[4], z[4];
  if (KIND == MOT_DET_XYSR) {  // ops.hpp:188-197
    const float w = x2 - x1, h = y2 - y1;
    z[0] = x1 + w * 0.5f; z[1] = y1 + h * 0.5f; z[2] = w * h; z[3] = (h > 1e-6f) ? (w / h) : 0.0f;
    box[0] = x1; box[1] = y1; box[2] = x2; box[3] = y2;
  } else if (KIND == MOT_DET_XYAH) {  // bytetrack.cpp:29-33: xyxy2xywh -> xywh2tlwh -> tlwh2xyah
    const float w = x2 - x1, h = y2 - y1;
    const float xc = x1 + w * 0.5f, yc = y1 + h * 0.5f;
    const float tl = xc - w * 0.5f, tt = yc - h * 0.5f;
    z[0] = tl + w * 0.5f; z[1] = tt + h * 0.5f; z[2] = (h > 0.0f) ? (w / h) : 0.0f; z[3] = h;
    box[0] = xc - w * 0.5f; box[1] = yc - h * 0.5f; box[2] = xc + w * 0.5f; box[3] = yc + h * 0.5f;  // xywh2xyxy
  } else if (KIND == MOT_DET_TLWH) {  // strongsort.cpp:948-956 (tlwh), Detection::to_xyah :33-40
    const float w = x2 - x1, h = y2 - y1;
    box[0] = x1; box[1] = y1; box[2] = w; box[3] = h;
    z[0] = x1 + w / 2.0f; z[1] = y1 + h / 2.0f; z[2] = w / h; z[3] = h;
  } else {  // botsort.cpp:23-36, 171-181
    const float w = x2 - x1, h = y2 - y1;
    const float cx = x1 + w / 2.0f, cy = y1 + h / 2.0f;
    z[0] = cx; z[1] = cy; z[2] = w; z[3] = h;
    box[0] = cx - w / 2; box[1] = cy - h / 2; box[2] = cx + w / 2; box[3] = cy + h / 2;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (T.box) T.box[static_cast<size_t>(k) * T.ldb + i] = box[k];
    if (T.meas) T.meas[static_cast<size_t>(k) * T.ldm + i] = z[k];
  }
}

// ---- 8-state update, one lane per covariance ROW -------------------------------------------------------------------
// The lane-per-track kernel above keeps a whole 8x8 state (and the update's temporaries) in one lane: 242 VGPRs, two
// wavefronts per SIMD, and a 64-track wavefront waits on its 27 memory instructions with nothing else to run (measured
// 12 % of the HBM roof). Here a 256-thread workgroup owns 32 tracks at a time and lane (track, r) owns row r of P, K and K S.
// The factor of the 4x4 innovation covariance is computed once per track (one lane each, the first 32 lanes of the workgroup)
// and shared through LDS; the gain rows meet once in LDS for P - (K S) K^T.
// Every element is produced by the same operations in the same order as s8_predict / s8_update: results are bit-identical.
// Measured (tools/kf_update_microbench.py, 1 M gathered records): 3.3-3.4 TB/s of record bytes; the same loads and stores with
// the arithmetic removed reach 4.0 (gathered) - 4.7 TB/s (contiguous), which is what this access pattern (288-byte records
// read and rewritten in place through LDS) can reach; a float4 copy reaches 6.3.
constexpr int kUpdTracks = 32;
constexpr int kUpdGroups = 16;  // workgroups per task at most (they stride over the task's tiles)

template <int KIND, int TPB>
__global__ void __launch_bounds__(TPB * 8) __attribute__((amdgpu_waves_per_eu(KIND == MOT_KF_XYAH ? 6 : 8, KIND == MOT_KF_XYAH ? 6 : 8))) kf_update8_kernel(const mot_kf_task* __restrict__ tasks) {
  constexpr int kUpdTracks = TPB, NT = TPB * 8;
  constexpr int RS = tile_stride<8>();  // 76 floats = 19 float4
  __shared__ __attribute__((aligned(16))) float tile[kUpdTracks * RS];
  __shared__ __attribute__((aligned(16))) float kbuf[kUpdTracks * 36];  // (36: the eight tracks of a wavefront read their gain rows from different banks)
  __shared__ __attribute__((aligned(16))) float fbuf[kUpdTracks * 20];  // per track: the 4x4 factor (or inverse) of S, [16] = which
  __shared__ float zbuf[kUpdTracks];
  __shared__ int s_src[kUpdTracks], s_dst[kUpdTracks];
  const mot_kf_task T = tasks[blockIdx.y];
  // The launch is sized from a BOUND on the items of a task (the host does not know the count the previous kernel left on the device).
  // A workgroup that finds nothing to do still waits its turn for 17 KB of LDS and four wavefronts' registers: with a bound twice the
  // count, 0.94 M updates took 0.27 ms instead of 0.16. So a task gets at most kUpdGroups workgroups, which stride over its tiles.
  for (int base = blockIdx.x * kUpdTracks; base < T.n; base += gridDim.x * kUpdTracks) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // (what derives from the lane index is recomputed per tile instead of staying in registers across the loop)
  const int tr = tid >> 3, r = tid & 7;
  const int item = base + tr;
  const bool active = item < T.n;
  // the measurement of the lane's track: requested before the records so that its two dependent loads (index, then value) overlap theirs
  float z[4] = {0.f, 0.f, 0.f, 0.f};
  float zc = 0.0f;
  unsigned f = 0u;
  if (active) {
    const int c = T.midx ? T.midx[item] : item;
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = T.meas[static_cast<size_t>(k) * T.ldm + c];
    f = T.flags ? T.flags[item] : 0u;
    if (KIND == MOT_KF_XYAH && T.conf) zc = T.conf[c];
  }
  if (tid < kUpdTracks) {
    const int i = base + tid;
    const bool a = i < T.n;
    const int src = a ? (T.src ? T.src[i] : i) : -1;
    s_src[tid] = src;
    s_dst[tid] = a ? (T.dst ? T.dst[i] : src) : -1;
  }
  __syncthreads();
  {
    const float4* slab4 = reinterpret_cast<const float4*>(T.mean);
    float4* tile4 = reinterpret_cast<float4*>(tile);
#pragma unroll
    for (int p = tid; p < kUpdTracks * 18; p += NT) {
      const int rc = p / 18, q = p - rc * 18;
      const int slot = s_src[rc];
      // (a task with a dense mirror of the means keeps them THERE: the record's first 32 bytes are neither read nor written)
      // (a task with dense means keeps them THERE, and its slab holds covariance-only records of 64 floats: 256 bytes on a 256-byte boundary
      // are two lines of 128 bytes where a 288-byte record at any multiple of 288 touches 3.25 on average)
      if (slot >= 0) tile4[rc * 19 + q] = !T.mean_dense ? slab4[static_cast<size_t>(slot) * 18 + q]
                                          : (q < 2 ? reinterpret_cast<const float4*>(T.mean_dense)[static_cast<size_t>(slot) * 2 + q]
                                                   : slab4[static_cast<size_t>(slot) * 16 + (q - 2)]);
    }
  }
  __syncthreads();
  float* rec = tile + tr * RS;
  float P[8];
  {
    const float4 a = *reinterpret_cast<const float4*>(rec + 8 + r * 8), b = *reinterpret_cast<const float4*>(rec + 12 + r * 8);
    P[0] = a.x; P[1] = a.y; P[2] = a.z; P[3] = a.w; P[4] = b.x; P[5] = b.y; P[6] = b.z; P[7] = b.w;
  }
  float m = rec[r];
  if (f & MOT_KF_PREDICT_FIRST) {  // s8_predict: x' = F x, P' = F P F^T + Q with h taken before the motion step
    const bool zero_v7 = (f & MOT_KF_ZERO_V7) != 0;
    const float h = rec[3];
    if (zero_v7 && r == 7) m = 0.0f;
    if (r < 4) {
      float mh = rec[r + 4];
      if (zero_v7 && r == 3) mh = 0.0f;
      m = m + mh;
      const float4 a = *reinterpret_cast<const float4*>(rec + 8 + (r + 4) * 8), b = *reinterpret_cast<const float4*>(rec + 12 + (r + 4) * 8);
      const float hi[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) P[j] = P[j] + hi[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) P[j] = P[j] + P[j + 4];
    float sd = (r < 4) ? kWp * h : kWv * h;
    if (KIND == MOT_KF_XYAH) { if (r == 2) sd = 1e-2f; if (r == 6) sd = 1e-5f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == r) P[j] = P[j] + sd * sd;
  }
  __syncthreads();  // every lane has read the stored rows it needs
  rec[r] = m;
  *reinterpret_cast<float4*>(rec + 8 + r * 8) = make_float4(P[0], P[1], P[2], P[3]);
  *reinterpret_cast<float4*>(rec + 12 + r * 8) = make_float4(P[4], P[5], P[6], P[7]);
  __syncthreads();
  // s8_update on the (predicted) state in the tile
  const float h = rec[3];
  float sd[4] = {kWp * h, kWp * h, kWp * h, kWp * h};
  if (KIND == MOT_KF_XYAH) {
    sd[2] = 1e-1f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sd[i] = sd[i] * (1.0f - zc);  // NSA Kalman (kalman_filter.cpp:67); 0 unless the task carries confidences
  }
  // S = H P H^T + R is read from the tile where it is needed (the factor, K S) instead of staying in registers.
  auto load_S = [&](const float* rc, const float sdv[4], float S[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(rc + 8 + i * 8);
      S[i][0] = a.x; S[i][1] = a.y; S[i][2] = a.z; S[i][3] = a.w;
      S[i][i] = S[i][i] + sdv[i] * sdv[i];
    }
  };
  // The factor of S (XYAH: Cholesky; XYWH, or a failed Cholesky: the pivoted-LU inverse) is the same for the eight rows of a track and
  // is a serial chain of square roots and divisions: ONE lane per track computes it (the first TPB lanes of the workgroup, a track each)
  // and leaves it in LDS, instead of every row lane repeating it - a quarter of the kernel's VALU instructions otherwise.
  if (r == 0) zbuf[tr] = zc;
  __syncthreads();
  if (tid < kUpdTracks && base + tid < T.n) {
    const float* rc = tile + tid * RS;
    const float hA = rc[3];
    float sdA[4] = {kWp * hA, kWp * hA, kWp * hA, kWp * hA};
    if (KIND == MOT_KF_XYAH) {
      const float zcA = zbuf[tid];
      sdA[2] = 1e-1f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sdA[i] = sdA[i] * (1.0f - zcA);
    }
    float* fb = fbuf + tid * 20;
    bool ok = false;
    if (KIND == MOT_KF_XYAH) {
      float L[4][4];
      load_S(rc, sdA, L);
      ok = chol4(L);
      if (ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(fb + 4 * i) = make_float4(L[i][0], L[i][1], L[i][2], L[i][3]);
      }
    }
    if (!ok) {
      float S[4][4], Sinv[4][4];
      load_S(rc, sdA, S);
      inv_lu4(S, Sinv);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(fb + 4 * i) = make_float4(Sinv[i][0], Sinv[i][1], Sinv[i][2], Sinv[i][3]);
    }
    fb[16] = ok ? 1.0f : 0.0f;
  }
  __syncthreads();
  float K[4];
  {
    const float* fb = fbuf + tr * 20;
    float F[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(fb + 4 * i);
      F[i][0] = a.x; F[i][1] = a.y; F[i][2] = a.z; F[i][3] = a.w;
    }
    if (KIND == MOT_KF_XYAH && fb[16] != 0.0f) {
      K[0] = P[0]; K[1] = P[1]; K[2] = P[2]; K[3] = P[3];
      chol4_solve(F, K);
    } else {
      const float pr[4] = {P[0], P[1], P[2], P[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) K[j] = dot4(pr, F[0][j], F[1][j], F[2][j], F[3][j]);
    }
  }
  float inn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) inn[i] = z[i] - rec[i];
  const float m_new = m + dot4(K, inn[0], inn[1], inn[2], inn[3]);
  float KS[4];
  {
    float S[4][4];
    load_S(rec, sd, S);
#pragma unroll
    for (int j = 0; j < 4; ++j) KS[j] = dot4(K, S[0][j], S[1][j], S[2][j], S[3][j]);
  }
  *reinterpret_cast<float4*>(kbuf + tr * 36 + r * 4) = make_float4(K[0], K[1], K[2], K[3]);
  __syncthreads();  // gains published; every lane is done reading S and the mean from the tile
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 kj = *reinterpret_cast<const float4*>(kbuf + tr * 36 + j * 4);
    P[j] = P[j] - dot4(KS, kj.x, kj.y, kj.z, kj.w);
  }
  rec[r] = m_new;
  *reinterpret_cast<float4*>(rec + 8 + r * 8) = make_float4(P[0], P[1], P[2], P[3]);
  *reinterpret_cast<float4*>(rec + 12 + r * 8) = make_float4(P[4], P[5], P[6], P[7]);
  __syncthreads();
  {
    float4* slab4 = reinterpret_cast<float4*>(T.mean);
    const float4* tile4 = reinterpret_cast<const float4*>(tile);
#pragma unroll
    for (int p = tid; p < kUpdTracks * 18; p += NT) {
      const int rc = p / 18, q = p - rc * 18;
      const int slot = s_dst[rc];
      if (slot >= 0) {
        const float4 val = tile4[rc * 19 + q];
        if (!T.mean_dense) slab4[static_cast<size_t>(slot) * 18 + q] = val;
        else if (q < 2) reinterpret_cast<float4*>(T.mean_dense)[static_cast<size_t>(slot) * 2 + q] = val;  // (the dense means, motcpp_amd.h)
        else slab4[static_cast<size_t>(slot) * 16 + (q - 2)] = val;
      }
    }
  }
  if (T.boxes && active && r < 4) {  // xyxy of the written state: lane r writes component r
    float w = rec[2];
    const float hh = rec[3];
    if (KIND == MOT_KF_XYAH) w = rec[2] * rec[3];
    const float b = (r == 0) ? rec[0] - w * 0.5f : (r == 1) ? rec[1] - hh * 0.5f : (r == 2) ? rec[0] + w * 0.5f : rec[1] + hh * 0.5f;
    T.boxes[static_cast<size_t>(r) * T.ldb + item] = b;
  }
  __syncthreads();  // the tile, the gains, the factors and the slot lists are free for the next tile
  }
}

// ---- 8-state XYAH update on BLOCK-STRUCTURED covariances (round 6) -----------------------------------------------------------------
// A track of these filters is born with a diagonal covariance (initiate) and from then on only predicted (F = I + shift: component c takes
// its velocity c + 4) and updated (H = [I 0], R diagonal): by induction P(i, j) is an exact zero unless j = i (mod 4), the innovation
// covariance S = H P H^T + R is DIAGONAL, its Cholesky factor is diag(sqrt(S_cc)), a gain row has one non-zero entry, and the whole update is
// four independent two-state filters (component, velocity). Every product the dense code forms with a structural zero is an exact zero
// as long as the other factor is FINITE, and adding an exact zero changes nothing (up to the sign of a zero, which no comparison sees and
// no later operation turns into a non-zero): the four blocks below are the non-zero terms of s8_predict / s8_update / kf_update8_kernel in
// the same order — K = (p / l) / l with l = sqrt(S) (chol4_solve's two substitutions), KS = K * S, P -= KS * K — bit for bit. A track that
// meets a non-positive or non-finite S, or any non-finite input or intermediate (0 * inf would put NaN where the blocks keep a zero), is NOT
// touched here: its blocks are expanded into its 64-float record, its flag is set, and it goes — this frame and from now on — through
// kf_update8_kernel (the `fallback` task of the same stream, filled here). 96 bytes in and out per track instead of 576, four lanes per track.
constexpr int kBlkTracks = 64;
template <int KIND, int kBlkItems>  // kBlkItems: tracks per quad of lanes and turn
__global__ void __launch_bounds__(kBlkTracks * 4) kf_update_blocks_kernel(const mot_kf_task* __restrict__ tasks, mot_kf_task* __restrict__ fallback) {
  static_assert(KIND == MOT_KF_XYAH, "block form: the XYAH filter (ByteTrack)");
  // grid = (tasks, turns): consecutive workgroups — which the dispatcher deals round-robin to the eight XCDs — are the same turn of consecutive streams.
  // With the turns in x, a stream's four busy workgroups of eight always landed on XCDs 0-3 and the four that exit at once on XCDs 4-7: with the GPU to
  // itself the kernel took 0.39 ms instead of 0.22
  const mot_kf_task T = tasks[blockIdx.x];
  mot_kf_task& F = fallback[blockIdx.x];
  const int tid = threadIdx.x, c = tid & 3;
  const bool fast = T.src && T.dst && T.midx && T.flags && T.meas4 && !T.conf;  // (uniform)
  // A workgroup's turn is kBlkItems tracks per quad of lanes, their loads issued level by level (list entries; then mean, blocks, measurement): the
  // kernel is two dependent round trips and a store per track, and with one track per quad the CU's 32 wavefronts kept too few of them in flight
  // (0.33 of HBM; the destination index, read only when the result was ready, was a third trip in front of the stores)
  for (int base0 = blockIdx.y * kBlkTracks * kBlkItems; base0 < T.n; base0 += gridDim.y * kBlkTracks * kBlkItems) {
    int slot_u[kBlkItems], mi_u[kBlkItems], ds_u[kBlkItems];
    unsigned f_u[kBlkItems];
    float z_u[kBlkItems], zc_u[kBlkItems], m0_u[kBlkItems], m1_u[kBlkItems];
    float4 blk_u[kBlkItems];
    bool dense_u[kBlkItems];
    if (fast) {
      // the device lifecycle's task: every list is there, measurements as [n][4], no confidences. Unconditional loads through global pointers, an entry
      // past the end reads the last one and is masked afterwards — the generic form below compiles to a branch and a full wait per optional pointer
      // (ten dependent groups of flat loads where two levels are meant)
      typedef const int32_t __attribute__((address_space(1))) * gint;
      typedef const float __attribute__((address_space(1))) * gfloat;
      typedef float __attribute__((ext_vector_type(4))) f4v;
      typedef const f4v __attribute__((address_space(1))) * gfloat4;
      typedef const unsigned char __attribute__((address_space(1))) * gbyte;
      const gint src = (gint)T.src, dst = (gint)T.dst, midx = (gint)T.midx;
      const gbyte flg = (gbyte)T.flags, dfl = (gbyte)T.dense_flag;
      const gfloat md = (gfloat)T.mean_dense, m4 = (gfloat)T.meas4;
      const gfloat4 cb = (gfloat4)T.cov_blocks;
#pragma unroll
      for (int u = 0; u < kBlkItems; ++u) {
        const int item = base0 + u * kBlkTracks + (tid >> 2);
        const int it = (item < T.n) ? item : T.n - 1;
        slot_u[u] = src[it]; mi_u[u] = midx[it]; f_u[u] = flg[it]; ds_u[u] = dst[it];
      }
#pragma unroll
      for (int u = 0; u < kBlkItems; ++u) {
        const size_t slot = static_cast<size_t>(slot_u[u]);
        z_u[u] = m4[static_cast<size_t>(mi_u[u]) * 4 + c];
        zc_u[u] = 0.0f;
        m0_u[u] = md[slot * 8 + c];
        m1_u[u] = md[slot * 8 + c + 4];
        const f4v bv = cb[slot * 4 + c];
        blk_u[u] = make_float4(bv.x, bv.y, bv.z, bv.w);
        dense_u[u] = dfl[slot] != 0;
      }
#pragma unroll
      for (int u = 0; u < kBlkItems; ++u) {
        const bool active = base0 + u * kBlkTracks + (tid >> 2) < T.n;
        if (!active) { slot_u[u] = 0; mi_u[u] = 0; f_u[u] = 0u; z_u[u] = 0.0f; m0_u[u] = 0.0f; m1_u[u] = 0.0f; blk_u[u] = make_float4(0.f, 0.f, 0.f, 0.f); dense_u[u] = false; }
      }
    } else {
#pragma unroll
    for (int u = 0; u < kBlkItems; ++u) {
      const int item = base0 + u * kBlkTracks + (tid >> 2);
      const bool active = item < T.n;
      slot_u[u] = active ? (T.src ? T.src[item] : item) : 0;
      mi_u[u] = active ? (T.midx ? T.midx[item] : item) : 0;
      f_u[u] = (active && T.flags) ? T.flags[item] : 0u;
      ds_u[u] = (active && T.dst) ? T.dst[item] : -1;
    }
#pragma unroll
    for (int u = 0; u < kBlkItems; ++u) {
      const int item = base0 + u * kBlkTracks + (tid >> 2);
      const bool active = item < T.n;
      const int slot = slot_u[u], mi = mi_u[u];
      z_u[u] = active ? (T.meas4 ? T.meas4[static_cast<size_t>(mi) * 4 + c] : T.meas[static_cast<size_t>(c) * T.ldm + mi]) : 0.0f;
      zc_u[u] = (active && T.conf) ? T.conf[mi] : 0.0f;
      m0_u[u] = active ? T.mean_dense[static_cast<size_t>(slot) * 8 + c] : 0.0f;
      m1_u[u] = active ? T.mean_dense[static_cast<size_t>(slot) * 8 + c + 4] : 0.0f;
      blk_u[u] = active ? reinterpret_cast<const float4*>(T.cov_blocks)[static_cast<size_t>(slot) * 4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
      dense_u[u] = active && T.dense_flag[slot] != 0;
    }
    }
#pragma unroll
    for (int u = 0; u < kBlkItems; ++u) {
    const int item = base0 + u * kBlkTracks + (tid >> 2);
    if (base0 + u * kBlkTracks >= T.n) break;  // (uniform)
    const bool active = item < T.n;
    const int slot = slot_u[u], mi = mi_u[u];
    const unsigned f = f_u[u];
    const float z = z_u[u], zc = zc_u[u];
    float m0 = m0_u[u], m1 = m1_u[u];
    const float4 blk = blk_u[u];
    const bool dense = dense_u[u];
    float a = blk.x, b = blk.y, bp = blk.z, d = blk.w;
    const int l3 = (tid & 63 & ~3) | 3;  // the lane of this track's component 3 (the height)
    if (f & MOT_KF_PREDICT_FIRST) {  // s8_predict: x' = F x, P' = F P F^T + Q, the standard deviations from the height BEFORE the motion step
      const bool zero_v7 = (f & MOT_KF_ZERO_V7) != 0;
      const float h = __shfl(m0, l3, 64);
      if (zero_v7 && c == 3) m1 = 0.0f;
      m0 = m0 + m1;
      a = a + bp; b = b + d;    // row c += row c + 4
      a = a + b; bp = bp + d;   // column c += column c + 4
      float sp = kWp * h, sv = kWv * h;
      if (c == 2) { sp = 1e-2f; sv = 1e-5f; }
      a = a + sp * sp; d = d + sv * sv;
    }
    const float h = __shfl(m0, l3, 64);
    float sdm = kWp * h;
    if (c == 2) sdm = 1e-1f;
    sdm = sdm * (1.0f - zc);  // NSA Kalman (kalman_filter.cpp:67); zc = 0 unless the task carries confidences
    const float S = a + sdm * sdm;
    const float l = sqrtf(S);
    const float Kt = (a / l) / l, Kb = (bp / l) / l;
    const float inn = z - m0;
    const float m0n = m0 + Kt * inn, m1n = m1 + Kb * inn;
    const float KSt = Kt * S, KSb = Kb * S;
    const float an = a - KSt * Kt, bn = b - KSt * Kb, bpn = bp - KSb * Kt, dn = d - KSb * Kb;
    auto fin = [](float x) { return fabsf(x) < 3.0e38f; };  // (false for NaN and inf)
    bool ok = S > 0.0f && fin(S) && fin(a) && fin(b) && fin(bp) && fin(d) && fin(m0) && fin(m1) && fin(inn) && fin(Kt) && fin(Kb) && fin(KSt) && fin(KSb) &&
              fin(blk.x) && fin(blk.y) && fin(blk.z) && fin(blk.w) && fin(z);
    ok = ok && !dense;
    // all four blocks of the track, or none
    const unsigned long long bal = __ballot(ok || !active);
    const int lane = tid & 63;
    const bool track_ok = ((bal >> (lane & ~3)) & 0xfull) == 0xfull;
    if (active && track_ok) {
      const int ds = T.dst ? ds_u[u] : slot;
      T.mean_dense[static_cast<size_t>(ds) * 8 + c] = m0n;
      T.mean_dense[static_cast<size_t>(ds) * 8 + c + 4] = m1n;
      reinterpret_cast<float4*>(T.cov_blocks)[static_cast<size_t>(ds) * 4 + c] = make_float4(an, bn, bpn, dn);
    } else if (active) {
      if (!dense) {  // the blocks as they were loaded become the track's 8 x 8 record: lane c writes rows c and c + 4
        float4* rec = reinterpret_cast<float4*>(T.mean) + static_cast<size_t>(slot) * 16;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r0a = z4, r0b = z4, r1a = z4, r1b = z4;  // row c: (c) = a, (c + 4) = b; row c + 4: (c) = b', (c + 4) = d
        if (c == 0) { r0a.x = blk.x; r0b.x = blk.y; r1a.x = blk.z; r1b.x = blk.w; }
        if (c == 1) { r0a.y = blk.x; r0b.y = blk.y; r1a.y = blk.z; r1b.y = blk.w; }
        if (c == 2) { r0a.z = blk.x; r0b.z = blk.y; r1a.z = blk.z; r1b.z = blk.w; }
        if (c == 3) { r0a.w = blk.x; r0b.w = blk.y; r1a.w = blk.z; r1b.w = blk.w; }
        rec[2 * c] = r0a; rec[2 * c + 1] = r0b; rec[2 * (c + 4)] = r1a; rec[2 * (c + 4) + 1] = r1b;
      }
      if (c == 0) {
        T.dense_flag[slot] = 1;
        const int k = atomicAdd(&F.n, 1);
        const_cast<int32_t*>(F.src)[k] = slot;
        const_cast<int32_t*>(F.dst)[k] = T.dst ? ds_u[u] : slot;
        const_cast<int32_t*>(F.midx)[k] = mi;
        const_cast<uint8_t*>(F.flags)[k] = static_cast<uint8_t>(f);
      }
    }
    }
  }
}

// ---- gating: squared distances between the projected states and the frame's measurements --------------------------
// BaseKalmanFilter::gating_distance (kalman_filter.cpp:148-176, the XYAH filter StrongSORT carries) and
// KalmanFilterXYWH::gating_distance (xywh_kf.hpp:140-176), with the two callers' blends fused into the same pass:
// utils::fuse_motion (matching.hpp:60-94) and StrongSORT's gate_cost_matrix (strongsort.cpp:449-492).
// Grid: x = 256-column chunks, y = track row, z = task. A thread owns one measurement; the row's 4x4 innovation covariance
// and its factor are uniform over the block (built from 20 scalar-cached floats of the record, a few dozen flops), so the
// pass reads the cost matrix once and writes the result once: 8 B per pair + 16 B per measurement per row chunk.
constexpr int kGateThreads = 256;

__device__ __forceinline__ bool chol2(float A[2][2]) {  // the same unblocked recurrence on the leading 2x2
  float x = A[0][0];
  if (!(x > 0.0f)) return false;
  x = sqrtf(x);
  A[0][0] = x;
  A[1][0] = A[1][0] / x;
  float y = A[1][1];
  y -= A[1][0] * A[1][0];
  if (!(y > 0.0f)) return false;
  A[1][1] = sqrtf(y);
  return true;
}

template <int KIND>
__global__ void __launch_bounds__(kGateThreads) gate_kernel(const mot_gate_task* __restrict__ tasks) {
  const mot_gate_task T = tasks[blockIdx.z];
  const int row = blockIdx.y;
  const int j = blockIdx.x * kGateThreads + threadIdx.x;
  if (row >= T.n || blockIdx.x * kGateThreads >= T.m) return;
  const int slot = T.src ? T.src[row] : row;
  const float* rec = T.mean + static_cast<size_t>(slot) * rec_floats<8>();
  float pm[4], S[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pm[k] = rec[k];  // H = [I 0]: the projected mean is the first four components
  const float h = pm[3];
  float sd[4] = {kWp * h, kWp * h, kWp * h, kWp * h};
  if (KIND == MOT_KF_XYAH) {
    sd[2] = 1e-1f;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd[k] = sd[k] * (1.0f - 0.0f);  // project() with the default confidence 0 (kalman_filter.cpp:67)
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) S[a][b] = rec[8 + a * 8 + b] + ((a == b) ? sd[a] * sd[a] : 0.0f);
  const bool pos = T.only_position != 0;
  // row-uniform factorisation
  float L4[4][4], L2[2][2], Sinv[4][4];
  bool ok = true;
  if (KIND == MOT_KF_XYAH) {
    if (T.metric == 0) {
      if (pos) {
        L2[0][0] = S[0][0]; L2[0][1] = S[0][1]; L2[1][0] = S[1][0]; L2[1][1] = S[1][1];
        ok = chol2(L2);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) L4[a][b] = S[a][b];
        ok = chol4(L4);
      }
    }
  } else {
    inv_lu4(S, Sinv);
  }
  if (j >= T.m) return;
  float d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = T.meas[static_cast<size_t>(k) * T.ldm + j] - pm[k];
  float g;
  if (KIND == MOT_KF_XYAH) {
    if (T.metric != 0 || !ok) {  // "gaussian", or the LLT failed: plain squared norm (:161-167)
      g = d[0] * d[0];
      g += d[1] * d[1];
      if (!pos) { g += d[2] * d[2]; g += d[3] * d[3]; }
    } else if (pos) {  // z = (L L^T)^-1 d, |z|^2 (:169-170: the solve, not the half-solve)
      float z0 = d[0] / L2[0][0];
      float z1 = d[1] - z0 * L2[1][0];
      z1 = z1 / L2[1][1];
      z1 = z1 / L2[1][1];
      z0 = z0 - L2[1][0] * z1;
      z0 = z0 / L2[0][0];
      g = z0 * z0;
      g += z1 * z1;
    } else {
      chol4_solve(L4, d);
      g = d[0] * d[0];
      g += d[1] * d[1];
      g += d[2] * d[2];
      g += d[3] * d[3];
    }
  } else {  // d^T S^-1 d with the LU inverse; position only: the leading 2x2 block of the 4x4 inverse (:168-174)
    if (pos) {
      float t0 = d[0] * Sinv[0][0]; t0 += d[1] * Sinv[1][0];
      float t1 = d[0] * Sinv[0][1]; t1 += d[1] * Sinv[1][1];
      g = t0 * d[0];
      g += t1 * d[1];
    } else {
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a = d[0] * Sinv[0][c];
        a += d[1] * Sinv[1][c];
        a += d[2] * Sinv[2][c];
        a += d[3] * Sinv[3][c];
        t[c] = a;
      }
      g = t[0] * d[0];
      g += t[1] * d[1];
      g += t[2] * d[2];
      g += t[3] * d[3];
    }
  }
  float out = g;
  const int gmode = T.mode & 0xff;
  if (gmode == MOT_GATE_FUSE_MOTION) {
    const float thr = pos ? 5.9915f : 9.4877f;  // chi2inv95[dim - 1] (matching.hpp:16-26)
    const float c = T.cost[static_cast<size_t>(row) * T.ldc + j];
    out = (g > thr) ? __builtin_inff() : (T.lambda * c + (1.0f - T.lambda) * g);
  } else if (gmode == MOT_GATE_STRONGSORT) {
    float c = T.cost[static_cast<size_t>(row) * T.ldc + j];
    if (g > 9.4877f) c = T.gated_cost;  // the 4-dof quantile whatever only_position is (strongsort.cpp:461)
    out = T.lambda * c + (1.0f - T.lambda) * g;
  }
  if ((T.mode & MOT_GATE_CLAMP) && out > T.clamp_above) out = T.clamp_above + 1e-5f;  // min_cost_matching, strongsort.cpp:376-379
  T.out[static_cast<size_t>(row) * T.ldo + j] = out;
}

template <int OP>
hipError_t launch_kf(int kind, const mot_kf_task* tasks, int ntasks, int max_n, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  const int want = (max_n + kThreads - 1) / kThreads, cap = (OP == OP_INIT) ? 2 : 64;
  dim3 grid((want < cap) ? want : cap, ntasks), block(kThreads);
  switch (kind) {
    case MOT_KF_XYSR: hipLaunchKernelGGL((kf_kernel<MOT_KF_XYSR, OP>), grid, block, 0, st, tasks); break;
    case MOT_KF_XYAH: hipLaunchKernelGGL((kf_kernel<MOT_KF_XYAH, OP>), grid, block, 0, st, tasks); break;
    case MOT_KF_XYWH: hipLaunchKernelGGL((kf_kernel<MOT_KF_XYWH, OP>), grid, block, 0, st, tasks); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

namespace mot {
// ByteTrack's device lifecycle: the block-form update over every task, then the dense update over the tasks' fallback lists (usually empty: one
// workgroup per task looks at a zero count)
hipError_t launch_kf_update_blocks(const mot_kf_task* tasks, mot_kf_task* fallback, int ntasks, int max_n, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  static const int items = [] { const char* e = std::getenv("MOT_KF_BLK_ITEMS"); const int v = (e && *e) ? std::atoi(e) : 4; return (v == 1 || v == 2) ? v : 4; }();
  const int want = (max_n + kBlkTracks * items - 1) / (kBlkTracks * items);
  dim3 grid(ntasks, (want < kUpdGroups) ? want : kUpdGroups), block(kBlkTracks * 4);
  // (MOT_KF_BLK_ITEMS = 1 / 2: the narrower forms, for measurements — north-star frame, same box, same runs: 2.98-3.00 / 3.00-3.04 / 3.08-3.14 M frames/s)
  if (items == 1) hipLaunchKernelGGL((kf_update_blocks_kernel<MOT_KF_XYAH, 1>), grid, block, 0, st, tasks, fallback);
  else if (items == 2) hipLaunchKernelGGL((kf_update_blocks_kernel<MOT_KF_XYAH, 2>), grid, block, 0, st, tasks, fallback);
  else hipLaunchKernelGGL((kf_update_blocks_kernel<MOT_KF_XYAH, 4>), grid, block, 0, st, tasks, fallback);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((kf_update8_kernel<MOT_KF_XYAH, kUpdTracks>), dim3(1, ntasks), dim3(kUpdTracks * 8), 0, st, fallback);
  return hipGetLastError();
}
hipError_t launch_kf_op(int op, int kind, const mot_kf_task* tasks, int ntasks, int max_n, hipStream_t st) {
  switch (op) {
    case OP_INIT: return launch_kf<OP_INIT>(kind, tasks, ntasks, max_n, st);
    case OP_PREDICT: return launch_kf<OP_PREDICT>(kind, tasks, ntasks, max_n, st);
    case OP_UPDATE:
      static const bool lane_per_track = std::getenv("MOT_KF_UPDATE_LANE_PER_TRACK") != nullptr;  // measurement aid (tools/kf_update_microbench.py)
      if ((kind == MOT_KF_XYAH || kind == MOT_KF_XYWH) && !lane_per_track) {  // the 8-state filters: one lane per covariance row
        if (ntasks <= 0 || max_n <= 0) return hipSuccess;
        static const int max_gx = std::getenv("MOT_KF_UPDATE_GX") ? std::atoi(std::getenv("MOT_KF_UPDATE_GX")) : kUpdGroups;  // (measurement aid)
        const int want = (max_n + kUpdTracks - 1) / kUpdTracks;
        dim3 grid((want < max_gx) ? want : max_gx, ntasks), block(kUpdTracks * 8);
        if (kind == MOT_KF_XYAH) hipLaunchKernelGGL((kf_update8_kernel<MOT_KF_XYAH, kUpdTracks>), grid, block, 0, st, tasks);
        else hipLaunchKernelGGL((kf_update8_kernel<MOT_KF_XYWH, kUpdTracks>), grid, block, 0, st, tasks);
        return hipGetLastError();
      }
      return launch_kf<OP_UPDATE>(kind, tasks, ntasks, max_n, st);
    case OP_BOXES: return launch_kf<OP_BOXES>(kind, tasks, ntasks, max_n, st);
    case OP_WARP: return (kind == MOT_KF_XYAH) ? hipErrorInvalidValue : launch_kf<OP_WARP>(kind, tasks, ntasks, max_n, st);
    case OP_PREDICT_WARP: return (kind == MOT_KF_XYAH) ? hipErrorInvalidValue : launch_kf<OP_PREDICT_WARP>(kind, tasks, ntasks, max_n, st);
    case OP_PREDICT_BOXES: return launch_kf<OP_PREDICT_BOXES>(kind, tasks, ntasks, max_n, st);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_gate(int kind, const mot_gate_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  if (max_n > 65535) return hipErrorInvalidValue;
  dim3 grid((max_m + kGateThreads - 1) / kGateThreads, max_n, ntasks), block(kGateThreads);
  switch (kind) {
    case MOT_KF_XYAH: hipLaunchKernelGGL((gate_kernel<MOT_KF_XYAH>), grid, block, 0, st, tasks); break;
    case MOT_KF_XYWH: hipLaunchKernelGGL((gate_kernel<MOT_KF_XYWH>), grid, block, 0, st, tasks); break;
    default: return hipErrorInvalidValue;  // the XYSR filter has no gating distance in the reference
  }
  return hipGetLastError();
}
hipError_t launch_det(int kind, const mot_det_task* tasks, int ntasks, int max_n, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  dim3 grid((max_n + kThreads - 1) / kThreads, ntasks), block(kThreads);
  switch (kind) {
    case MOT_DET_XYSR: hipLaunchKernelGGL((det_kernel<MOT_DET_XYSR>), grid, block, 0, st, tasks); break;
    case MOT_DET_XYAH: hipLaunchKernelGGL((det_kernel<MOT_DET_XYAH>), grid, block, 0, st, tasks); break;
    case MOT_DET_XYWH: hipLaunchKernelGGL((det_kernel<MOT_DET_XYWH>), grid, block, 0, st, tasks); break;
    case MOT_DET_TLWH: hipLaunchKernelGGL((det_kernel<MOT_DET_TLWH>), grid, block, 0, st, tasks); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace mot
