// Small dense helpers shared by the Kalman kernels (kf_kernels.hip, boost_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mot {
namespace kfs {
// Partial-pivot LU inverse of a 4x4 (XYWH's S.inverse(), xywh_kf.hpp:124).
__device__ __forceinline__ void inv_lu4(const float S[4][4], float inv[4][4]) {
  float lu[4][4];
  int perm[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) lu[i][j] = S[i][j];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int p = k;
    float best = fabsf(lu[k][k]);
#pragma unroll
    for (int i = k + 1; i < 4; ++i) {
      float v = fabsf(lu[i][k]);
      if (v > best) { best = v; p = i; }
    }
    // row swap with a data-dependent index done as selects to keep everything in registers
#pragma unroll
    for (int i = k + 1; i < 4; ++i) {
      if (p == i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { float t = lu[k][j]; lu[k][j] = lu[i][j]; lu[i][j] = t; }
        int tp = perm[k]; perm[k] = perm[i]; perm[i] = tp;
      }
    }
#pragma unroll
    for (int i = k + 1; i < 4; ++i) lu[i][k] /= lu[k][k];
#pragma unroll
    for (int i = k + 1; i < 4; ++i)
#pragma unroll
      for (int j = k + 1; j < 4; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (perm[i] == c) ? 1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = i + 1; r < 4; ++r) b[r] -= b[i] * lu[r][i];
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      b[i] /= lu[i][i];
#pragma unroll
      for (int r = 0; r < i; ++r) b[r] -= b[i] * lu[r][i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) inv[i][c] = b[i];
  }
}
}  // namespace kfs
}  // namespace mot
