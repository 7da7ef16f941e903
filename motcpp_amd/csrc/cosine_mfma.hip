// Cosine-distance matrix on the gfx950 matrix cores (utils::embedding_distance,
// src/utils/matching.cpp:67-92): out[i][j] = max(0, 1 - a_i.b_j / (|a_i||b_j| + 1e-10)).
//
// The contraction runs on v_mfma_f32_32x32x2_f32 — f32 in, f32 accumulate, which on CDNA4 is
// bit-for-bit a k-ordered fmaf chain (one rounding per product, no wider accumulator). That is
// the property that lets this kernel be bit-identical to the CPU restatement's dot products and
// stay inside the 1e-4 budget without any fp16/bf16 rounding of the embeddings.
// Workgroup = 4 wavefronts = 64 x 64 output tile (2 x 2 MFMA tiles of 32 x 32); K is walked in
// slabs of 32 staged through LDS with coalesced row loads (row stride padded by one float: the
// MFMA operand read, lane -> (row = lane&31, k = lane>>5), is then conflict-free); each slab
// feeds 16 chained MFMAs per wavefront. Row norms come from a pre-pass (one lane per row, the
// same fmaf chain) and are applied in the epilogue together with the clamp.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 64;
constexpr int kSlab = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(kThreads) norm_kernel(const mot_cos_task* __restrict__ tasks) {
  const mot_cos_task T = tasks[blockIdx.y];
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= T.n + T.m) return;
  const bool isa = i < T.n;
  const int r = isa ? i : i - T.n;
  const float* p = isa ? T.a + static_cast<size_t>(T.aidx ? T.aidx[r] : r) * T.lda
                       : T.b + static_cast<size_t>(T.bidx ? T.bidx[r] : r) * T.ldb;
  float s = 0.0f;
  for (int k = 0; k < T.d; ++k) s = __builtin_fmaf(p[k], p[k], s);
  (isa ? T.norm_a : T.norm_b)[r] = sqrtf(s);
}

__global__ void __launch_bounds__(kThreads) cosine_kernel(const mot_cos_task* __restrict__ tasks) {
  const mot_cos_task T = tasks[blockIdx.z];
  const int row0 = blockIdx.y * kTile, col0 = blockIdx.x * kTile;
  if (row0 >= T.n || col0 >= T.m) return;
  __shared__ float As[kTile][kSlab + 1];
  __shared__ float Bs[kTile][kSlab + 1];
  __shared__ const float* arow[kTile];
  __shared__ const float* brow[kTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < kTile) {
    const int r = row0 + tid;
    arow[tid] = (r < T.n) ? T.a + static_cast<size_t>(T.aidx ? T.aidx[r] : r) * T.lda : nullptr;
  } else if (tid < 2 * kTile) {
    const int c = col0 + tid - kTile;
    brow[tid - kTile] = (c < T.m) ? T.b + static_cast<size_t>(T.bidx ? T.bidx[c] : c) * T.ldb : nullptr;
  }
  __syncthreads();
  const int wr = wave >> 1, wc = wave & 1;  // wavefront -> 32x32 sub-tile
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < T.d; k0 += kSlab) {
    // stage 64 rows x 32 k of A and B: 8 consecutive lanes read 32 consecutive floats of one row
    for (int e = tid; e < kTile * kSlab; e += kThreads) {
      const int r = e >> 5, k = e & 31;
      const float* pa = arow[r];
      const float* pb = brow[r];
      As[r][k] = (pa && k0 + k < T.d) ? pa[k0 + k] : 0.0f;
      Bs[r][k] = (pb && k0 + k < T.d) ? pb[k0 + k] : 0.0f;
    }
    __syncthreads();
    const int kk_end = min(kSlab, T.d - k0);
    for (int kk = 0; kk < kk_end; kk += 2) {  // k-ascending chain: D = fma(a_k1,b_k1, fma(a_k0,b_k0, C))
      const float a = As[wr * 32 + (lane & 31)][kk + (lane >> 5)];
      const float b = Bs[wc * 32 + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int c = col0 + wc * 32 + (lane & 31);
  if (c < T.m) {
    const float nb = T.norm_b[c];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (r < T.n) {
        const float sim = acc[reg] / (T.norm_a[r] * nb + 1e-10f);
        const float v = 1.0f - sim;
        T.out[static_cast<size_t>(r) * T.ldo + c] = (0.0f < v) ? v : 0.0f;  // std::max(0.0f, v)
      }
    }
  }
}

}  // namespace

namespace mot {
hipError_t launch_cosine(const mot_cos_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  dim3 g1((max_n + max_m + kThreads - 1) / kThreads, ntasks);
  hipLaunchKernelGGL(norm_kernel, g1, dim3(kThreads), 0, st, tasks);
  dim3 g2((max_m + kTile - 1) / kTile, (max_n + kTile - 1) / kTile, ntasks);
  hipLaunchKernelGGL(cosine_kernel, g2, dim3(kThreads), 0, st, tasks);
  return hipGetLastError();
}
}  // namespace mot
