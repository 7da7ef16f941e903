// Appearance-distance matrices (utils::embedding_distance, src/utils/matching.cpp:67-101).
//
// cosine: out[i][j] = max(0, 1 - a_i.b_j / (|a_i||b_j| + 1e-10)) — the contraction runs on v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, which on CDNA4 is bit-for-bit a k-ordered fmaf chain (one rounding per product, no wider accumulator).
// That is the property that keeps this kernel inside the 1e-4 budget without any fp16/bf16 rounding of the embeddings (and
// bit-identical to the CPU restatement, whose dot products are that same chain).
// Workgroup = 4 wavefronts = 64 x 64 output tile (2 x 2 MFMA tiles of 32 x 32); K is walked in slabs of 32 through LDS
// (row stride 33 floats: the MFMA operand read, lane -> (row = lane & 31, k = lane >> 5), is conflict-free). The next slab
// is already in flight (two 16-byte loads per matrix per thread, held in registers) while the 16 chained MFMAs of the current
// one run; the row norms are accumulated from the same LDS slabs — lane l < 32 of wavefront w owns row 32 w + l of the
// 128 rows of the tile pair and extends its k-ordered chain by the slab's 32 entries — so no separate pass reads the
// features again. dot: the raw inner product (DeepOC-SORT's embedding similarity, deepocsort.cpp:404) — same kernel, no
// norms. euclidean (matching.cpp:93-101): |a_i - b_j|, a k-ordered chain of squared differences on the vector ALUs (it is
// not a contraction) over the same slabs.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 64;
constexpr int kSlab = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { kCosine = 0, kDot = 1, kEuclid = 2 };

template <int METRIC>
__global__ void __launch_bounds__(kThreads) embed_kernel(const mot_cos_task* __restrict__ tasks) {
  const mot_cos_task T = tasks[blockIdx.z];
  const int row0 = blockIdx.y * kTile, col0 = blockIdx.x * kTile;
  if (row0 >= T.n || col0 >= T.m) return;
  __shared__ float As[kTile][kSlab + 1];
  __shared__ float Bs[kTile][kSlab + 1];
  __shared__ float nrm[2 * kTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // staging role: row tid >> 2 of each tile, k quads (tid & 3) and 4 + (tid & 3) of a slab
  const int sr = tid >> 2, sq = tid & 3;
  const float* pa = nullptr;
  const float* pb = nullptr;
  {
    const int r = row0 + sr, c = col0 + sr;
    if (r < T.n) pa = T.a + static_cast<size_t>(T.aidx ? T.aidx[r] : r) * T.lda;
    if (c < T.m) pb = T.b + static_cast<size_t>(T.bidx ? T.bidx[c] : c) * T.ldb;
  }
  const bool vec = ((T.lda | T.ldb | T.d) & 3) == 0 && ((reinterpret_cast<size_t>(T.a) | reinterpret_cast<size_t>(T.b)) & 15) == 0;
  float4 ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + 16 * h + 4 * sq;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (vec && k + 3 < T.d) {
        if (pa) va = *reinterpret_cast<const float4*>(pa + k);
        if (pb) vb = *reinterpret_cast<const float4*>(pb + k);
      } else {
        float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (pa && k + e < T.d) ta[e] = pa[k + e];
          if (pb && k + e < T.d) tb[e] = pb[k + e];
        }
        va = make_float4(ta[0], ta[1], ta[2], ta[3]); vb = make_float4(tb[0], tb[1], tb[2], tb[3]);
      }
      ra[h] = va; rb[h] = vb;
    }
  };
  const int wr = wave >> 1, wc = wave & 1;  // wavefront -> 32x32 sub-tile
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // norm role (cosine): lane < 32 of wavefront w owns row q = 32 w + lane of [A rows 0..63 | B rows 0..63]
  const int q = wave * 32 + (lane & 31);
  float nsum = 0.0f;
  fetch(0);
  for (int k0 = 0; k0 < T.d; k0 += kSlab) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 16 * h + 4 * sq;
      As[sr][k] = ra[h].x; As[sr][k + 1] = ra[h].y; As[sr][k + 2] = ra[h].z; As[sr][k + 3] = ra[h].w;
      Bs[sr][k] = rb[h].x; Bs[sr][k + 1] = rb[h].y; Bs[sr][k + 2] = rb[h].z; Bs[sr][k + 3] = rb[h].w;
    }
    __syncthreads();
    if (k0 + kSlab < T.d) fetch(k0 + kSlab);  // in flight while this slab is consumed
    const int kk_end = min(kSlab, T.d - k0);
    if constexpr (METRIC == kEuclid) {
      // 64 x 64 distances on the vector ALUs: thread -> row tid >> 2, columns (tid & 3) + 4 j; chains in k order
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float sacc = acc[j];
        const int cc = sq + 4 * j;
        for (int kk = 0; kk < kk_end; ++kk) { const float df = As[sr][kk] - Bs[cc][kk]; sacc = __builtin_fmaf(df, df, sacc); }
        acc[j] = sacc;
      }
    } else {
      if (METRIC == kCosine && lane < 32) {
        const float* rowp = (q < kTile) ? As[q] : Bs[q - kTile];
        for (int kk = 0; kk < kk_end; ++kk) nsum = __builtin_fmaf(rowp[kk], rowp[kk], nsum);
      }
      for (int kk = 0; kk < kk_end; kk += 2) {  // k-ascending chain: D = fma(a_k1,b_k1, fma(a_k0,b_k0, C))
        const float a = As[wr * 32 + (lane & 31)][kk + (lane >> 5)];
        const float b = Bs[wc * 32 + (lane & 31)][kk + (lane >> 5)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if constexpr (METRIC == kEuclid) {
    const int r = row0 + sr;
    if (r < T.n) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = col0 + sq + 4 * j;
        if (c < T.m) T.out[static_cast<size_t>(r) * T.ldo + c] = sqrtf(acc[j]);
      }
    }
    return;
  }
  if (METRIC == kCosine) {
    if (lane < 32) nrm[q] = sqrtf(nsum);
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int cl = wc * 32 + (lane & 31);
  const int c = col0 + cl;
  if (c < T.m) {
    const float nb = (METRIC == kCosine) ? nrm[kTile + cl] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rl = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int r = row0 + rl;
      if (r < T.n) {
        if (METRIC == kDot) { T.out[static_cast<size_t>(r) * T.ldo + c] = acc[reg]; continue; }
        const float sim = acc[reg] / (nrm[rl] * nb + 1e-10f);
        const float v = 1.0f - sim;
        T.out[static_cast<size_t>(r) * T.ldo + c] = (0.0f < v) ? v : 0.0f;  // std::max(0.0f, v)
      }
    }
  }
}

}  // namespace

namespace mot {
// metric: 0 cosine distance, 1 raw dot product, 2 euclidean distance
hipError_t launch_embed(int metric, const mot_cos_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  dim3 g2((max_m + kTile - 1) / kTile, (max_n + kTile - 1) / kTile, ntasks);
  if (metric == kCosine) hipLaunchKernelGGL(embed_kernel<kCosine>, g2, dim3(kThreads), 0, st, tasks);
  else if (metric == kDot) hipLaunchKernelGGL(embed_kernel<kDot>, g2, dim3(kThreads), 0, st, tasks);
  else if (metric == kEuclid) hipLaunchKernelGGL(embed_kernel<kEuclid>, g2, dim3(kThreads), 0, st, tasks);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_cosine(const mot_cos_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  return launch_embed(kCosine, tasks, ntasks, max_n, max_m, st);
}
}  // namespace mot
