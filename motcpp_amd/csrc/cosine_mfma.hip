// Appearance-distance matrices (utils::embedding_distance, src/utils/matching.cpp:67-101).
//
// cosine: out[i][j] = max(0, 1 - a_i.b_j / (|a_i||b_j| + 1e-10)) — the contraction runs on v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, which on CDNA4 is bit-for-bit a k-ordered fmaf chain (one rounding per product, no wider accumulator).
// That is the property that keeps this kernel inside the 1e-4 budget without any fp16/bf16 rounding of the embeddings (and
// bit-identical to the CPU restatement, whose dot products are that same chain).
// Workgroup = 4 wavefronts = 128 x 128 output tile (each wavefront 2 x 2 MFMA tiles of 32 x 32; 64 x 64 for problems with fewer than
// 128 rows or columns); K is walked in slabs of 32 through TWO LDS buffers (row stride 36 floats, the slab's even ks first, then the
// odd ks: the four operands a lane needs for four consecutive MFMA steps are one ds_read_b128, conflict-free), one barrier per slab;
// the next slab's global loads are issued a slab ahead and stay in flight across the 64 chained MFMAs of the current one; the row
// norms are accumulated from the same LDS slabs (k-ordered fmaf chains, one row per lane) — so no separate pass reads the
// features again. dot: the raw inner product (DeepOC-SORT's embedding similarity, deepocsort.cpp:404) — same kernel, no
// norms. euclidean (matching.cpp:93-101): |a_i - b_j|, a k-ordered chain of squared differences on the vector ALUs (it is
// not a contraction) over the same slabs.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "../../include/motcpp_amd.h"

namespace {

constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { kCosine = 0, kDot = 1, kEuclid = 2 };

// TILE = 64: a wavefront owns one 32 x 32 MFMA tile. TILE = 128 (problems with at least 128 rows and columns): a wavefront owns
// 64 x 64 = 2 x 2 MFMA tiles — four independent accumulator chains per wavefront, one LDS operand read per MFMA instead of two,
// and half the feature bytes per flop out of L2 (each row tile is re-read once per column tile of the problem: at 64 x 64 the
// fp32 MFMA rate would need ~10 TB/s of L2 reads). Every output element is still the k-ordered chain of its own products.
template <int METRIC, int TILE, int kSlab>
__global__ void __launch_bounds__(kThreads) embed_kernel(const mot_cos_task* __restrict__ tasks, int tiles_x, int tiles_y) {
  constexpr int TPR = kThreads / TILE;  // threads staging one row of a slab
  constexpr int QPT = (kSlab / 4) / TPR;  // 16-byte pieces of a slab row per thread
  constexpr int kHalf = kSlab / 2;      // the slab's even ks, then its odd ks
  static_assert(QPT >= 1 && (kSlab == 16 || kSlab == 32), "slab width");
  constexpr int NA = TILE / 64;         // MFMA tiles per wavefront and dimension
  static_assert(METRIC != kEuclid || TILE == 64, "the euclidean variant keeps the 64 x 64 tile");
  // XCD-aware order of the 1-D grid: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own L2.
  // The tiles of one task share its feature rows (a row tile is read by every column tile, and the other way round), so the ids an
  // XCD receives are mapped to CONSECUTIVE tiles: a task's tiles run on one XCD at about the same time and meet in its L2 instead of
  // each fetching the features from HBM.
  int v;
  {
    const int N = static_cast<int>(gridDim.x), L = static_cast<int>(blockIdx.x), x = L & 7;
    const int base = N >> 3, rem = N & 7;                  // XCD y owns base + (y < rem) ids
    v = x * base + ((x < rem) ? x : rem) + (L >> 3);
  }
  const int per_task = tiles_x * tiles_y;
  const int task = v / per_task, tv = v - task * per_task;
  const int ty = tv / tiles_x, tx = tv - ty * tiles_x;
  const mot_cos_task T = tasks[task];
  const int row0 = ty * TILE, col0 = tx * TILE;
  if (row0 >= T.n || col0 >= T.m) return;
  // LDS slab of a matrix tile: row stride kLd floats; inside a row the slab's 32 k-values are stored EVEN ks first, then ODD ks
  // (position of k = 16 (k & 1) + (k >> 1)). An MFMA step t takes k = 2 t + h from lane half h = lane >> 5, so the four operands a
  // lane needs for four consecutive steps are one aligned 16-byte read (ds_read_b128), and eight lanes' reads hit all 32 banks once
  // (stride 36 floats = 4 banks per row). Two slabs in LDS: the next one is written while the current one feeds the MFMAs — one barrier
  // per slab — and its global loads were issued a slab earlier.
  constexpr int kLd = kSlab + 4;
  __shared__ __attribute__((aligned(16))) float As[2][TILE][kLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][TILE][kLd];
  __shared__ float nrm[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sr = tid / TPR, sq = tid % TPR;
  const float* pa = nullptr;
  const float* pb = nullptr;
  {
    const int r = row0 + sr, c = col0 + sr;
    if (r < T.n) pa = T.a + static_cast<size_t>(T.aidx ? T.aidx[r] : r) * T.lda;
    if (c < T.m) pb = T.b + static_cast<size_t>(T.bidx ? T.bidx[c] : c) * T.ldb;
  }
  const bool vec = ((T.lda | T.ldb | T.d) & 3) == 0 && ((reinterpret_cast<size_t>(T.a) | reinterpret_cast<size_t>(T.b)) & 15) == 0;
  float4 ra[QPT], rb[QPT];
  const float* pa_ld = pa ? pa : T.a;  // (rows outside the problem load a valid address and are zeroed by a select: no branch, so
  const float* pb_ld = pb ? pb : T.b;  //  that the loads stay in flight until the values are staged — a branch would wait for them)
  auto fetch = [&](auto fast, int k0) {  // (entries beyond d are zeros: they extend every chain by fma(0, 0, s) = s)
    if constexpr (decltype(fast)::value) {  // whole slabs of aligned rows: unconditional 16-byte loads
#pragma unroll
      for (int h = 0; h < QPT; ++h) {
        const int k = k0 + 4 * (h * TPR + sq);
        const float4 va = *reinterpret_cast<const float4*>(pa_ld + k), vb = *reinterpret_cast<const float4*>(pb_ld + k);
        ra[h] = pa ? va : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[h] = pb ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
#pragma unroll
    for (int h = 0; h < QPT; ++h) {
      const int k = k0 + 4 * (h * TPR + sq);
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
  auto stage = [&](int buf) {  // registers -> LDS slab `buf`: k, k+2 to the even half, k+1, k+3 to the odd half (two 8-byte stores each)
#pragma unroll
    for (int h = 0; h < QPT; ++h) {
      const int k = 4 * (h * TPR + sq);
      *reinterpret_cast<float2*>(&As[buf][sr][k >> 1]) = make_float2(ra[h].x, ra[h].z);
      *reinterpret_cast<float2*>(&As[buf][sr][kHalf + (k >> 1)]) = make_float2(ra[h].y, ra[h].w);
      *reinterpret_cast<float2*>(&Bs[buf][sr][k >> 1]) = make_float2(rb[h].x, rb[h].z);
      *reinterpret_cast<float2*>(&Bs[buf][sr][kHalf + (k >> 1)]) = make_float2(rb[h].y, rb[h].w);
    }
  };
  const int wr = wave >> 1, wc = wave & 1;  // wavefront -> (TILE/2) x (TILE/2) sub-tile
  // (round 5) 32 x 32 MFMA tiles that lie wholly outside the problem are skipped (wavefront-uniform): a pool of 900 tracks x 450 detections
  // pads to 1024 x 512 in 128 x 128 workgroup tiles — 23 % of the matrix-core time went into rows and columns nobody reads; at the 32-wide
  // granularity it is 928 x 480. The workgroup still meets at its barriers, but the matrix pipe is free for the other workgroups of the CU.
  bool row_ok[NA], col_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    row_ok[i] = row0 + wr * (TILE / 2) + 32 * i < T.n;
    col_ok[i] = col0 + wc * (TILE / 2) + 32 * i < T.m;
  }
  const bool all_ok = row_ok[NA - 1] && col_ok[NA - 1];
  f32x16 acc[NA][NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // norm role (cosine): the 2 * TILE rows [A rows | B rows] are dealt to the threads, TILE / 128 ... one row per (wave, lane slot)
  constexpr int NQ = (2 * TILE) / 128;            // 1 (lanes < 32 of each wavefront) or 2 (every lane)
  const bool norm_lane = (NQ == 2) || lane < 32;
  const int q = (NQ == 2) ? wave * 64 + lane : wave * 32 + (lane & 31);
  float nsum = 0.0f;
  auto contract = [&](auto fast) {
    fetch(fast, 0);
    stage(0);
    if (kSlab < T.d) fetch(fast, kSlab);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < T.d; k0 += kSlab) {
      if constexpr (METRIC == kEuclid) {
        // 64 x 64 distances on the vector ALUs: thread -> row tid >> 2, columns (tid & 3) + 4 j; chains in k order
  #pragma unroll
        for (int j = 0; j < 16; ++j) {
          float sacc = acc[0][0][j];
          const int cc = sq + 4 * j;
  #pragma unroll
          for (int g = 0; g < kSlab / 8; ++g) {  // even and odd halves by 16-byte reads (the same access type as the stores: float vectors), k order
            const float4 ae = *reinterpret_cast<const float4*>(&As[buf][sr][4 * g]), ao = *reinterpret_cast<const float4*>(&As[buf][sr][kHalf + 4 * g]);
            const float4 be = *reinterpret_cast<const float4*>(&Bs[buf][cc][4 * g]), bo = *reinterpret_cast<const float4*>(&Bs[buf][cc][kHalf + 4 * g]);
            float df;
            df = ae.x - be.x; sacc = __builtin_fmaf(df, df, sacc); df = ao.x - bo.x; sacc = __builtin_fmaf(df, df, sacc);
            df = ae.y - be.y; sacc = __builtin_fmaf(df, df, sacc); df = ao.y - bo.y; sacc = __builtin_fmaf(df, df, sacc);
            df = ae.z - be.z; sacc = __builtin_fmaf(df, df, sacc); df = ao.z - bo.z; sacc = __builtin_fmaf(df, df, sacc);
            df = ae.w - be.w; sacc = __builtin_fmaf(df, df, sacc); df = ao.w - bo.w; sacc = __builtin_fmaf(df, df, sacc);
          }
          acc[0][0][j] = sacc;
        }
      } else {
        if (METRIC == kCosine && norm_lane) {
          const float* rowp = (q < TILE) ? As[buf][q] : Bs[buf][q - TILE];
  #pragma unroll
          for (int g = 0; g < kSlab / 8; ++g) {  // even and odd halves by 16-byte reads, consumed in k order
            const float4 ev = *reinterpret_cast<const float4*>(rowp + 4 * g), od = *reinterpret_cast<const float4*>(rowp + kHalf + 4 * g);
            nsum = __builtin_fmaf(ev.x, ev.x, nsum); nsum = __builtin_fmaf(od.x, od.x, nsum);
            nsum = __builtin_fmaf(ev.y, ev.y, nsum); nsum = __builtin_fmaf(od.y, od.y, nsum);
            nsum = __builtin_fmaf(ev.z, ev.z, nsum); nsum = __builtin_fmaf(od.z, od.z, nsum);
            nsum = __builtin_fmaf(ev.w, ev.w, nsum); nsum = __builtin_fmaf(od.w, od.w, nsum);
          }
        }
        const int hoff = kHalf * (lane >> 5);
  #pragma unroll
        for (int g = 0; g < kSlab / 8; ++g) {  // four MFMA steps per 16-byte operand read; k-ascending chain: D = fma(a_k1,b_k1, fma(a_k0,b_k0, C))
          float4 a4[NA], b4[NA];
  #pragma unroll
          for (int i = 0; i < NA; ++i) {
            a4[i] = *reinterpret_cast<const float4*>(&As[buf][wr * (TILE / 2) + 32 * i + (lane & 31)][hoff + 4 * g]);
            b4[i] = *reinterpret_cast<const float4*>(&Bs[buf][wc * (TILE / 2) + 32 * i + (lane & 31)][hoff + 4 * g]);
          }
          auto steps = [&](auto guarded) {
  #pragma unroll
            for (int e = 0; e < 4; ++e)
  #pragma unroll
              for (int i = 0; i < NA; ++i)
  #pragma unroll
                for (int j = 0; j < NA; ++j) {
                  const float av = (e == 0) ? a4[i].x : (e == 1) ? a4[i].y : (e == 2) ? a4[i].z : a4[i].w;
                  const float bv = (e == 0) ? b4[j].x : (e == 1) ? b4[j].y : (e == 2) ? b4[j].z : b4[j].w;
                  if (!decltype(guarded)::value || (row_ok[i] && col_ok[j])) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
          };
          if (all_ok) steps(std::false_type()); else steps(std::true_type());  // (interior wavefronts keep their unbroken chain of MFMAs)
        }
      }
      if (k0 + kSlab < T.d) {
        stage(buf ^ 1);                                   // the slab fetched while the previous one was consumed
        if (k0 + 2 * kSlab < T.d) fetch(fast, k0 + 2 * kSlab);  // and the one after it is requested now
      }
      __syncthreads();
      buf ^= 1;
    }
  };
  // two instances of the loop: the common case (aligned rows, d a multiple of the slab) has no branch around its loads, so they stay in
  // flight across the MFMAs of the current slab; everything else takes the guarded loads
  if (vec && (T.d % kSlab) == 0) contract(std::true_type()); else contract(std::false_type());
  if constexpr (METRIC == kEuclid) {
    const int r = row0 + sr;
    if (r < T.n) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = col0 + sq + 4 * j;
        if (c < T.m) T.out[static_cast<size_t>(r) * T.ldo + c] = sqrtf(acc[0][0][j]);
      }
    }
    return;
  }
  if (METRIC == kCosine) {
    if (norm_lane) nrm[q] = sqrtf(nsum);
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int cl = wc * (TILE / 2) + 32 * j + (lane & 31);
    const int c = col0 + cl;
    if (c >= T.m) continue;
    const float nb = (METRIC == kCosine) ? nrm[TILE + cl] : 0.0f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rl = wr * (TILE / 2) + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int r = row0 + rl;
        if (r < T.n) {
          if (METRIC == kDot) { T.out[static_cast<size_t>(r) * T.ldo + c] = acc[i][j][reg]; continue; }
          const float sim = acc[i][j][reg] / (nrm[rl] * nb + 1e-10f);
          const float v = 1.0f - sim;
          T.out[static_cast<size_t>(r) * T.ldo + c] = (0.0f < v) ? v : 0.0f;  // std::max(0.0f, v)
        }
      }
    }
  }
}

}  // namespace

namespace mot {
// metric: 0 cosine distance, 1 raw dot product, 2 euclidean distance
hipError_t launch_embed(int metric, const mot_cos_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  static const bool force64 = std::getenv("MOT_EMBED_TILE64") != nullptr;  // measurement aid (A/B of the two tilings)
  const bool big = max_n >= 128 && max_m >= 128 && metric != kEuclid && !force64;
  const int tile = big ? 128 : 64;
  const int tx = (max_m + tile - 1) / tile, ty = (max_n + tile - 1) / tile;
  const long long nblk = static_cast<long long>(tx) * ty * ntasks;
  if (nblk > 0x7fffffffll) return hipErrorInvalidValue;
  const dim3 g1(static_cast<unsigned>(nblk));
  // 128 x 128 tiles walk K in slabs of 16 (41 KB of LDS: three workgroups per CU, so that one's epilogue and prologue hide behind the
  // others' MFMAs; slabs of 32 take 74 KB: two per CU. MOT_EMBED_SLAB=32 selects them, for A/B measurements)
  static const int slab_big = (std::getenv("MOT_EMBED_SLAB") && std::atoi(std::getenv("MOT_EMBED_SLAB")) == 32) ? 32 : 16;
  if (metric == kCosine && big && slab_big == 16) hipLaunchKernelGGL((embed_kernel<kCosine, 128, 16>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else if (metric == kCosine && big) hipLaunchKernelGGL((embed_kernel<kCosine, 128, 32>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else if (metric == kCosine) hipLaunchKernelGGL((embed_kernel<kCosine, 64, 32>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else if (metric == kDot && big && slab_big == 16) hipLaunchKernelGGL((embed_kernel<kDot, 128, 16>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else if (metric == kDot && big) hipLaunchKernelGGL((embed_kernel<kDot, 128, 32>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else if (metric == kDot) hipLaunchKernelGGL((embed_kernel<kDot, 64, 32>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else if (metric == kEuclid) hipLaunchKernelGGL((embed_kernel<kEuclid, 64, 32>), g1, dim3(kThreads), 0, st, tasks, tx, ty);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_cosine(const mot_cos_task* tasks, int ntasks, int max_n, int max_m, hipStream_t st) {
  return launch_embed(kCosine, tasks, ntasks, max_n, max_m, st);
}
}  // namespace mot
