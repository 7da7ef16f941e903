// N x M box-cost matrices for gfx950: IoU family (iou.hpp:63-100, matching.cpp:62-65,130-143,
// botsort.cpp:433-466), OC-SORT's IoU + velocity-direction cost (ocsort.cpp:624-679,699) and the
// BoT-SORT feature maintenance (botsort.cpp:38-46,158-169).
//
// Tiling: a 256-thread workgroup (4 wavefronts) owns a 64 x 64 output tile. The 64 row boxes and
// 64 column boxes (+ area, + per-column confidence) are staged once in LDS (2.8 KB) and every
// thread produces a 4(rows, stride 16) x 4(consecutive columns) micro-tile, so each output row
// segment is written as one float4 per lane = 256 contiguous bytes per 16 lanes. The matrix is
// written exactly once and never re-read here: algorithmic traffic = 4*n*m + 16*(n+m) bytes.
// Grid: x = column tiles, y = row tiles, z = task (one association problem per stream/stage).
//
// Arithmetic is the reference's order verbatim (+,-,*,/,min,max with std::min/std::max NaN
// behaviour), -ffp-contract=off, so costs are bit-identical to the CPU restatement.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"
#include "cost_math.hpp"

namespace {

constexpr int kTile = 64;
constexpr int kThreads = 256;

using mot::iou_pair;
using mot::smax;
using mot::smin;

struct BoxTile {
  float c[4][kTile];
  float area[kTile];
};

// loads up to 64 boxes (optionally gathered) of a SoA [4][ld] array into LDS; lanes 0..63 of the block do it
__device__ __forceinline__ void stage_boxes(BoxTile& t, const float* base, int ld, const int* idx, int first, int count,
                                            int lane) {
  if (lane >= 0 && lane < kTile) {
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < count) {
      const int g = idx ? idx[first + lane] : first + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) b[k] = base[static_cast<size_t>(k) * ld + g];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) t.c[k][lane] = b[k];
    t.area[lane] = (b[2] - b[0]) * (b[3] - b[1]);
  }
}

// GENERAL: any mot_assoc measure; the default instance evaluates plain IoU only (half the registers, twice as fast)
template <bool GENERAL>
__global__ void __launch_bounds__(kThreads) iou_kernel(const mot_iou_task* __restrict__ tasks) {
  const mot_iou_task T = tasks[blockIdx.z];
  const int row0 = blockIdx.y * kTile, col0 = blockIdx.x * kTile;
  if (row0 >= T.n || col0 >= T.m) return;
  __shared__ BoxTile A, B;
  __shared__ float conf[kTile];
  const int tid = threadIdx.x;
  const int nrow = min(kTile, T.n - row0), ncol = min(kTile, T.m - col0);
  stage_boxes(A, T.a, T.lda, T.aidx, row0, nrow, tid);
  stage_boxes(B, T.b, T.ldb, T.bidx, col0, ncol, tid - 64);
  if (tid >= 128 && tid < 128 + kTile) {
    const int l = tid - 128;
    float c = 0.f;
    if (T.bconf && l < ncol) c = T.bconf[T.bidx ? T.bidx[col0 + l] : col0 + l];
    conf[l] = c;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  const mot::CostParams cp{T.mode, T.prox_thresh, T.app_thresh, T.fuse, T.emb != nullptr, T.emb == nullptr && T.lde < 0, T.assoc, T.frame_diag};
  float bb[4][4], barea[4], bc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = tx * 4 + q;
#pragma unroll
    for (int k = 0; k < 4; ++k) bb[q][k] = B.c[k][c];
    barea[q] = B.area[c];
    bc[q] = conf[c];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = ty + 16 * p;
    if (r >= nrow) continue;
    const float aa[4] = {A.c[0][r], A.c[1][r], A.c[2][r], A.c[3][r]};
    const float aarea = A.area[r];
    float out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float iou;
      if constexpr (GENERAL) iou = mot::assoc_pair(cp.assoc, cp.frame_diag, aa, aarea, bb[q], barea[q]);
      else iou = iou_pair(aa, aarea, bb[q], barea[q]);
      const int cq = tx * 4 + q;
      const float v = mot::cost_from_iou(cp, iou, bc[q], [&]() {
        return (cq < ncol) ? T.emb[static_cast<size_t>(row0 + r) * T.lde + col0 + cq] : 0.f;
      });
      out[q] = v;
    }
    const int gr = row0 + r;
    const int gc = col0 + tx * 4;
    if (T.cost) {
      float* dst = T.cost + static_cast<size_t>(gr) * T.ldc + gc;
      if (tx * 4 + 3 < ncol && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
        *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (tx * 4 + q < ncol) dst[q] = out[q];
      }
    }
    if (T.pairs) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tx * 4 + q < ncol && out[q] < T.pair_thresh) {
          const int k = atomicAdd(T.npairs, 1);
          if (k < T.pairs_cap) { T.pairs[2 * k] = gr; T.pairs[2 * k + 1] = gc + q; }
        }
    }
    if (T.dup_a) {  // idempotent byte stores: no counter, no overflow
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tx * 4 + q < ncol && out[q] < T.pair_thresh) {
          if (T.age_a[gr] > T.age_b[gc + q]) T.dup_b[gc + q] = 1;
          else T.dup_a[gr] = 1;
        }
    }
  }
}

// ---- OC-SORT: rows = detections, columns = tracks --------------------------------------------
struct TrkTile {
  float c[4][kTile];
  float area[kTile];
  float vy[kTile], vx[kTile];
  float pcx[kTile], pcy[kTile], valid[kTile];
};

template <bool GENERAL>
__global__ void __launch_bounds__(kThreads) ocsort_kernel(const mot_ocsort_task* __restrict__ tasks) {
  const mot_ocsort_task T = tasks[blockIdx.z];
  const int row0 = blockIdx.y * kTile, col0 = blockIdx.x * kTile;
  if (row0 >= T.nd || col0 >= T.nt) return;
  __shared__ BoxTile D;
  __shared__ float dcx[kTile], dcy[kTile], dscore[kTile];
  __shared__ TrkTile K;
  const int tid = threadIdx.x;
  const int nrow = min(kTile, T.nd - row0), ncol = min(kTile, T.nt - col0);
  stage_boxes(D, T.dbox, T.ldd, T.didx, row0, nrow, tid);
  if (tid < kTile) {
    float s = 0.f;
    if (tid < nrow) s = T.dconf[T.didx ? T.didx[row0 + tid] : row0 + tid];
    dscore[tid] = s;
  }
  if (tid >= 64 && tid < 128) {
    const int l = tid - 64;
    float b[4] = {0.f, 0.f, 0.f, 0.f}, vy = 0.f, vx = 0.f, p[5] = {0.f, 0.f, 0.f, 0.f, -1.f};
    if (l < ncol) {
      const int g = col0 + l;
#pragma unroll
      for (int k = 0; k < 4; ++k) b[k] = T.tbox[static_cast<size_t>(k) * T.ldt + g];
      vy = T.vel[g]; vx = T.vel[static_cast<size_t>(T.ldv) + g];
#pragma unroll
      for (int k = 0; k < 5; ++k) p[k] = T.prev[static_cast<size_t>(k) * T.ldp + g];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) K.c[k][l] = b[k];
    K.area[l] = (b[2] - b[0]) * (b[3] - b[1]);
    K.vy[l] = vy; K.vx[l] = vx;
    K.pcx[l] = (p[0] + p[2]) / 2.0f;  // ocsort.cpp:639-640
    K.pcy[l] = (p[1] + p[3]) / 2.0f;
    K.valid[l] = (p[4] >= 0.0f) ? 1.0f : 0.0f;
  }
  __syncthreads();
  if (tid < kTile) {  // detection centres, ocsort.cpp:637-638
    dcx[tid] = (D.c[0][tid] + D.c[2][tid]) / 2.0f;
    dcy[tid] = (D.c[1][tid] + D.c[3][tid]) / 2.0f;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  const float PI = 3.14159265358979323846f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = ty + 16 * p;
    if (r >= nrow) continue;
    const float da[4] = {D.c[0][r], D.c[1][r], D.c[2][r], D.c[3][r]};
    const float darea = D.area[r];
    float oc[4], oi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tx * 4 + q;
      const float tb[4] = {K.c[0][c], K.c[1][c], K.c[2][c], K.c[3][c]};
      float iou;
      if constexpr (GENERAL) iou = mot::assoc_pair(T.assoc, T.frame_diag, da, darea, tb, K.area[c]);
      else iou = iou_pair(da, darea, tb, K.area[c]);
      const float dx = dcx[r] - K.pcx[c], dy = dcy[r] - K.pcy[c];
      const float norm = sqrtf(dx * dx + dy * dy) + 1e-6f;
      const float Y = dy / norm, X = dx / norm;
      float cs = K.vx[c] * X + K.vy[c] * Y;
      cs = smin(smax(cs, -1.0f), 1.0f);
      const float ac = static_cast<float>(acos(static_cast<double>(cs)));  // canonical fp32 acos
      const float diff = (PI / 2.0f - fabsf(ac)) / PI;
      const float ang = ((K.valid[c] * diff) * T.vdc_weight) * dscore[r];
      oc[q] = -(iou + ang);
      oi[q] = iou;
    }
    const size_t off = static_cast<size_t>(row0 + r) * T.ldc + col0 + tx * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tx * 4 + q < ncol) { T.cost[off + q] = oc[q]; T.iou[off + q] = oi[q]; }
  }
}

// ---- appearance rows: normalise / exponential moving average ------------------------------------------------------------
// mode 0: feat = src / |src| (BotSTrack ctor, botsort.cpp:38-46); 1: feat = alpha*feat + (1-alpha)*src, renormalised
// (update_features, botsort.cpp:158-169); 2: feat = src / |src| if |src| > 1e-6 else src (ReIDBackend::normalize_features,
// reid_backend.cpp:72-88); 3: the EMA with that 1e-6 rule (update_emb, deepocsort.cpp:132-150); 4: StrongSORT's EMA (Track::update,
// strongsort.cpp:165-182: blend with an already normalised src, renormalise when the norm exceeds 1e-10, else the stored row stays);
// 5: src / |src| where |src| > 1e-10 (cosine_distance :317-331); 6: plain copy.
// The squared norm is a k-ordered fmaf chain — that order IS the result, so one lane walks one row for it — but everything
// else is parallel: a 256-thread workgroup owns 32 rows, its four wavefronts stream the rows into an LDS tile with 256-byte
// coalesced runs and blend them on the way (the EMA is elementwise), lanes 0..31 run the 32 chains out of LDS (odd row
// stride: conflict-free), and the wavefronts divide and write back coalesced. A row of up to 256 columns is read once and
// written once; longer rows go through the tile in 256-column chunks (the chain carries over) with one more round trip
// for the division.
constexpr int kFeatRows = 32, kFeatCols = 256;
__global__ void __launch_bounds__(256) feat_kernel(const mot_feat_task* __restrict__ tasks) {
  __shared__ float tile[kFeatRows * (kFeatCols + 1)];
  __shared__ unsigned long long s_f[kFeatRows], s_s[kFeatRows];
  __shared__ float s_alpha[kFeatRows], s_nrm[kFeatRows];
  const mot_feat_task T = tasks[blockIdx.y];
  const int i0 = blockIdx.x * kFeatRows;
  if (i0 >= T.n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = (T.n - i0 < kFeatRows) ? T.n - i0 : kFeatRows;
  if (tid < rows) {
    const int i = i0 + tid;
    s_f[tid] = static_cast<unsigned long long>(T.slot ? T.slot[i] : i) * T.ldf;
    s_s[tid] = static_cast<unsigned long long>(T.sidx ? T.sidx[i] : i) * T.lds;
    s_alpha[tid] = T.alpha_i ? T.alpha_i[i] : T.alpha;
  }
  __syncthreads();
  const bool ema = T.mode == 1 || T.mode == 3 || T.mode == 4;
  const bool single = T.d <= kFeatCols;
  constexpr int RS = kFeatCols + 1;
  float nn = 0.0f;
  // Round 5: rows of exactly 256 floats on 16-byte boundaries (every appearance feature of the device lifecycles) move as one float4 per lane — a
  // row is ONE load instruction of a wavefront instead of four, all of a wavefront's eight rows are in flight together, and the rows are written
  // back the same way. Element k = 4 * lane + c sits at tile column c * 64 + lane (conflict-free for the wavefronts' stores and for the 32 chains,
  // whose lanes are 257 words apart); the chain still adds the squares in the order k = 0, 1, 2, ...: same sums, same quotients.
  const bool vec = T.d == kFeatCols && ((T.ldf | T.lds) & 3) == 0 && ((reinterpret_cast<size_t>(T.src) | reinterpret_cast<size_t>(T.feat)) & 15) == 0;
  if (vec) {
    constexpr int kRowsPerWave = kFeatRows / 4;
    float4 v[kRowsPerWave], f[kRowsPerWave];
#pragma unroll
    for (int u = 0; u < kRowsPerWave; ++u) {
      const int r = wave + 4 * u;
      if (r < rows) {
        v[u] = reinterpret_cast<const float4*>(T.src + s_s[r])[lane];
        if (ema) f[u] = reinterpret_cast<const float4*>(T.feat + s_f[r])[lane];
      }
    }
#pragma unroll
    for (int u = 0; u < kRowsPerWave; ++u) {
      const int r = wave + 4 * u;
      if (r < rows) {
        float4 w = v[u];
        if (ema) {
          const float a = s_alpha[r];
          w.x = a * f[u].x + (1.0f - a) * w.x;  // botsort.cpp:163 / deepocsort.cpp:143
          w.y = a * f[u].y + (1.0f - a) * w.y;
          w.z = a * f[u].z + (1.0f - a) * w.z;
          w.w = a * f[u].w + (1.0f - a) * w.w;
        }
        float* trow = tile + r * RS + lane;
        trow[0] = w.x; trow[64] = w.y; trow[128] = w.z; trow[192] = w.w;
      }
    }
    __syncthreads();
    if (tid < rows) {
      const float* row = tile + tid * RS;
#pragma unroll 8
      for (int q = 0; q < kFeatCols / 4; ++q) {
        nn = __builtin_fmaf(row[q], row[q], nn);
        nn = __builtin_fmaf(row[64 + q], row[64 + q], nn);
        nn = __builtin_fmaf(row[128 + q], row[128 + q], nn);
        nn = __builtin_fmaf(row[192 + q], row[192 + q], nn);
      }
      const float nrm = sqrtf(nn);
      const bool go = (T.mode == 6) ? false : ((T.mode >= 4) ? (nrm > 1e-10f) : ((T.mode >= 2) ? (nrm > 1e-6f) : (nrm > 0.0f)));
      s_nrm[tid] = go ? nrm : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRowsPerWave; ++u) {
      const int r = wave + 4 * u;
      if (r < rows) {
        const float nrm = s_nrm[r];
        if (T.mode == 4 && nrm == 0.0f) continue;  // strongsort.cpp:176-179
        const float* trow = tile + r * RS + lane;
        float4 w{trow[0], trow[64], trow[128], trow[192]};
        if (nrm != 0.0f) { w.x = w.x / nrm; w.y = w.y / nrm; w.z = w.z / nrm; w.w = w.w / nrm; }
        reinterpret_cast<float4*>(T.feat + s_f[r])[lane] = w;
      }
    }
    return;
  }
  for (int c0 = 0; c0 < T.d; c0 += kFeatCols) {
    const int cols = (T.d - c0 < kFeatCols) ? T.d - c0 : kFeatCols;
    for (int r = wave; r < rows; r += 4) {
      const float* sp = T.src + s_s[r] + c0;
      const float* fp = T.feat + s_f[r] + c0;
      const float a = s_alpha[r];
#pragma unroll 4
      for (int k = lane; k < cols; k += 64) {
        float v = sp[k];
        if (ema) v = a * fp[k] + (1.0f - a) * v;  // botsort.cpp:163 / deepocsort.cpp:143
        tile[r * RS + k] = v;
      }
    }
    __syncthreads();
    if (tid < rows) {
      const float* row = tile + tid * RS;
#pragma unroll 8
      for (int k = 0; k < cols; ++k) nn = __builtin_fmaf(row[k], row[k], nn);
      if (c0 + cols >= T.d) {
        const float nrm = sqrtf(nn);
        const bool go = (T.mode == 6) ? false : ((T.mode >= 4) ? (nrm > 1e-10f) : ((T.mode >= 2) ? (nrm > 1e-6f) : (nrm > 0.0f)));
        s_nrm[tid] = go ? nrm : 0.0f;  // 0 = the row is stored as it is (mode 4: the stored row stays as it was)
      }
    }
    __syncthreads();
    for (int r = wave; r < rows; r += 4) {
      float* fp = T.feat + s_f[r] + c0;
      const float nrm = single ? s_nrm[r] : 0.0f;
      if (T.mode == 4 && single && nrm == 0.0f) continue;  // strongsort.cpp:176-179: the smoothed feature is only taken when it can be normalised
#pragma unroll 4
      for (int k = lane; k < cols; k += 64) {
        const float v = tile[r * RS + k];
        fp[k] = (nrm != 0.0f) ? v / nrm : v;
      }
    }
    __syncthreads();
  }
  if (!single) {  // long rows: the blended values were stored un-normalised; divide them now
    for (int r = wave; r < rows; r += 4) {
      const float nrm = s_nrm[r];
      if (nrm == 0.0f) continue;
      float* fp = T.feat + s_f[r];
      for (int k = lane; k < T.d; k += 64) fp[k] = fp[k] / nrm;
    }
  }
}

// ---- DeepOC-SORT's embedding term (deepocsort.cpp:294-346, 419-441) ----------------------------------------------------
// aw_stats: one wavefront per row (blockIdx.x < nd) or column of emb' = (iou <= 0 ? 0 : emb): its two largest VALUES (a value
// that occurs twice is both) -> the weight of that row / column.
__device__ __forceinline__ void top2_merge_f(float& m1, float& m2, float o1, float o2) {
  const float lo = (o1 < m1) ? o1 : m1;   // the smaller of the two maxima
  const float hi = (o1 < m1) ? m1 : o1;
  const float s = (o2 < m2) ? m2 : o2;    // the larger of the two seconds
  m1 = hi;
  m2 = (s < lo) ? lo : s;
}
__global__ void __launch_bounds__(64) deep_stats_kernel(const mot_deep_task* __restrict__ tasks) {
  const mot_deep_task T = tasks[blockIdx.y];
  if (T.aw_off) return;
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= T.nd + T.nt) return;
  const bool row = q < T.nd;
  const int idx = row ? q : q - T.nd, len = row ? T.nt : T.nd;
  const float ninf = -__builtin_huge_valf();
  float m1 = ninf, m2 = ninf;
  for (int k = lane; k < len; k += 64) {
    const size_t oe = row ? static_cast<size_t>(idx) * T.lde + k : static_cast<size_t>(k) * T.lde + idx;
    const size_t oi = row ? static_cast<size_t>(idx) * T.ldi + k : static_cast<size_t>(k) * T.ldi + idx;
    const float v = (T.iou[oi] <= 0.0f) ? 0.0f : T.emb[oe];
    if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) m2 = v;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float o1 = __shfl_xor(m1, d, 64), o2 = __shfl_xor(m2, d, 64);
    top2_merge_f(m1, m2, o1, o2);
  }
  if (lane == 0) {
    float wgt = 1.0f;  // fewer than two entries: the reference skips the row / column
    if (len >= 2) wgt = (m1 == 0.0f) ? 0.0f : 1.0f - mot::smax((m2 / m1) - T.aw_param, 0.0f) / (1.0f - T.aw_param);
    (row ? T.rw : T.cw)[idx] = wgt;
  }
}
__global__ void __launch_bounds__(256) deep_combine_kernel(const mot_deep_task* __restrict__ tasks) {
  const mot_deep_task T = tasks[blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (i >= T.nd || j >= T.nt) return;
  const float e = (T.iou[static_cast<size_t>(i) * T.ldi + j] <= 0.0f) ? 0.0f : T.emb[static_cast<size_t>(i) * T.lde + j];
  float fin;
  if (T.aw_off) fin = e * T.w;
  else {
    // w_emb = Constant(w); row(i) *= rw (or setZero); col(j) *= cw (or setZero); then w_emb .* emb
    const float wr = (T.nt >= 2) ? T.w * T.rw[i] : T.w;
    const float wc = (T.nd >= 2) ? wr * T.cw[j] : wr;
    fin = wc * e;
  }
  float* c = T.cost + static_cast<size_t>(i) * T.ldc + j;
  *c = *c - fin;  // -(iou + angle) - fin == -((iou + angle) + fin): negation and subtraction round alike
}

}  // namespace

namespace mot {
hipError_t launch_deep(const mot_deep_task* tasks, int ntasks, int max_nd, int max_nt, hipStream_t st) {
  if (ntasks <= 0 || max_nd <= 0 || max_nt <= 0) return hipSuccess;
  hipLaunchKernelGGL(deep_stats_kernel, dim3(max_nd + max_nt, ntasks), dim3(64), 0, st, tasks);
  hipLaunchKernelGGL(deep_combine_kernel, dim3((max_nt + 255) / 256, max_nd, ntasks), dim3(256), 0, st, tasks);
  return hipGetLastError();
}
hipError_t launch_iou(const mot_iou_task* tasks, int ntasks, int max_n, int max_m, bool iou_only, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  dim3 grid((max_m + kTile - 1) / kTile, (max_n + kTile - 1) / kTile, ntasks);
  if (iou_only) hipLaunchKernelGGL(iou_kernel<false>, grid, dim3(kThreads), 0, st, tasks);
  else hipLaunchKernelGGL(iou_kernel<true>, grid, dim3(kThreads), 0, st, tasks);
  return hipGetLastError();
}
hipError_t launch_ocsort(const mot_ocsort_task* tasks, int ntasks, int max_nd, int max_nt, bool iou_only, hipStream_t st) {
  if (ntasks <= 0 || max_nd <= 0 || max_nt <= 0) return hipSuccess;
  dim3 grid((max_nt + kTile - 1) / kTile, (max_nd + kTile - 1) / kTile, ntasks);
  if (iou_only) hipLaunchKernelGGL(ocsort_kernel<false>, grid, dim3(kThreads), 0, st, tasks);
  else hipLaunchKernelGGL(ocsort_kernel<true>, grid, dim3(kThreads), 0, st, tasks);
  return hipGetLastError();
}
hipError_t launch_feat(const mot_feat_task* tasks, int ntasks, int max_n, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return hipSuccess;
  dim3 grid((max_n + kFeatRows - 1) / kFeatRows, ntasks);
  hipLaunchKernelGGL(feat_kernel, grid, dim3(256), 0, st, tasks);
  return hipGetLastError();
}
}  // namespace mot
