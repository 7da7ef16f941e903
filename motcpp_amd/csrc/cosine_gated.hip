// Gated appearance distances for BoT-SORT's associations (src/trackers/botsort.cpp:433-466, :591-623).
//
// The reference builds the whole n x m cosine-distance matrix (utils::embedding_distance, src/utils/matching.cpp:67-92), halves it, and then
// FORCES every entry to 1 whose pair is masked by the proximity test `iou_distance > proximity_thresh` (botsort.cpp:439-447) before taking
// min(iou_dists, emb_dists). The assignment solvers here recompute a pair's cost on the fly (cost_math.hpp::cost_from_iou) and read the
// appearance matrix only for pairs that pass that test — for a pool of 861 tracks against 450 detections that is about one entry per track
// of 387 k. This kernel therefore writes exactly those entries: it repeats the proximity test with the solvers' own arithmetic (iou_pair on the
// boxes and areas as the solvers stage them) and evaluates the cosine distance of the pairs that pass, each as the k-ordered fmaf chain the
// fp32 MFMA kernel (cosine_mfma.hip) and the CPU oracle compute — the same bits, so the solvers see the same costs and the dense matrix
// (0.8 GB written per launch of 512 north-star-sized problems, 0.62 ms on the MFMA) is never produced. Entries of pairs that fail the test are
// left as they are: no solver reads them (cost_from_iou evaluates emb_at() only when !far).
//
// Workgroup = 256 threads = a strip of a problem's rows: the column boxes sit in LDS, a lane tests its columns against one row at a time
// (disjoint boxes cost a dozen instructions: iou_pair skips the division when no lane of the wavefront intersects), pairs that pass go to a
// list in LDS, and then one lane per listed pair runs the three chains (a.b, a.a, b.b) over float4 loads of the two feature rows. A strip
// whose list overflows (degenerate input: thousands of coincident boxes) evaluates its pairs in place instead, one at a time.
#include <hip/hip_runtime.h>

#include "../../include/motcpp_amd.h"
#include "cost_math.hpp"

namespace {

constexpr int kGcThreads = 256;
constexpr int kGcRows = 128;    // row boxes staged at a time
constexpr int kGcList = 4096;   // pairs a strip may queue

// the cosine distance of one pair: matching.cpp:83-90 with the k-ordered chains of cosine_mfma.hip (norms included)
__device__ __forceinline__ float pair_cosine(const float* __restrict__ pa, const float* __restrict__ pb, int d, bool vec) {
  float dp = 0.0f, na = 0.0f, nb = 0.0f;
  if (vec) {
    const float4* a4 = reinterpret_cast<const float4*>(pa);
    const float4* b4 = reinterpret_cast<const float4*>(pb);
    const int q = d >> 2;
#pragma unroll 4
    for (int k = 0; k < q; ++k) {  // (four pairs of 16-byte loads in flight per lane; sixteen measured slower: 0.167 against 0.143 ms per launch at C3)
      const float4 a = a4[k], b = b4[k];
      dp = __builtin_fmaf(a.x, b.x, dp); na = __builtin_fmaf(a.x, a.x, na); nb = __builtin_fmaf(b.x, b.x, nb);
      dp = __builtin_fmaf(a.y, b.y, dp); na = __builtin_fmaf(a.y, a.y, na); nb = __builtin_fmaf(b.y, b.y, nb);
      dp = __builtin_fmaf(a.z, b.z, dp); na = __builtin_fmaf(a.z, a.z, na); nb = __builtin_fmaf(b.z, b.z, nb);
      dp = __builtin_fmaf(a.w, b.w, dp); na = __builtin_fmaf(a.w, a.w, na); nb = __builtin_fmaf(b.w, b.w, nb);
    }
  } else {
    for (int k = 0; k < d; ++k) {
      const float a = pa[k], b = pb[k];
      dp = __builtin_fmaf(a, b, dp); na = __builtin_fmaf(a, a, na); nb = __builtin_fmaf(b, b, nb);
    }
  }
  const float sim = dp / (sqrtf(na) * sqrtf(nb) + 1e-10f);
  const float v = 1.0f - sim;
  return (0.0f < v) ? v : 0.0f;  // std::max(0.0f, v)
}

__global__ void __launch_bounds__(kGcThreads) embed_gated_kernel(const mot_cos_task* __restrict__ cos, const mot_lap_task* __restrict__ lap, int lap_stride,
                                                                 int m_pad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* cb = smem;                                      // [5][m_pad] column boxes x1 y1 x2 y2 area
  float* rb = cb + 5 * static_cast<size_t>(m_pad);       // [kGcRows][8]: x1 y1 x2 y2 area (one 16-byte broadcast read brings a row's box)
  unsigned* list = reinterpret_cast<unsigned*>(rb + 8 * kGcRows);  // [kGcList] (row << 16 | column), unsigned: rows up to 65535
  __shared__ int s_count;
  const mot_cos_task C = cos[blockIdx.y];
  const int n = C.n, m = C.m;
  if (n <= 0 || m <= 0) return;
  const int per = (n + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
  const int r0 = static_cast<int>(blockIdx.x) * per;
  const int r1 = (r0 + per < n) ? r0 + per : n;
  if (r0 >= n) return;
  const mot_iou_task& G = lap[static_cast<size_t>(blockIdx.y) * lap_stride].geom;
  const int t = threadIdx.x;
  // every pair is wanted when the task's cost is not the gated one (FUSE_IOU reads the whole matrix), or when there are no boxes to test
  const float prox = G.prox_thresh;
  // (... or when a pair of disjoint boxes — IoU +0, distance 1 — passes the test: the quick rejection below would be wrong then)
  const bool all = G.a == nullptr || G.mode != MOT_COST_BOTSORT || m > m_pad || n > 65535 || m > 65535 || !(1.0f > prox);
  if (!all) {
    for (int j = t; j < m; j += kGcThreads) {
      const int gj = G.bidx ? G.bidx[j] : j;
      float b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { b[k] = G.b[static_cast<size_t>(k) * G.ldb + gj]; cb[k * m_pad + j] = b[k]; }
      cb[4 * m_pad + j] = (b[2] - b[0]) * (b[3] - b[1]);
    }
  }
  if (t == 0) s_count = 0;
  const bool vec = ((C.lda | C.ldb | C.d) & 3) == 0 && ((reinterpret_cast<size_t>(C.a) | reinterpret_cast<size_t>(C.b)) & 15) == 0;
  auto row_ptr = [&](int i) { return C.a + static_cast<size_t>(C.aidx ? C.aidx[i] : i) * C.lda; };
  auto col_ptr = [&](int j) { return C.b + static_cast<size_t>(C.bidx ? C.bidx[j] : j) * C.ldb; };
  if (all) {  // (not BoT-SORT's gated cost: the plain matrix, one pair per lane)
    const long long total = static_cast<long long>(r1 - r0) * m;
    for (long long p = t; p < total; p += kGcThreads) {
      const int i = r0 + static_cast<int>(p / m), j = static_cast<int>(p % m);
      C.out[static_cast<size_t>(i) * C.ldo + j] = pair_cosine(row_ptr(i), col_ptr(j), C.d, vec);
    }
    return;
  }
  // pass(i, j): the solvers' proximity test on pair (i, j) — cost_from_iou's `far` (botsort.cpp:439), NaN included
  bool overflow = false;
  for (int pass_no = 0; pass_no < 2; ++pass_no) {  // 0: queue the pairs; 1 (only after an overflow): evaluate them in place
    for (int c0 = r0; c0 < r1; c0 += kGcRows) {
      const int rows = (r1 - c0 < kGcRows) ? r1 - c0 : kGcRows;
      __syncthreads();  // (the previous rows' boxes are no longer read; the column boxes are staged)
      if (t < rows) {
        const int gi = G.aidx ? G.aidx[c0 + t] : c0 + t;
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = G.a[static_cast<size_t>(k) * G.lda + gi]; rb[t * 8 + k] = a[k]; }
        rb[t * 8 + 4] = (a[2] - a[0]) * (a[3] - a[1]);
      }
      __syncthreads();
      for (int j0 = 0; j0 < m; j0 += kGcThreads) {
        const int j = j0 + t;
        const bool jv = j < m;
        const int jj = jv ? j : 0;
        const float b[4] = {cb[jj], cb[m_pad + jj], cb[2 * m_pad + jj], cb[3 * m_pad + jj]};
        const float barea = cb[4 * m_pad + jj];
        for (int r = 0; r < rows; ++r) {
          const float4 a4 = *reinterpret_cast<const float4*>(rb + r * 8);
          // Boxes that do not overlap strictly (or hold a NaN where it matters) have w == 0 or h == 0 in iou_pair: IoU +0, distance 1 > prox.
          // Almost every (row, 64 columns) group is like that: four compares and no arithmetic.
          const bool touch = a4.x < b[2] && a4.z > b[0] && a4.y < b[3] && a4.w > b[1];
          if (__builtin_amdgcn_ballot_w64(touch) == 0) continue;
          const float a[4] = {a4.x, a4.y, a4.z, a4.w};
          const float iou = mot::iou_pair(a, rb[r * 8 + 4], b, barea);
          const float dist = 1.0f - iou;
          const bool far = dist > prox;
          if (jv && !far) {
            const int i = c0 + r;
            if (pass_no == 0) {
              const int pos = atomicAdd(&s_count, 1);
              if (pos < kGcList) list[pos] = (static_cast<unsigned>(i) << 16) | static_cast<unsigned>(j);
            } else {
              C.out[static_cast<size_t>(i) * C.ldo + j] = pair_cosine(row_ptr(i), col_ptr(j), C.d, vec);
            }
          }
        }
      }
    }
    if (pass_no == 1) return;
    __syncthreads();
    overflow = s_count > kGcList;
    if (!overflow) break;
  }
  const int P = s_count;
  for (int p = t; p < P; p += kGcThreads) {
    const int e = list[p];
    const int i = static_cast<int>(e >> 16), j = static_cast<int>(e & 0xffffu);
    C.out[static_cast<size_t>(i) * C.ldo + j] = pair_cosine(row_ptr(i), col_ptr(j), C.d, vec);
  }
}

}  // namespace

namespace mot {
// cos[s] (n x m, features) is paired with lap[s * lap_stride] (its geom: the boxes, the cost mode and the proximity threshold of the
// association that reads cos[s].out). max_n / max_m bound the tasks' sizes as in launch_embed.
hipError_t launch_embed_gated(const mot_cos_task* cos, const mot_lap_task* lap, int lap_stride, int ntasks, int max_n, int max_m, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0 || max_m <= 0) return hipSuccess;
  const int m_pad = (max_m + 3) & ~3;
  const size_t lds = sizeof(float) * (5 * static_cast<size_t>(m_pad) + 8 * kGcRows) + sizeof(int) * kGcList;
  if (lds > 150 * 1024) return hipErrorInvalidValue;  // (the callers fall back to launch_embed)
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_gated_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  // strips of ~128 rows, at least enough workgroups to fill the chip a few times over
  int split = (max_n + kGcRows - 1) / kGcRows;
  if (split < 1) split = 1;
  while (split < 8 && static_cast<long long>(split) * ntasks < 2048 && split * 32 < max_n) split *= 2;
  hipLaunchKernelGGL(embed_gated_kernel, dim3(split, ntasks), dim3(kGcThreads), lds, st, cos, lap, lap_stride, m_pad);
  return hipGetLastError();
}
}  // namespace mot
