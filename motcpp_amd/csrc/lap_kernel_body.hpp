// The exact assignment kernel's body (see lap_kernel.hip for the design notes and the launch logic). Included by the three translation units
// that instantiate its variants: lap_kernel.hip (one wavefront per problem, plain / BoT-SORT costs), lap_kernel_wide.hip (4 / 8 wavefronts per
// problem) and lap_kernel_general.hip (every association measure) — one file took six minutes to compile.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/motcpp_amd.h"
#include "lap_core.hpp"
#include "lap_cost.hpp"
#include "lap_sparse.hpp"

namespace mot {
struct LapDiag {
  long long (*scr)[36];   // [512][36]
  long long (*sum)[40];   // [2][40]
};
// what a translation unit exports for its variants: (threads, lds_mode, RPL, FLAVOR) -> launch / set the dynamic-LDS attribute
struct LapLaunchArgs {
  int threads, mode, rpl, flavor, grid;
  size_t lds;
  hipStream_t st;
  const mot_lap_task* tasks;
  int ntasks, check_status;
  const int* declined;
  int fs_lds;
  LapDiag diag;
};
hipError_t lap_narrow_attr(); bool lap_narrow_launch(const LapLaunchArgs& A);
hipError_t lap_wide_attr(); bool lap_wide_launch(const LapLaunchArgs& A);
hipError_t lap_general_attr(); bool lap_general_launch(const LapLaunchArgs& A);
}  // namespace mot

namespace {

constexpr int kScratch = 1024;  // DevGroup reduction scratch: 2 halves x 16 wavefronts x 32 bytes
constexpr int kFsLds = (mot::kFsWsInts * 4 + 15) & ~15;  // fast scratch of the parallel scan steps (matrix-cost launches only)
constexpr int kLdsBudget = 160 * 1024;

template <int kThreads, class Cost, class Work>
__device__ __forceinline__ int gate_and_solve(mot::DevGroup& g, const Cost& C, const mot_lap_task& T, const Work& W) {
  const int nr = T.n, nc = T.m, t = threadIdx.x;
  int path = 0;
  if (T.mode == MOT_LAP_GATE_MIN) {
    double mn = 1e300;
    for (int i = 0; i < nr; ++i) {
      const typename Cost::Row R = C.row(i);
      for (int j = t; j < nc; j += kThreads) { const double c = C.at(R, j); if (c < mn) mn = c; }
    }
    mn = g.reduce_min(mn);
    if (!(mn < static_cast<double>(T.gate))) path = 2;
  } else if (T.mode == MOT_LAP_OCSORT) {
    // a = (iou > gate); trivial one-to-one case iff max row sum == 1 and max col sum == 1 (ocsort.cpp:684-696)
    // one coalesced sweep: lane t owns columns t, t+T, ... and walks them down the rows; row hits go through atomics
    int max_row = 0, max_col = 0;
    for (int i = t; i < nr; i += kThreads) { W.x[i] = -1; W.fr[i] = 0; }
    g.sync();
    for (int j = t; j < nc; j += kThreads) {
      int c = 0, last = -1;
      for (int i = 0; i < nr; ++i)
        if (mot::gld(T.iou, static_cast<size_t>(i) * T.ldi + j) > T.gate) {
          ++c; last = i;
          mot::DevGroup::atomic_add(W.fr.raw(i), 1);
          mot::DevGroup::atomic_max(W.x.raw(i), j);
        }
      W.y[j] = (c == 1) ? last : -1;
      if (c > max_col) max_col = c;
    }
    g.sync();
    for (int i = t; i < nr; i += kThreads) {
      const int c = W.fr[i];
      if (c != 1) W.x[i] = -1;
      if (c > max_row) max_row = c;
    }
    max_row = g.reduce_max(max_row);
    max_col = g.reduce_max(max_col);
    g.sync();
    if (max_row == 1 && max_col == 1) path = 1;
  }
  if (path == 0) {
    const mot::LapDims P{nr, nc, static_cast<double>(T.thresh) / 2.0};
    mot::lap_solve(g, C, P, W);
    g.sync();
    for (int i = t; i < nr; i += kThreads) { const int v = W.x[i]; W.x[i] = (v >= nc) ? -1 : v; }
    for (int j = t; j < nc; j += kThreads) { const int v = W.y[j]; W.y[j] = (v >= nr) ? -1 : v; }
  } else if (path == 2) {
    for (int i = t; i < nr; i += kThreads) W.x[i] = -1;
    for (int j = t; j < nc; j += kThreads) W.y[j] = -1;
  }
  g.sync();
  for (int i = t; i < nr; i += kThreads) {
    const int xi = W.x[i];
    T.x[i] = xi;
    if (T.xval) {
      float v = 0.f;
      if (xi >= 0) v = T.iou ? mot::gld(T.iou, static_cast<size_t>(i) * T.ldi + xi) : static_cast<float>(C.at(i, xi));
      T.xval[i] = v;
    }
  }
  for (int j = t; j < nc; j += kThreads) T.y[j] = W.y[j];
  return path;
}

// lds_mode 4 = lean + d[] in LDS (20 B per extended row). lds_mode 3 = lean: duals v[] and y[] in LDS, everything else (x, free list, boxes) in global scratch / L2 —
// 12 B of LDS per extended row, so ~8 north-star-sized problems stay resident per CU.
// lds_mode (compile-time, so every pointer has a static address space — a run-time choice makes the compiler fall back
// to flat_* instructions for the LDS state, which costs hundreds of cycles per dependent access): 0 = solver state in global scratch, 1 = hot state + column boxes in LDS (row boxes
// in global scratch), 2 = hot state + column boxes + row boxes (+ the row bounds) in LDS, 5 = 2 + the shortest-path search's d / pred / cols / inv / tie / tmp / lst in
// LDS (32 B more per extended row: the launches behind the fast path, where per-problem latency is all that counts).
// second launch bound = wavefronts per SIMD the register allocator must leave room for: the solver is latency-bound per
// wavefront, throughput comes from co-resident ones (RPL 8: 2, i.e. <= 256 VGPRs; RPL 4: 3, <= 168; else whatever fits)
constexpr int lap_min_waves(int threads, int rpl, bool general) { return (threads > 64 || general) ? 1 : (rpl >= 8 ? 2 : (rpl >= 4 ? 3 : 4)); }
// FLAVOR of the on-the-fly cost: 0 plain IoU modes only, 1 + BoT-SORT's gated appearance term, 2 every association measure
// Diagnostics of the problems solved BEHIND the fast path (the ones the sparse solver declined): per-block scratch for lap_solve's
// cycle / event counters (mot_lap_task.prof layout, 36 entries), summed into g_behind[0] ([39] = problems), the slowest problem's
// own counters kept in g_behind[1] ([39] = its cycles). Read by mot_lap_behind_stats. Diagnostics only: launches of different HIP streams
// that run at the same time share the scratch rows (a block's counters can be mixed with another launch's in that rare overlap; nothing
// the solver computes depends on them).
// (round 5: the kernel's variants are compiled in three translation units — lap_kernel.hip, lap_kernel_wide.hip, lap_kernel_general.hip — so the
// two tables live in lap_kernel.hip and reach the kernels as arguments: LapDiag)
using mot::LapDiag;
template <int kThreads, int lds_mode, int RPL, int FLAVOR>
__device__ __forceinline__ void lap_one(const mot_lap_task& T, int check_status, int fs_lds, const LapDiag& DG) {
  constexpr bool GENERAL = FLAVOR == 2;
  constexpr bool PLAIN = FLAVOR == 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nr = T.n, nc = T.m, n = nr + nc;
  const int t = threadIdx.x;
  // the fast path (lap_sparse_kernel) ran first over the same tasks: problems it finished carry status 1
  if (check_status && *reinterpret_cast<const int*>(static_cast<const char*>(T.work) + mot::lap_task_scratch_bytes(nr, nc) - 16) == 1) return;
  if (nr <= 0 || nc <= 0) {
    for (int i = t; i < nr; i += kThreads) { T.x[i] = -1; if (T.xval) T.xval[i] = 0.f; }
    for (int j = t; j < nc; j += kThreads) T.y[j] = -1;
    if (T.info && t == 0) T.info[0] = 2;
    return;
  }
  if constexpr (PLAIN) {
    if (T.geom.a != nullptr && T.geom.mode >= MOT_COST_BOTSORT) {  // (BOTSORT, FUSE_IOU) MOT_LAP_F_PLAIN was a false promise: refuse loudly
      for (int i = t; i < nr; i += kThreads) { T.x[i] = -1; if (T.xval) T.xval[i] = 0.f; }
      for (int j = t; j < nc; j += kThreads) T.y[j] = -1;
      if (T.info && t == 0) T.info[0] = -1;
      return;
    }
  }
  mot::DevGroup g(smem);
  // global scratch layout: [hot (mode 0 only)] [cold] [row boxes 5*nr floats] [col boxes 6*nc floats]
  char* gw = static_cast<char*>(T.work);
  const size_t hot_b = (mot::lap_hot_bytes(n) + 15) & ~size_t(15), cold_b = (mot::lap_cold_bytes(n) + 15) & ~size_t(15);
  constexpr int kVS = (lds_mode == 0) ? mot::kMemGlobal : mot::kMemLds;  // v, y
  // lds_mode 5 = mode 2 + the shortest-path search's arrays (d, pred, cols, inv, tie, tmp, lst: 32 B per extended row) in LDS: the launches behind
  // the fast path, where a handful of problems run and the sub-batch waits for the slowest
  constexpr bool kFull = lds_mode == 2 || lds_mode == 5;
  constexpr int kXS = kFull ? mot::kMemLds : mot::kMemGlobal;  // x, free list
  constexpr int kRS = kFull ? mot::kMemLds : mot::kMemGlobal;  // row boxes
  constexpr int kDS = (lds_mode == 4 || lds_mode == 5 || lds_mode == 6) ? mot::kMemLds : mot::kMemGlobal;  // shortest-path distances
  constexpr int kBS = kFull ? mot::kMemLds : mot::kMemGlobal;  // row bounds (on-the-fly costs only)
  constexpr int kCS = (lds_mode == 5) ? mot::kMemLds : mot::kMemGlobal;  // pred, cols, inv, tie
  // lds_mode 6 (round 5, wide matrix problems: OC-SORT's first association) = mode 4 + cols[] / inv[] in LDS, they and y[] as 16-bit entries (22 B per
  // extended row), the rows' list lengths as bytes; pred[] stays in global memory (written by the steps, read when a search is over)
  constexpr bool k16 = lds_mode == 6;
  using IdxT = std::conditional_t<k16, short, int>;
  mot::LapWorkT<kVS, kXS, kDS, kBS, kCS, k16 ? mot::kMemLds : kCS, IdxT, k16 ? mot::kMemLds : mot::kMemGlobal, mot::kMemGlobal> W;
  char* lds = smem + kScratch;
  if (fs_lds) {  // the launch reserved the fast scratch: row lists are usable by the tasks that bring the memory for them
    if (T.rowlist != nullptr && T.geom.a == nullptr) { mot::lap_carve_rowlist(W, T.rowlist, nr); W.fsw.p = reinterpret_cast<int*>(lds); }
    lds += kFsLds;
  }
  if constexpr (kFull) { mot::lap_carve_hot(W, lds, n); lds += hot_b; }
  else mot::lap_carve_hot(W, gw, n);
  if constexpr (lds_mode == 3 || lds_mode == 4 || lds_mode == 6) {  // lean: only the per-column duals and column->row map in LDS (12 B per extended row)
    W.v.p = reinterpret_cast<double*>(lds);
    if constexpr (!k16) W.y.p = reinterpret_cast<int*>(lds + sizeof(double) * static_cast<size_t>(n));
  }
  mot::lap_carve_cold(W, gw + hot_b, n);
  if constexpr (lds_mode == 4)  // + the distances of the shortest-path search (wide matrix problems: every scan step reads and writes them)
    W.d.p = reinterpret_cast<double*>(lds + ((12 * static_cast<size_t>(n) + 15) & ~size_t(15)));
  if constexpr (k16) {
    char* q = lds;
    W.v.p = reinterpret_cast<double*>(q); q += 8 * static_cast<size_t>(n);
    W.d.p = reinterpret_cast<double*>(q); q += 8 * static_cast<size_t>(n);
    W.y.p = reinterpret_cast<short*>(q); q += 2 * static_cast<size_t>(n);
    W.cols.p = reinterpret_cast<short*>(q); q += 2 * static_cast<size_t>(n);
    W.inv.p = reinterpret_cast<short*>(q); q += 2 * static_cast<size_t>(n);
    W.lst16.p = reinterpret_cast<unsigned short*>(smem + kScratch);  // over the fast scratch (fs_lds is set for every launch of this mode)
    if (W.rl_cnt.p != nullptr) {  // (the row lists are in use)
      W.rl_n.p = reinterpret_cast<unsigned char*>(q);
      W.ycost.p = W.rmin.p;  // nc floats of the cold scratch that only the on-the-fly costs use otherwise
    }
  }
  const bool behind_diag = check_status != 0 && T.prof == nullptr && blockIdx.x < 512;
  W.cyc = behind_diag ? DG.scr[blockIdx.x] : T.prof;
  W.cyc_ext = true;  // (mot_lap_task.prof holds 36 entries)
  int path;
  if (T.geom.a != nullptr) {
    float* gbox = reinterpret_cast<float*>(gw + hot_b + cold_b);
    // column boxes: global scratch (they are copied into registers below; memory only backs arbitrary-column reads);
    // row boxes: LDS in mode 2 (one uniform read per row pass), global scratch otherwise
    float* cp = gbox + 5 * nr;
    float* cf = cp + 5 * nc;
    float* rp;
    if constexpr (kFull) {
      rp = reinterpret_cast<float*>(lds);
      char* q = lds + ((20 * static_cast<size_t>(nr) + 16 + 15) & ~size_t(15));
      W.rlb.p = reinterpret_cast<double*>(q);  // behind the row boxes (8 B per real row)
      if constexpr (lds_mode == 5) {
        q += (8 * static_cast<size_t>(nr) + 15) & ~size_t(15);
        W.d.p = reinterpret_cast<double*>(q); q += 8 * static_cast<size_t>(n);
        W.pred.p = reinterpret_cast<int*>(q); q += 4 * static_cast<size_t>(n);
        W.cols.p = reinterpret_cast<int*>(q); q += 4 * static_cast<size_t>(n);
        W.inv.p = reinterpret_cast<int*>(q); q += 4 * static_cast<size_t>(n);
        W.tie.p = reinterpret_cast<int*>(q); q += 4 * static_cast<size_t>(n);
        W.tmp.p = reinterpret_cast<int*>(q); q += 4 * static_cast<size_t>(n);  // (_find_dense's record flags and compacted positions: a store
        W.lst.p = reinterpret_cast<int*>(q);                                    //  followed by a dependent load, several times per call)
      }
    } else rp = gbox;
    const mot_iou_task& G = T.geom;
    for (int i = t; i < nr; i += kThreads) {
      const int gi = G.aidx ? G.aidx[i] : i;
      float b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { b[k] = G.a[static_cast<size_t>(k) * G.lda + gi]; rp[k * nr + i] = b[k]; }
      rp[4 * nr + i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    for (int j = t; j < nc; j += kThreads) {
      const int gj = G.bidx ? G.bidx[j] : j;
      float b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { b[k] = G.b[static_cast<size_t>(k) * G.ldb + gj]; cp[k * nc + j] = b[k]; }
      cp[4 * nc + j] = (b[2] - b[0]) * (b[3] - b[1]);
      cf[j] = G.bconf ? G.bconf[gj] : 0.0f;
    }
    g.sync();
    mot::IouCostT<RPL, kRS, GENERAL, PLAIN> C;
    C.rows = mot::BoxPlanes<kRS>{rp, nr};
    C.cols = mot::BoxPlanes<mot::kMemGlobal>{cp, nc};
    C.conf = G.bconf ? cf : nullptr;
    C.prm = mot::CostParams{G.mode, G.prox_thresh, G.app_thresh, G.fuse, G.emb != nullptr, G.emb == nullptr && G.lde < 0, G.assoc, G.frame_diag};
    C.emb = G.emb;
    C.lde = G.lde;
    C.load_owned(t, kThreads, nc);
    path = gate_and_solve<kThreads>(g, C, T, W);
  } else {
    const mot::MatrixCost C{T.cost, T.ldc};
    path = gate_and_solve<kThreads>(g, C, T, W);
  }
  if (T.info && t == 0) T.info[0] = path;
  if (behind_diag && t == 0) {
    const long long* c = DG.scr[blockIdx.x];
    const long long tot = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 36; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(&DG.sum[0][k]), static_cast<unsigned long long>(c[k]));
    atomicAdd(reinterpret_cast<unsigned long long*>(&DG.sum[0][39]), 1ull);
    const long long prev = static_cast<long long>(atomicMax(reinterpret_cast<unsigned long long*>(&DG.sum[1][39]), static_cast<unsigned long long>(tot)));
    if (tot > prev) for (int k = 0; k < 36; ++k) DG.sum[1][k] = c[k];
  }
}

template <int kThreads, int lds_mode, int RPL, int FLAVOR>
__global__ void __launch_bounds__(kThreads, lap_min_waves(kThreads, RPL, FLAVOR == 2)) lap_kernel(const mot_lap_task* __restrict__ tasks, int ntasks, int check_status, const int* declined, int fs_lds, LapDiag DG) {
  // behind the fast path: its count of declined problems; usually zero, and then there is nothing to look for
  if (check_status && declined != nullptr && *declined == 0) return;
  // A handful of problems the sparse solver declined, one wavefront each, and their whole sub-batch waits for the slowest: the other
  // HIP streams' kernels fill the same SIMDs (four or five wavefronts each), so without help this wavefront gets a fraction of the issue
  // slots (measured at the north-star shape: 3.2 ms alone on the GPU, 8 ms on average inside the benchmark). check_status 1 = raise the
  // wavefront's issue priority; 2 = leave it (MOT_LAP_BEHIND_PRIO=0, for A/B measurements).
  if (check_status == 1) __builtin_amdgcn_s_setprio(3);
  // behind the fast path the grid is smaller than the task array: a block walks its share of it and solves what is left
  for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
    const mot_lap_task T = tasks[task];
    lap_one<kThreads, lds_mode, RPL, FLAVOR>(T, check_status, fs_lds, DG);
    if (task + static_cast<int>(gridDim.x) < ntasks) __syncthreads();  // the LDS state of this problem is dead before the next one starts
  }
}


}  // namespace

#define MOT_LAP_TU_EXPORTS(NAME, VARIANTS)                                                                                         \
  namespace mot {                                                                                                                  \
  hipError_t NAME##_attr() {                                                                                                       \
    hipError_t e = hipSuccess;                                                                                                     \
    VARIANTS(MOT_LAP_ATTR_ONE)                                                                                                     \
    return e;                                                                                                                      \
  }                                                                                                                                \
  bool NAME##_launch(const LapLaunchArgs& A) {                                                                                     \
    VARIANTS(MOT_LAP_TRY_ONE)                                                                                                      \
    return false;                                                                                                                  \
  }                                                                                                                                \
  }
#define MOT_LAP_ATTR_ONE(T, M, R, G)                                                                                               \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_kernel<T, M, R, G>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
#define MOT_LAP_TRY_ONE(T, M, R, G)                                                                                                \
  if (A.threads == T && A.mode == M && A.rpl == R && A.flavor == G) {                                                              \
    hipLaunchKernelGGL((lap_kernel<T, M, R, G>), dim3(A.grid), dim3(T), A.lds, A.st, A.tasks, A.ntasks, A.check_status, A.declined, A.fs_lds, A.diag); \
    return true;                                                                                                                   \
  }
