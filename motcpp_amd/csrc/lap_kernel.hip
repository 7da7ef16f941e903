// Linear-assignment kernel for gfx950: the exact lapjv restatement of lap_core.hpp, ONE WAVEFRONT per problem
// by default. Measured on MI355X a row pass of lapjv costs ~4.5k cycles of serial overhead when a problem is spread
// over 4 wavefronts (cross-wavefront merges through LDS + two barriers + dependent scalar chains) almost regardless
// of its size; with one wavefront per problem there are no barriers and no LDS merges (reductions are DPP-only),
// and — because only the HOT solver state (duals, assignments, free list: 20 bytes per extended row) plus the
// staged boxes sit in LDS — 4 to 8 problems are resident per CU, one or two per SIMD, hiding each other's
// latencies. The COLD arrays of the general shortest-path search live in global scratch. Large problems with too
// few instances to fill the chip anyway (n+m > 3072 and < 512 problems) use 4 wavefronts per problem instead.
//
// Two cost sources:
//  * geometry (task.geom.a != NULL): IoU-family costs are recomputed on the fly from the row/column boxes staged
//    once (lap_cost.hpp) — the N x M matrix is never written or read; a problem's HBM traffic is its boxes + results;
//  * matrix (task.cost): a materialised float matrix, rows read by coalesced loads (OC-SORT's IoU+direction cost,
//    or any caller-supplied cost).
// Throughput comes from many problems in flight (grid = problems: streams x stages); one problem is
// dependency-bound by construction (lapjv's sequential rows), so this kernel is reported against the HBM roofline
// only because the brief asks for it — its real bound is issue latency of the serial row passes.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <mutex>

#include "../../include/motcpp_amd.h"
#include "lap_core.hpp"
#include "lap_cost.hpp"
#include "lap_sparse.hpp"

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
__device__ long long g_behind_scr[512][36];
__device__ long long g_behind[2][40];
template <int kThreads, int lds_mode, int RPL, int FLAVOR>
__device__ __forceinline__ void lap_one(const mot_lap_task& T, int check_status, int fs_lds) {
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
  constexpr int kDS = (lds_mode == 4 || lds_mode == 5) ? mot::kMemLds : mot::kMemGlobal;  // shortest-path distances
  constexpr int kBS = kFull ? mot::kMemLds : mot::kMemGlobal;  // row bounds (on-the-fly costs only)
  constexpr int kCS = (lds_mode == 5) ? mot::kMemLds : mot::kMemGlobal;  // pred, cols, inv, tie
  mot::LapWorkT<kVS, kXS, kDS, kBS, kCS> W;
  char* lds = smem + kScratch;
  if (fs_lds) {  // the launch reserved the fast scratch: row lists are usable by the tasks that bring the memory for them
    if (T.rowlist != nullptr && T.geom.a == nullptr) { mot::lap_carve_rowlist(W, T.rowlist, nr); W.fsw.p = reinterpret_cast<int*>(lds); }
    lds += kFsLds;
  }
  if constexpr (kFull) { mot::lap_carve_hot(W, lds, n); lds += hot_b; }
  else mot::lap_carve_hot(W, gw, n);
  if constexpr (lds_mode == 3 || lds_mode == 4) {  // lean: only the per-column duals and column->row map in LDS (12 B per extended row)
    W.v.p = reinterpret_cast<double*>(lds);
    W.y.p = reinterpret_cast<int*>(lds + sizeof(double) * static_cast<size_t>(n));
  }
  mot::lap_carve_cold(W, gw + hot_b, n);
  if constexpr (lds_mode == 4)  // + the distances of the shortest-path search (wide matrix problems: every scan step reads and writes them)
    W.d.p = reinterpret_cast<double*>(lds + ((12 * static_cast<size_t>(n) + 15) & ~size_t(15)));
  const bool behind_diag = check_status != 0 && T.prof == nullptr && blockIdx.x < 512;
  W.cyc = behind_diag ? g_behind_scr[blockIdx.x] : T.prof;
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
    const long long* c = g_behind_scr[blockIdx.x];
    const long long tot = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 36; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(&g_behind[0][k]), static_cast<unsigned long long>(c[k]));
    atomicAdd(reinterpret_cast<unsigned long long*>(&g_behind[0][39]), 1ull);
    const long long prev = static_cast<long long>(atomicMax(reinterpret_cast<unsigned long long*>(&g_behind[1][39]), static_cast<unsigned long long>(tot)));
    if (tot > prev) for (int k = 0; k < 36; ++k) g_behind[1][k] = c[k];
  }
}

template <int kThreads, int lds_mode, int RPL, int FLAVOR>
__global__ void __launch_bounds__(kThreads, lap_min_waves(kThreads, RPL, FLAVOR == 2)) lap_kernel(const mot_lap_task* __restrict__ tasks, int ntasks, int check_status, const int* declined, int fs_lds) {
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
    lap_one<kThreads, lds_mode, RPL, FLAVOR>(T, check_status, fs_lds);
    if (task + static_cast<int>(gridDim.x) < ntasks) __syncthreads();  // the LDS state of this problem is dead before the next one starts
  }
}

}  // namespace

namespace mot {
size_t lap_scratch_bytes(int n, int m) { return lap_task_scratch_bytes(n, m); }
size_t lap_rowlist_scratch_bytes(int n) { return lap_rowlist_bytes(n); }
hipError_t launch_lap_sparse(const mot_lap_task* tasks, int ntasks, int max_n, int max_m, bool plain_costs, int* declined, hipStream_t st,
                             int hint_n, int hint_m, int active_tasks);
hipError_t lap_behind_stats(long long* out80, bool reset, hipStream_t st) {
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (out80) { e = hipMemcpyFromSymbol(out80, HIP_SYMBOL(g_behind), sizeof(long long) * 80); if (e != hipSuccess) return e; }
  if (reset) { static const long long z[80] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_behind), z, sizeof(z)); }
  return e;
}

namespace {
// one counter per launch in flight (the sparse kernel counts the problems it declines, the exact kernel reads it): a ring of
// device ints per device, handed out round-robin — far more slots than launches can be in flight at once
constexpr int kDeclSlots = 4096;
int* decl_ring(int dev_slot) {
  static int* ring[64] = {};
  if (!ring[dev_slot]) { if (hipMalloc(reinterpret_cast<void**>(&ring[dev_slot]), sizeof(int) * kDeclSlots) != hipSuccess) ring[dev_slot] = nullptr; }
  return ring[dev_slot];
}
}  // namespace

// Threads per problem: one wavefront (no barriers, no LDS merges; 4-8 problems co-resident per CU) unless the problem is
// large AND there are too few problems to fill the chip anyway, where 4 wavefronts cut the latency of a row pass.
// mid_event: recorded between the sparse solver and the exact kernel (profiling). try_fast = false: skip the certified sparse solver (a caller that saw it decline every problem of its previous launches — OC-SORT's
// first association once quirk Q4 has put duplicated tracks into every frame — saves its enumeration; results are the exact
// kernel's either way). declined_out: the device counter of the problems the sparse solver declined in THIS launch (valid once the
// stream has passed the launch; the slot is recycled after kDeclSlots further launches). prezeroed: a device int of the caller's, zero when
// the stream reaches this launch, used as that counter instead of a ring slot + memset. active_tasks (0: all): how many of the tasks are
// not empty, when the caller knows — a launch with a handful of problems is tuned for latency (more wavefronts per problem).
hipError_t launch_lap(const mot_lap_task* tasks, int ntasks, int max_n, int max_m, bool geom, bool general_assoc, bool plain_costs,
                      hipStream_t st, int hint_n, int hint_m, bool try_fast, int** declined_out, hipEvent_t mid_event, int* prezeroed, int active_tasks) {
  if (declined_out) *declined_out = nullptr;
  if (ntasks <= 0) return hipSuccess;
  // fast path first (not for the general association measures: there a pair that does not intersect has no constant cost)
  const bool fast = !general_assoc && try_fast;
  int* declined = nullptr;
  if (fast && prezeroed) {  // the caller's own counter, cleared by a kernel of its own in front of this launch: no memset on the stream
    declined = prezeroed;
    if (declined_out) *declined_out = declined;
    hipError_t e = launch_lap_sparse(tasks, ntasks, max_n, max_m, plain_costs, declined, st, hint_n, hint_m, active_tasks);
    if (e != hipSuccess) return e;
    if (mid_event) { e = hipEventRecord(mid_event, st); if (e != hipSuccess) return e; }
  } else if (fast) {
    static std::mutex ring_mu;
    static unsigned next_slot[64] = {};
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    const int ds = (dev0 >= 0 && dev0 < 64) ? dev0 : 0;
    {
      std::lock_guard<std::mutex> lk(ring_mu);
      int* ring = decl_ring(ds);
      if (!ring) return hipErrorOutOfMemory;
      declined = ring + (next_slot[ds]++ % kDeclSlots);
    }
    hipError_t e = hipMemsetAsync(declined, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    if (declined_out) *declined_out = declined;
    e = launch_lap_sparse(tasks, ntasks, max_n, max_m, plain_costs, declined, st, hint_n, hint_m, active_tasks);
    if (e != hipSuccess) return e;
    if (mid_event) { e = hipEventRecord(mid_event, st); if (e != hipSuccess) return e; }  // (profiling: the sparse kernel alone ends here)
  }
  const size_t n = max_n > 0 ? max_n : 1, m = max_m > 0 ? max_m : 1, nm = n + m;
  const size_t hot = (lap_hot_bytes(nm) + 15) & ~size_t(15);
  const int fs_lds = (!geom && n * m >= 16384) ? 1 : 0;  // matrix costs of some size: room for the parallel scan steps' scratch
  const size_t fsb = fs_lds ? kFsLds : 0;
  const size_t b2 = kScratch + fsb + hot + (geom ? 28 * n + 48 : 0);  // full hot state (+ row boxes and row bounds) in LDS
  const size_t b3 = kScratch + fsb + 12 * nm + 16;                    // lean: duals + y
  const size_t b4 = kScratch + fsb + 20 * nm + 32;                    // lean + distances
  int mode;
  size_t lds;
  if (b2 <= 18 * 1024) { mode = 2; lds = b2; }             // >= 8 problems per CU
  else if (b3 <= 40 * 1024) { mode = 3; lds = b3; }        // >= 4 (north-star: 8) problems per CU
  else if (b2 <= static_cast<size_t>(kLdsBudget) - 1024) { mode = 2; lds = b2; }
  else { mode = 0; lds = kScratch + fsb; }
  // Behind the fast path the exact kernel sees the few problems the sparse solver declined, and the launch lasts as long as its
  // slowest problem: four wavefronts per problem cut that latency (MOT_LAP_BEHIND_T=64 keeps one wavefront per problem).
  static const int behind_t = std::getenv("MOT_LAP_BEHIND_T") ? std::atoi(std::getenv("MOT_LAP_BEHIND_T")) : 64;
  const bool behind = fast && behind_t == 256 && n * m >= 65536 && nm <= 3072;
  // (matrix costs of some size: any number of problems — with 64 threads the row lists' parallel scan steps are not available; geometry
  // launches keep the old bound on the problem count, so that large batches stay on the one-wavefront register-cached variants)
  const bool wide = (nm > 3072 && (ntasks < 512 || fs_lds)) || behind;
  // mode 4 (distances in LDS for the scan steps) exists for cost flavour 1 only
  if (fs_lds && wide && !general_assoc && b4 <= static_cast<size_t>(kLdsBudget) - 1024) { mode = 4; lds = b4; }
  // lane-owned column boxes in registers: one wavefront per problem, <= 8 real columns per lane
  int rpl = 0;
  if (geom && !wide && !general_assoc) rpl = (m <= 256) ? 4 : (m <= 512 ? 8 : 0);
  // the dynamic-LDS attribute is per device: set once for each device this process launches on
  static std::mutex attr_mu;
  static bool attr_set_dev[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  const int dev_slot = (dev_id >= 0 && dev_id < 64) ? dev_id : 0;
  // behind the fast path almost every block would find its problem finished: launch fewer and let them walk the task array
  const int grid = (fast && ntasks > 512) ? 512 : ntasks;
#define MOT_LAP_VARIANTS(X) X(64, 0, 0, 1) X(64, 2, 0, 1) X(64, 3, 0, 1) X(64, 0, 4, 1) X(64, 2, 4, 1) X(64, 3, 4, 1) \
                            X(64, 0, 8, 1) X(64, 2, 8, 1) X(64, 3, 8, 1) X(256, 0, 0, 1) X(256, 2, 0, 1) X(256, 3, 0, 1) \
                            X(512, 0, 0, 1) X(512, 2, 0, 1) X(512, 3, 0, 1) X(512, 4, 0, 1) X(256, 4, 0, 1) \
                            X(64, 0, 4, 0) X(64, 2, 4, 0) X(64, 3, 4, 0) X(64, 0, 8, 0) X(64, 2, 8, 0) X(64, 3, 8, 0) \
                            X(64, 5, 4, 0) X(64, 5, 8, 0) X(64, 5, 4, 1) X(64, 5, 8, 1) \
                            X(64, 0, 0, 2) X(64, 2, 0, 2) X(64, 3, 0, 2) X(256, 0, 0, 2) X(256, 2, 0, 2) X(256, 3, 0, 2)
  // plain-cost variants exist for the register-cached column layouts only (the hot ones)
  const int flavor = general_assoc ? 2 : ((plain_costs && rpl > 0) ? 0 : 1);
  // Behind the fast path a plain-cost launch of the lean mode moves to the full LDS state when it fits in 112 KB: the lean mode buys
  // residency (8 problems per CU), which a handful of declined problems has no use for, while the row boxes in LDS open the sparse
  // column minima of phase 1 (lap_core.hpp::sparse_column_minima, Cost::kPlain) and LDS row fetches in every row pass — per-problem
  // latency is what the waiting sub-batch pays. MOT_LAP_BEHIND_FULL=0 keeps the lean mode (A/B measurements).
  static const bool behind_full = !(std::getenv("MOT_LAP_BEHIND_FULL") && std::getenv("MOT_LAP_BEHIND_FULL")[0] == '0');
  if (fast && behind_full && flavor == 0 && rpl > 0 && mode == 3 && b2 <= 112 * 1024) { mode = 2; lds = b2; }
  // ... and, plain or BoT-SORT costs alike, to the all-LDS state when that fits: the search's d / pred / cols / inv / tie next to it
  // (a sweep of the search is otherwise two or three dependent global round trips; MOT_LAP_BEHIND_ALL=0 switches it off)
  static const bool behind_all = !(std::getenv("MOT_LAP_BEHIND_ALL") && std::getenv("MOT_LAP_BEHIND_ALL")[0] == '0');
  const size_t b5 = b2 + 32 * nm + 64;
  if (fast && behind_full && behind_all && geom && flavor != 2 && rpl > 0 && (mode == 3 || mode == 2) && n * m >= 16384 &&
      b5 <= static_cast<size_t>(kLdsBudget) - 4096) { mode = 5; lds = b5; }
  static const bool behind_prio = !(std::getenv("MOT_LAP_BEHIND_PRIO") && std::getenv("MOT_LAP_BEHIND_PRIO")[0] == '0');
  std::lock_guard<std::mutex> attr_lock(attr_mu);
  if (!attr_set_dev[dev_slot]) {
#define MOT_ATTR(T, M, R, G)                                                                                               \
    { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_kernel<T, M, R, G>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget); \
      if (e != hipSuccess) return e; }
    MOT_LAP_VARIANTS(MOT_ATTR)
#undef MOT_ATTR
    attr_set_dev[dev_slot] = true;
  }
  // 8 wavefronts per problem when there are no more problems than CUs anyway (OC-SORT 4096 x 2048: the dense row sweeps of the
  // shortest-path search are 6144 columns wide). Measured on C4: 4 / 8 / 16 wavefronts = 53 / 88 / 66-79 frames/s — the uniform
  // part of a sweep is paid by every wavefront, and 16 of them leave 128 VGPRs each. MOT_LAP_WIDE8=0 switches it off.
  static const bool wide8_ok = !(std::getenv("MOT_LAP_WIDE8") && std::getenv("MOT_LAP_WIDE8")[0] == '0');
  const bool wide8 = wide && wide8_ok && (ntasks <= 256 || mode == 4) && flavor == 1;  // (mode 4: one problem per CU whatever the width)
  static const int wide_t = std::getenv("MOT_LAP_WIDE_T") ? std::atoi(std::getenv("MOT_LAP_WIDE_T")) : 0;  // (experiments)
  const int threads = (wide && flavor == 1 && (wide_t == 256 || wide_t == 512)) ? wide_t : (wide8 ? 512 : (wide ? 256 : 64));
  bool launched = false;
#define MOT_TRY(T, M, R, G)                                                                                \
  if (!launched && threads == T && mode == M && rpl == R && flavor == G) {                                 \
    hipLaunchKernelGGL((lap_kernel<T, M, R, G>), dim3(grid), dim3(T), lds, st, tasks, ntasks, fast ? (behind_prio ? 1 : 2) : 0, declined, fs_lds); \
    launched = true;                                                                                       \
  }
  MOT_LAP_VARIANTS(MOT_TRY)
#undef MOT_TRY
#undef MOT_LAP_VARIANTS
  if (!launched) return hipErrorInvalidValue;
  return hipGetLastError();
}
}  // namespace mot
