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

#include "lap_kernel_body.hpp"

namespace {
__device__ long long g_behind_scr[512][36];
__device__ long long g_behind[2][40];
#define MOT_LAP_VARIANTS_NARROW(X) X(64, 0, 0, 1) X(64, 2, 0, 1) X(64, 3, 0, 1) X(64, 0, 4, 1) X(64, 2, 4, 1) X(64, 3, 4, 1) \
                                   X(64, 0, 8, 1) X(64, 2, 8, 1) X(64, 3, 8, 1) \
                                   X(64, 0, 4, 0) X(64, 2, 4, 0) X(64, 3, 4, 0) X(64, 0, 8, 0) X(64, 2, 8, 0) X(64, 3, 8, 0) \
                                   X(64, 5, 4, 0) X(64, 5, 8, 0) X(64, 5, 4, 1) X(64, 5, 8, 1)
}  // namespace
MOT_LAP_TU_EXPORTS(lap_narrow, MOT_LAP_VARIANTS_NARROW)

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
  const size_t b6 = kScratch + fsb + 22 * nm + n + 64;                // duals, distances, 16-bit y / cols / inv, list lengths
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
  // Round 5: a MATRIX problem behind the fast path (DeepOC-SORT, StrongSORT, UCMCTrack, BoostTrack, HybridSORT: their costs are materialised, and the
  // duplicated tracks of deepocsort.cpp:456-503 make the optimum non-unique in almost every frame) ran on ONE wavefront with everything but the duals in
  // global scratch: 2.3-3.4 ms per declined 128 x 256 problem, 71 % of DeepOC-SORT's GPU time (tools/f3_latency_probe.py). It gets four wavefronts
  // (the dense row sweeps of the shortest-path search are nm wide) and the full hot state in LDS. MOT_LAP_BEHIND_MATRIX=0 keeps the old choice (A/B).
  static const bool behind_matrix_ok = !(std::getenv("MOT_LAP_BEHIND_MATRIX") && std::getenv("MOT_LAP_BEHIND_MATRIX")[0] == '0');
  const bool behind_matrix = behind_matrix_ok && fast && !geom && !general_assoc && nm >= 128 && nm <= 3072 && b2 <= static_cast<size_t>(kLdsBudget) - 1024;
  const bool wide = (nm > 3072 && (ntasks < 512 || fs_lds)) || behind || behind_matrix;
  // mode 4 (distances in LDS for the scan steps) exists for cost flavour 1 only
  if (fs_lds && wide && !general_assoc && b4 <= static_cast<size_t>(kLdsBudget) - 1024) { mode = 4; lds = b4; }
  static const bool lds16_ok = !(std::getenv("MOT_LAP_LDS16") && std::getenv("MOT_LAP_LDS16")[0] == '0');  // (A/B measurements)
  if (lds16_ok && fs_lds && wide && !general_assoc && nm <= 2 * static_cast<size_t>(mot::kFsEvl) && b6 <= static_cast<size_t>(kLdsBudget) - 1024) { mode = 6; lds = b6; }
  if (behind_matrix) { mode = 2; lds = b2; }
  static const int wide_mode = std::getenv("MOT_LAP_WIDE_MODE") ? std::atoi(std::getenv("MOT_LAP_WIDE_MODE")) : -1;  // (experiments: 3 = lean state for the wide matrix problems)
  if (wide_mode == 3 && fs_lds && wide && !general_assoc && !behind_matrix) { mode = 3; lds = b3; }
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
  // Round 5: with more than 256 columns (the north-star shape) such a launch runs FOUR wavefronts per declined problem, two register-cached columns per
  // lane instead of eight: a declined problem is ~530 serial rounds of the row reduction, each evaluating every owned column and reducing over the
  // group — a quarter of the evaluations per lane, and the whole state is in LDS, so the three extra wavefronts cost one LDS exchange per reduction.
  // (Round 4 tried four wavefronts with the state in global scratch and no register cache: no gain. MOT_LAP_BEHIND_QUAD=0 keeps one wavefront.)
  static const bool quad_ok = !(std::getenv("MOT_LAP_BEHIND_QUAD") && std::getenv("MOT_LAP_BEHIND_QUAD")[0] == '0');  // (A/B measurements)
  const bool behind_quad = mode == 5 && rpl == 8 && quad_ok && !wide;
  if (behind_quad) rpl = 2;
  static const bool behind_prio = !(std::getenv("MOT_LAP_BEHIND_PRIO") && std::getenv("MOT_LAP_BEHIND_PRIO")[0] == '0');
  {
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (!attr_set_dev[dev_slot]) {
      hipError_t e = lap_narrow_attr();
      if (e == hipSuccess) e = lap_wide_attr();
      if (e == hipSuccess) e = lap_general_attr();
      if (e != hipSuccess) return e;
      attr_set_dev[dev_slot] = true;
    }
  }
  // 8 wavefronts per problem when there are no more problems than CUs anyway (OC-SORT 4096 x 2048: the dense row sweeps of the
  // shortest-path search are 6144 columns wide). Measured on C4: 4 / 8 / 16 wavefronts = 53 / 88 / 66-79 frames/s — the uniform
  // part of a sweep is paid by every wavefront, and 16 of them leave 128 VGPRs each. MOT_LAP_WIDE8=0 switches it off.
  static const bool wide8_ok = !(std::getenv("MOT_LAP_WIDE8") && std::getenv("MOT_LAP_WIDE8")[0] == '0');
  const bool wide8 = wide && wide8_ok && (ntasks <= 256 || mode == 4 || mode == 6) && flavor == 1;  // (mode 4: one problem per CU whatever the width)
  static const int wide_t = std::getenv("MOT_LAP_WIDE_T") ? std::atoi(std::getenv("MOT_LAP_WIDE_T")) : 0;  // (experiments)
  const int threads = (behind_quad || behind_matrix) ? 256 : ((wide && flavor == 1 && (wide_t == 256 || wide_t == 512)) ? wide_t : (wide8 ? 512 : (wide ? 256 : 64)));
  static LapDiag diag_dev[64] = {};
  {
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (!diag_dev[dev_slot].scr) {
      void *a = nullptr, *b = nullptr;
      hipError_t e = hipGetSymbolAddress(&a, HIP_SYMBOL(g_behind_scr));
      if (e == hipSuccess) e = hipGetSymbolAddress(&b, HIP_SYMBOL(g_behind));
      if (e != hipSuccess) return e;
      diag_dev[dev_slot].scr = reinterpret_cast<long long (*)[36]>(a);
      diag_dev[dev_slot].sum = reinterpret_cast<long long (*)[40]>(b);
    }
  }
  const LapLaunchArgs A{threads, mode, rpl, flavor, grid, lds, st, tasks, ntasks, fast ? (behind_prio ? 1 : 2) : 0, declined, fs_lds, diag_dev[dev_slot]};
  const bool launched = (flavor == 2) ? lap_general_launch(A) : (threads > 64 ? lap_wide_launch(A) : lap_narrow_launch(A));
  if (!launched) return hipErrorInvalidValue;
  return hipGetLastError();
}
}  // namespace mot
