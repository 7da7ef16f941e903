// Sparse exact assignment with a uniqueness certificate — the fast path in front of lap_core.hpp's exact lapjv emulation.
//
// What utils::linear_assignment (src/utils/matching.cpp:14-60 -> include/motcpp/association/lap_solver.hpp:251-332) solves:
// lapjv on the (nr+nc)^2 extension of the nr x nc cost (off-diagonal blocks thresh/2, corner 0). The cost of an extended
// assignment with real pairs M is  const + sum_{(i,j) in M} (c_ij - thresh): the real block of lapjv's answer is a
// minimum-weight matching of the bipartite graph whose edges are the pairs with c_ij < thresh ("viable" pairs), weight
// w_ij = c_ij - thresh < 0; every other choice lapjv makes (which dummy goes where) never leaves the solver. lapjv is an
// exact algorithm, so whenever that minimum-weight matching is UNIQUE, it is what lapjv returns, whatever its scan orders.
// Tracking costs are sparse (a detection is viable for a handful of tracks), so the matching is found over the viable
// pairs only:
//   1. enumerate the viable pairs, per column (geometry: rows bucketed by x1, only rows that can intersect the column are
//      evaluated, with the cost kernel's own arithmetic — a pair that does not intersect costs the same whatever the row
//      is; materialised matrix: one coalesced sweep), compacted into a CSR list;
//   2. column duals v_j = min_i w_ij, every column proposes to its best row, a row keeps its lowest proposer;
//   3. each remaining column is inserted by a shortest augmenting path search (Dijkstra over the viable pairs with the
//      reduced costs w_ij - u_i - v_j; a column may also end unmatched, at reduced cost -v_j), duals updated as in lapjv;
//   4. certificate: complementary slackness is re-checked on every viable pair, and an alternative matching within
//      eps of the optimum exists iff the graph of eps-tight non-matching pairs has an alternating cycle or an alternating
//      path between two vertices that can change their matched state for free — searched explicitly.
// Anything that is not provably the unique optimum by a margin — an eps-tie, a pair within eps of the threshold, a NaN or
// infinite cost, a non-intersecting pair that would itself be viable, an overflow of one of the fixed-size tables — is
// handed to the exact lapjv emulation (lap_core.hpp), which reproduces the reference's tie-breaks step by step. The fast
// path therefore never decides a tie; it only recognises problems that have none.
//
// eps = 1e-9 on costs of magnitude O(1): far above the rounding of either solver's fp64 duals (~n * 2^-52), far below the
// spacing of float costs (IoU-family costs are multiples of 2^-24), so in practice it fires on exact ties only.
//
// Memory: everything the serial part touches (duals, assignments, the CSR list, search slots, and — while the pairs are
// enumerated — the bucket-ordered row boxes, which share their bytes with the row duals/assignments that only exist
// afterwards) is "hot" and sits in LDS when the problem fits; the per-column staging lists of the enumeration, the free list
// and the certificate's arc list are "cold" (global scratch, streamed).
#pragma once
#include <type_traits>
#include "cost_math.hpp"
#include "grp.hpp"
#include "lap_core.hpp"
#include "mem.hpp"

#if !defined(__HIPCC__) && defined(MOT_SPARSE_DEBUG)
#include <cstdio>
#define SPDBG(...) std::fprintf(stderr, __VA_ARGS__)
#else
#define SPDBG(...) ((void)0)
#endif

namespace mot {

constexpr int kSpK = 32;           // viable pairs per column the staging lists of a matrix source hold (more: fall back)
constexpr int kSpSeg0 = 16;        // box source: list entries reserved for a column whose candidates are still coming in; a column that
                                   // outgrows its segment moves to one of twice the size ...
constexpr int kSpKMax = 64;        // ... up to this many viable pairs (more: fall back). A path search relaxes a column with one lane per pair.
constexpr int kSpBuckets = 256;    // x1 buckets of the row boxes
constexpr int kSpSlots = 63;       // rows one path search may reach (more: fall back); lane q holds slot q, lane 63 none
constexpr int kSpQ = 12;           // per-lane queue of candidate rows awaiting the exact arithmetic (drained when the wavefront has filled its queues or finished its windows)
constexpr double kSpEps = 1e-9;    // tie margin
constexpr double kSpTol = 1e-11;   // tolerated violation of dual feasibility / complementary slackness (fp64 rounding)
constexpr int kSpIntMax = 0x7fffffff;
constexpr float kSpHuge = 1.0e30f; // costs / confidences beyond this magnitude are left to the exact path
constexpr float kSpBoxHuge = 1.0e15f;  // ... and box coordinates beyond this one: below it no width, area or product of two extents overflows, so the
                                       // division-free pre-test of the candidate scan (products of coordinate differences) holds no inf or NaN
MOT_HD int sparse_arc_cap(int nr, int nc) { return nr + nc + 64; }  // eps-tight non-matching pairs the certificate may hold

struct alignas(16) SpBox { float x1, y1, x2, y2; };
// min / max of FINITE floats for the candidate pre-test (one v_min_f32 / v_max_f32 each; smin / smax keep std::min / std::max's NaN and
// signed-zero behaviour with a compare and a select, which only the exact cost arithmetic needs)
#if defined(__HIP_DEVICE_COMPILE__)
MOT_DEV float sp_fmin(float a, float b) { return __builtin_fminf(a, b); }
MOT_DEV float sp_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
#else
MOT_DEV float sp_fmin(float a, float b) { return (b < a) ? b : a; }
MOT_DEV float sp_fmax(float a, float b) { return (a < b) ? b : a; }
#endif

// Workspace. HS = address space of the hot arrays (LDS when the problem fits, else global scratch).
template <int HS>
struct SparseWorkT {
  // state of the matching: u | x | slot (16 bytes per row), v | y (12 bytes per column) — or, while the pairs are
  // enumerated, the bucket-ordered row boxes (over u, x, slot) and the x1 buckets + candidate queues (over v, y)
  MemPtr<double, HS> u;          // row duals
  MemPtr<int, HS> x;             // row -> column (-1: unmatched)
  MemPtr<int, HS> slot;          // per row: 1 + search slot during a path search; flag bits during the certificate
  MemPtr<double, HS> v;          // column duals
  MemPtr<int, HS> y;             // column -> row
  MemPtr<SpBox, HS> sbox;        // [nr] (aliases u, x, slot)
  MemPtr<int, HS> bstart, bcur, bmax;  // x1 buckets: [B+4] start, [B] fill cursor, [B] prefix maximum of the x2 keys (alias v, y)
  MemPtr<unsigned short, HS> hq; // [kSpQ][threads] per-lane queue of candidate positions worth the exact arithmetic (aliases v, y)
  MemPtr<unsigned short, HS> sidx;  // [nr] row index of a bucket-ordered position
  MemPtr<int, HS> eoff;          // [nc] CSR entry of a column: start | count << 24
  MemPtr<unsigned short, HS> erow;  // [ecap] row ...
  MemPtr<float, HS> ecost;       // [ecap] ... and cost of a viable pair
  MemPtr<int, HS> ctr;           // [4] counters
  int ecap = 0;
  MemPtr<int, kMemGlobal> strow;    // [nc][kSpK] staging of a column's viable pairs (matrix source): row ...
  MemPtr<float, kMemGlobal> stcost; // ... cost
  MemPtr<int, kMemGlobal> freel;    // [nc] columns still to insert
  MemPtr<int, kMemGlobal> arcs;     // [sparse_arc_cap][2] eps-tight pair: (row, owner of its column)
};
constexpr int kSpMaxThreads = 256;  // lanes that may enumerate one problem together
constexpr size_t kSpEnumBytes = 4 * (kSpBuckets + 4) + 8 * kSpBuckets + 2 * kSpQ * kSpMaxThreads;  // buckets + queues
MOT_HD size_t sparse_vy_bytes(int nc) { const size_t b = 12 * static_cast<size_t>(nc); return ((b > kSpEnumBytes ? b : kSpEnumBytes) + 15) & ~size_t(15); }
MOT_HD size_t sparse_hot_bytes(int nr, int nc, int ecap) {
  return static_cast<size_t>(nr) * 16 + sparse_vy_bytes(nc) + ((static_cast<size_t>(nr) * 2 + 15) & ~size_t(15)) + 4 * (static_cast<size_t>(nc) + 4) +
         ((static_cast<size_t>(ecap) * 2 + 15) & ~size_t(15)) + static_cast<size_t>(ecap) * 4 + 16 + 64;
}
MOT_HD int sparse_default_ecap(int nc) { return 3 * nc + 64; }   // in LDS (round 3: 4 * nc + 64 — three more kilobytes per north-star problem)
MOT_HD int sparse_global_ecap(int nc) { return 8 * nc + 64; }    // hot state in global scratch: room is not the issue
MOT_HD size_t sparse_cold_bytes(int nr, int nc) {
  return static_cast<size_t>(nc) * (8 * kSpK + 4) + static_cast<size_t>(sparse_arc_cap(nr, nc)) * 8 + 64;
}
// Scratch of one task (mot_lap_work_bytes): the exact solver's hot + cold arrays and staged boxes, then the fast path's
// cold lists and (when it does not fit in LDS) hot state with the default list capacity; the last 16 bytes hold the task's
// status word (1: finished by the fast path).
MOT_HD size_t lap_task_scratch_bytes(int n, int m) {
  const size_t rn = n > 0 ? n : 0, rm = m > 0 ? m : 0, nm = rn + rm;
  const int in = static_cast<int>(rn), im = static_cast<int>(rm);
  return ((lap_hot_bytes(static_cast<int>(nm)) + 15) & ~size_t(15)) + ((lap_cold_bytes(static_cast<int>(nm)) + 15) & ~size_t(15)) + 4 * (5 * rn + 6 * rm) + 256 +
         ((sparse_cold_bytes(in, im) + sparse_hot_bytes(in, im, sparse_global_ecap(im)) + 63) & ~size_t(15));
}
template <class W>
MOT_HD void sparse_carve_hot(W& w, void* base, int nr, int nc, int ecap) {
  char* p = static_cast<char*>(base);
  w.u.p = reinterpret_cast<double*>(p);
  w.sbox.p = reinterpret_cast<SpBox*>(p);
  w.x.p = reinterpret_cast<int*>(p + 8 * static_cast<size_t>(nr));
  w.slot.p = reinterpret_cast<int*>(p + 12 * static_cast<size_t>(nr));
  p += 16 * static_cast<size_t>(nr);
  w.v.p = reinterpret_cast<double*>(p);
  w.y.p = reinterpret_cast<int*>(p + 8 * static_cast<size_t>(nc));
  w.bstart.p = reinterpret_cast<int*>(p);
  w.bcur.p = w.bstart.p + kSpBuckets + 4;
  w.bmax.p = w.bcur.p + kSpBuckets;
  w.hq.p = reinterpret_cast<unsigned short*>(w.bmax.p + kSpBuckets);
  p += sparse_vy_bytes(nc);
  w.sidx.p = reinterpret_cast<unsigned short*>(p); p += (2 * static_cast<size_t>(nr) + 15) & ~size_t(15);
  w.erow.p = reinterpret_cast<unsigned short*>(p); p += (2 * static_cast<size_t>(ecap) + 15) & ~size_t(15);
  w.eoff.p = reinterpret_cast<int*>(p); p += 4 * (static_cast<size_t>(nc) + 4);
  w.ecost.p = reinterpret_cast<float*>(p); p += 4 * static_cast<size_t>(ecap);
  w.ctr.p = reinterpret_cast<int*>(p);
  w.ecap = ecap;
}
template <class W>
MOT_HD void sparse_carve_cold(W& w, void* base, int nr, int nc) {
  (void)nr;
  char* p = static_cast<char*>(base);
  w.strow.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(kSpK) * nc;
  w.stcost.p = reinterpret_cast<float*>(p); p += 4 * static_cast<size_t>(kSpK) * nc;
  w.freel.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(nc);
  w.arcs.p = reinterpret_cast<int*>(p);
}

// Outcome of the enumeration of viable pairs.
struct SparseEnum {
  SparseEnum() = default;
  MOT_DEV SparseEnum(int ok_, double mn) : ok(ok_), mincost(mn) {}
  long long c_stage = 0, c_cand = 0, c_csr = 0;  // diagnostics: shader cycles of bucketing the rows / the candidate sweep / the CSR build
  long long n_cand = 0, n_hit = 0;               // this lane's candidate rows looked at / intersecting pairs evaluated
  int ok = 0;      // 1 fine; <= 0: something the fast path does not decide was seen — 0 (matrix source: a tie with the threshold /
                   // NaN / inf), -10 more than kSpK viable pairs in a column, -11 the CSR list does not fit, -12 NaN / inf / out of
                   // range, -13 pairs that do not intersect would be viable, -14 a cost within eps of the threshold
  double mincost = 0.0;  // minimum cost over ALL pairs (MOT_LAP_GATE_MIN)
};

MOT_DEV int sp_e0(int e) { return e & 0xffffff; }          // packed CSR entry of a column: start | count << 24
MOT_DEV int sp_deg(int e) { return (e >> 24) & 0xff; }
// per-column staging lists (counts in y[j], which has no other use yet) -> CSR in the hot space; false when the list does not fit
template <class G, class W>
MOT_DEV bool sparse_build_csr(G& g, const W& w, int nc) {
  const int T = g.size(), t = g.tid();
  const int L = (nc + T - 1) / T;
  const int j0 = t * L, j1 = (j0 + L < nc) ? j0 + L : nc;
  int s = 0;
  for (int j = j0; j < j1; ++j) s += static_cast<int>(w.y[j]);
  int total;
  int base = g.exclusive_scan(s, &total);
  for (int j = j0; j < j1; ++j) {
    const int c = w.y[j];
    w.eoff[j] = base | (c << 24);
    base += c;
  }
  g.sync();
  if (total > w.ecap) return false;
  for (int j = t; j < nc; j += T) {
    const int e = w.eoff[j], b = sp_e0(e), c = sp_deg(e);
    for (int k = 0; k < c; ++k) {
      w.erow[b + k] = static_cast<unsigned short>(static_cast<int>(w.strow[static_cast<size_t>(j) * kSpK + k]));
      w.ecost[b + k] = w.stcost[static_cast<size_t>(j) * kSpK + k];
    }
  }
  g.sync();
  return true;
}

// ---- 1a. viable pairs from boxes ---------------------------------------------------------------------------------
// Box source of one problem: planes [4][ld] with an optional gather index, as in mot_iou_task.
struct SparseBoxes {
  const float* p; int ld; const int* idx;
  MOT_DEV int gather(int i) const { return idx ? gld(idx, i) : i; }
  MOT_DEV void load(int i, float b[4]) const {
    const int gi = gather(i);
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = gld(p, static_cast<size_t>(k) * ld + gi);
  }
  MOT_DEV float load_x1(int i) const { return gld(p, static_cast<size_t>(gather(i))); }
};
MOT_DEV bool sp_finite4(const float b[4]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) ok = ok && (b[k] > -kSpBoxHuge) && (b[k] < kSpBoxHuge);  // false for NaN
  return ok;
}
// EvalFn(row index, row box, row area, column box, column area, column confidence, column index) -> float cost with the
// cost kernel's arithmetic; zc(conf) = cost of a pair that does not intersect.
// A lane sweeps the candidate rows of one of its columns with box tests only, queueing the positions that intersect (LDS,
// kSpQ per lane), then evaluates the queued pairs back to back — the expensive arithmetic runs with every lane busy instead
// of under a one-in-seven branch — and writes the viable ones straight into a CSR segment it reserved for the queue's length.
constexpr int kSpRC = 20;  // rows per lane whose x1 / gather index stay in registers across the three bucketing passes (default of the
                           // RC template parameter below; the four-wavefront kernel passes 6: its rows are shared by 256 lanes, and the 100
                           // registers of the default were what limited it to four wavefronts per SIMD)
struct SpNoMinIou { MOT_DEV float operator()(float) const { return 0.0f; } };
template <int RC = kSpRC, int QD = kSpQ, class G, class W, class EvalFn, class ZeroFn, class MinIouFn = SpNoMinIou>
MOT_DEV SparseEnum sparse_enumerate_boxes(G& g, const W& w, int nr, int nc, const SparseBoxes& A, const SparseBoxes& Bx,
                                          const float* bconf, const int* bidx, float thresh, EvalFn eval, ZeroFn zero_cost,
                                          MinIouFn min_iou_of = MinIouFn()) {
  // min_iou_of(conf) > 0: the caller guarantees that a pair of that column whose IoU is not above the value (a 1 % margin
  // already taken off) costs more than thresh + eps — such pairs are dropped by a division-free test
  // (inter > min_iou * union) before the exact arithmetic; the minimum cost over all pairs is then not reported
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  const long long ec0 = MOT_CLOCK();
  long long n_cand = 0, n_hit = 0, c_drain = 0;
  // bits: 1 tie with the threshold, 2 column list full, 4 NaN / inf / out of range, 8 viable pairs that do not intersect, 16 CSR full
  // QD = depth of a lane's queue: the queues of all lanes share the 2 * kSpQ * kSpMaxThreads bytes behind the buckets (16 wavefronts: 2 each)
  static_assert(QD >= 1 && QD <= kSpQ, "queue depth");
  int bad = (nr > 65535 || T * QD > kSpMaxThreads * kSpQ) ? 4 : 0;
  // ---- rows into x1 buckets ----
  const bool cached = nr <= T * RC;
  float rx1[RC];
  int rgi[RC];
  float xlo = 3.0e38f, xhi = -3.0e38f;
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u) { const int i = t + u * T; rgi[u] = (i < nr) ? A.gather(i) : 0; }
#pragma unroll
    for (int u = 0; u < RC; ++u) { const int i = t + u * T; rx1[u] = (i < nr) ? gld(A.p, static_cast<size_t>(rgi[u])) : 0.0f; }
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const int i = t + u * T;
      if (i < nr) {
        const float a0 = rx1[u];
        if (!(a0 > -kSpBoxHuge && a0 < kSpBoxHuge)) bad |= 4;
        if (a0 < xlo) xlo = a0;
        if (a0 > xhi) xhi = a0;
      }
    }
  } else {
#pragma unroll 4
    for (int i = t; i < nr; i += T) {
      const float a0 = A.load_x1(i);
      if (!(a0 > -kSpBoxHuge && a0 < kSpBoxHuge)) bad |= 4;
      if (a0 < xlo) xlo = a0;
      if (a0 > xhi) xhi = a0;
    }
  }
  for (int b = t; b <= kSpBuckets; b += T) w.bstart[b] = 0;
  for (int b = t; b < kSpBuckets; b += T) w.bmax[b] = static_cast<int>(0x80000000u);
  if (t == 0) w.ctr[2] = 0;
  xlo = static_cast<float>(g.reduce_min(static_cast<double>(xlo)));
  xhi = -static_cast<float>(g.reduce_min(-static_cast<double>(xhi)));
  bad = g.reduce_max(bad);
  if (bad) return SparseEnum(-12, 0.0);
  const float span = xhi - xlo;
  const float scale = (span > 0.0f) ? static_cast<float>(kSpBuckets) / span : 0.0f;
  // monotone non-decreasing in x (so a0 < b2 implies bucket(a0) <= bucket(b2)); any x maps into [0, B)
  auto bucket = [&](float x) {
    const float f = (x - xlo) * scale;
    int b = (f > 0.0f) ? ((f < static_cast<float>(kSpBuckets)) ? static_cast<int>(f) : kSpBuckets - 1) : 0;
    return (b < kSpBuckets) ? b : kSpBuckets - 1;
  };
  g.sync();
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u)
      if (t + u * T < nr) G::atomic_add(w.bstart.raw(bucket(rx1[u]) + 1), 1);
  } else {
#pragma unroll 4
    for (int i = t; i < nr; i += T) G::atomic_add(w.bstart.raw(bucket(A.load_x1(i)) + 1), 1);
  }
  g.sync();
  {  // exclusive scan of the bucket counts: contiguous chunk per lane
    const int L = (kSpBuckets + T - 1) / T;
    const int b0 = t * L, b1 = (b0 + L < kSpBuckets) ? b0 + L : kSpBuckets;
    int s = 0;
    for (int b = b0; b < b1; ++b) s += static_cast<int>(w.bstart[b + 1]);
    int total;
    int base = g.exclusive_scan(s, &total);
    for (int b = b0; b < b1; ++b) {
      const int c = w.bstart[b + 1];
      w.bcur[b] = base;  // fill cursor = start of bucket b
      base += c;
    }
  }
  g.sync();
  auto place = [&](int i, const float a[4]) {
    if (!sp_finite4(a)) bad |= 4;
    const int b = bucket(a[0]);
    const int pos = G::atomic_add(w.bcur.raw(b), 1);
    w.sbox[pos] = SpBox{a[0], a[1], a[2], a[3]};
    w.sidx[pos] = static_cast<unsigned short>(i);
    G::atomic_max(w.bmax.raw(b), f32_key(a[2]));
  };
  if (cached) {
    float r1[RC], r2[RC], r3[RC];
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const bool in = t + u * T < nr;
      r1[u] = in ? gld(A.p, static_cast<size_t>(A.ld) + rgi[u]) : 0.0f;
      r2[u] = in ? gld(A.p, static_cast<size_t>(2) * A.ld + rgi[u]) : 0.0f;
      r3[u] = in ? gld(A.p, static_cast<size_t>(3) * A.ld + rgi[u]) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < RC; ++u)
      if (t + u * T < nr) { const float a[4] = {rx1[u], r1[u], r2[u], r3[u]}; place(t + u * T, a); }
  } else {
#pragma unroll 2
    for (int i = t; i < nr; i += T) { float a[4]; A.load(i, a); place(i, a); }
  }
  g.sync();
  // after the fill bcur[b] = end of bucket b; bstart[b] = start: bstart[0] = 0, bstart[b+1] = bcur[b]
  for (int b = t; b < kSpBuckets; b += T) w.bstart[b + 1] = w.bcur[b];
  if (t == 0) w.bstart[0] = 0;
  {  // prefix maximum of the x2 keys over the buckets
    const int L = (kSpBuckets + T - 1) / T;
    const int b0 = t * L, b1 = (b0 + L < kSpBuckets) ? b0 + L : kSpBuckets;
    int run = static_cast<int>(0x80000000u);
    for (int b = b0; b < b1; ++b) {
      const int k2 = w.bmax[b];
      if (k2 > run) run = k2;
      w.bmax[b] = run;
    }
    const double ex = g.exclusive_scan_min(-static_cast<double>(run));
    if (ex < 1e299) {
      const int prev = static_cast<int>(-ex);
      for (int b = b0; b < b1; ++b)
        if (static_cast<int>(w.bmax[b]) < prev) w.bmax[b] = prev;
    }
  }
  g.sync();
  // ---- columns: candidates = rows of the buckets [blo, bhi] ----
  const long long ec1 = MOT_CLOCK();
  double mn = 1e300;
  // the next column's box and confidence are fetched while the current one is processed
  float nb[4] = {0.f, 0.f, 0.f, 0.f}, nconf = 0.0f;
  auto fetch_column = [&](int j) {
    Bx.load(j, nb);
    const int gj = bidx ? gld(bidx, j) : j;
    nconf = bconf ? gld(bconf, gj) : 0.0f;
  };
  if (t < nc) fetch_column(t);
  // every lane runs every round of the column loop (a lane without a column has an empty window): the rounds hold a wavefront-wide vote
  for (int j0 = 0; j0 < nc; j0 += T) {
    const int j = j0 + t;
    const bool have = j < nc;
    const float b[4] = {nb[0], nb[1], nb[2], nb[3]};
    const float conf = nconf;
    if (j + T < nc) fetch_column(j + T);
    if (have && (!sp_finite4(b) || !(conf > -kSpHuge && conf < kSpHuge))) bad |= 4;
    const float barea = (b[2] - b[0]) * (b[3] - b[1]);
    const float zc = zero_cost(conf);
    const float min_iou = min_iou_of(conf);
    if (have && !(static_cast<double>(zc) > th + kSpEps)) bad |= 8;  // a non-intersecting pair would be viable (or zc is NaN)
    int nq = 0, ne = 0, base = 0, seg = -1, ninter = 0;
    {
      // Round 5: a candidate must pass hits(), i.e. inter > min_iou * union. With union >= the column's area and ih <= its height that forces
      // iw > min_iou * (b2 - b0), and iw <= b2 - a0 and iw <= a2 - b0: a row that can pass starts before b2 - m and ends after b0 + m,
      // m = 0.9 * min_iou * width (the tenth dwarfs every rounding: the terms are float expressions of the same magnitude). The x window of the
      // column shrinks by that margin at both ends (a fifth fewer candidates at the north-star shape); a width that is not a positive finite
      // number leaves it as it was.
      const float bw = b[2] - b[0];
      const float mrg = (min_iou > 0.0f && bw > 0.0f && bw < kSpHuge) ? 0.9f * min_iou * bw : 0.0f;
      const int bhi = bucket(b[2] - mrg);
      const int kb0 = f32_key(b[0] + mrg);
      int lo = 0, hi = bhi + 1;  // first bucket whose prefix maximum of x2 exceeds b0
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (static_cast<int>(w.bmax[mid]) > kb0) hi = mid; else lo = mid + 1; }
      const bool scan = have && !bad;
      const int ps = scan ? static_cast<int>(w.bstart[lo]) : 0, pe = scan ? static_cast<int>(w.bstart[bhi + 1]) : 0;
      n_cand += pe - ps;
      // evaluates the queued pairs, marks the viable ones, reserves exactly that many CSR entries (the most a column may
      // hold if more candidates are still to come) and evaluates the viable ones once more to store them
      auto pair_cost = [&](int q, int* row) {
        const int pp = w.hq[q * T + t];
        const SpBox s = w.sbox[pp];
        *row = w.sidx[pp];
        const float a[4] = {s.x1, s.y1, s.x2, s.y2};
        const float aarea = (a[2] - a[0]) * (a[3] - a[1]);
        return eval(*row, a, aarea, b, barea, conf, j);
      };
      auto drain = [&](bool last) {
        if (nq == 0) return;
        unsigned viable = 0u;
        float cq[QD];  // the queued pairs' costs and rows stay in registers between the viability pass and the store (round 6: the viable
        int rq[QD];    // ones used to be evaluated a second time — a correctly rounded division each)
#pragma unroll
        for (int q = 0; q < QD; ++q) {
          cq[q] = 0.0f; rq[q] = 0;
          if (q < nq) {
            const float c = pair_cost(q, &rq[q]);
            cq[q] = c;
            if (!(c > -kSpHuge && c < kSpHuge)) bad |= 4;
            else {
              const double cd = static_cast<double>(c);
              if (cd < mn) mn = cd;
              // a cost EQUAL to the threshold stays in the graph as a pair of weight 0: it matters only if it is tight in the
              // optimum, which the certificate checks like any other tie (float costs are not otherwise within eps of it)
              if (cd < th - kSpEps || cd == th) viable |= 1u << q;
              else if (!(cd > th + kSpEps)) bad |= 1;
            }
          }
        }
        const int nv = __builtin_popcount(viable);
        if (seg < 0 && nv > 0) {
          // (shallow queues drain before a column's scan is over nearly every time: they start with a small segment — it doubles when outgrown)
          constexpr int seg0 = (QD < kSpQ) ? 4 : kSpSeg0;
          seg = last ? nv : ((nv > seg0) ? nv : seg0);
          if (seg > kSpKMax) { bad |= 2; seg = 0; }
          base = (seg > 0) ? G::atomic_add(w.ctr.raw(2), seg) : 0;
          if (base + seg > w.ecap) { bad |= 16; seg = 0; }
        }
#pragma unroll
        for (int q = 0; q < QD; ++q) {
          if (!(viable & (1u << q))) continue;
          if (ne == seg && seg > 0 && !bad) {  // the column outgrew its segment (a pile-up of lost tracks on one detection): move it
            const int ns = (2 * seg < kSpKMax) ? 2 * seg : kSpKMax;
            if (ns == seg) bad |= 2;
            else {
              const int nb = G::atomic_add(w.ctr.raw(2), ns);
              if (nb + ns > w.ecap) bad |= 16;
              else {
                for (int k = 0; k < ne; ++k) { w.erow[nb + k] = static_cast<unsigned short>(static_cast<unsigned short>(w.erow[base + k])); w.ecost[nb + k] = static_cast<float>(w.ecost[base + k]); }
                base = nb; seg = ns;
              }
            }
          }
          if (ne < seg) { w.erow[base + ne] = static_cast<unsigned short>(rq[q]); w.ecost[base + ne] = cq[q]; ++ne; }
          else if (!(bad & 16)) bad |= 2;
        }
        ninter += nq;
        nq = 0;
      };
      // A candidate is worth the exact arithmetic iff iou_pair's intersection is positive — w = xx2 - xx1 > 0 and h = yy2 - yy1 > 0, the same
      // float expressions as there; anything else costs zc, which is not viable — and, when the column has a minimum IoU, iff
      // inter > min_iou * union (iou_pair's inter and union without the division). Round 6: branch-free and on native min / max — with
      // inter' = max(w, 0) * h the one test  inter' > min_iou * union  covers both: w <= 0 gives inter' = 0 and h <= 0 gives inter' <= 0, against a
      // positive right-hand side (min_iou = 0: against 0); a degenerate box with union <= 0 can only let a pair through that the exact
      // arithmetic then prices at zc. 14 VALU instructions per candidate instead of 26 (four compares in front of a branch that some lane of
      // the wavefront nearly always took, compare + select pairs for every min / max); box coordinates are below kSpBoxHuge, so no product
      // overflows (and a NaN cannot occur). The kernel's four wavefronts per SIMD kept the vector ALU busy two thirds of the time: this loop was a third of its instructions.
      const float bh = b[3] - b[1];
      auto passes = [&](const SpBox& s) {
        // w = min(x2, b2) - max(x1, b0) is the smallest of the four differences x2 - x1, x2 - b0, b2 - x1, b2 - b0 — rounding is monotone, so the
        // smallest rounded difference IS fl(min - max), bit for bit — and the row's own extents are needed for its area anyway. Differences are
        // arithmetic results: the minima below need no quieting of signalling NaNs (min / max of loaded values cost three instructions each).
        const float wr = s.x2 - s.x1, hr = s.y2 - s.y1;
        const float dx1 = s.x2 - b[0], dy1 = s.y2 - b[1], dx2 = b[2] - s.x1, dy2 = b[3] - s.y1;
        const float iw = sp_fmin(sp_fmin(sp_fmin(wr, dx1), dx2), bw), ih = sp_fmin(sp_fmin(sp_fmin(hr, dy1), dy2), bh);
        const float inter = sp_fmax(iw, 0.0f) * ih;
        const float uni = wr * hr + barea - inter;
        return inter > min_iou * uni;
      };
      // Round 6: EPOCHS. A lane used to drain its queue the moment it was full — a wavefront then ran the eight unrolled evaluations (a
      // correctly rounded division, double-precision compares) for that ONE lane, three or four times per round of columns (a column in
      // twenty has more than eight candidates that pass; with 64 columns per wavefront some lane always does), before the common drain at
      // the end: about 64 evaluation slots issued per wavefront and problem for 7 slots' worth of pairs. Now a lane whose queue is full
      // stops scanning and waits; when every lane of the wavefront has either finished its window or filled its queue they all drain
      // together, and the few with candidates left go round again. With a queue of twelve the second epoch is rare.
      int p = ps, pc = ps;
      unsigned m = 0u;
      bool more = pe > ps;
      for (;;) {
        if (more) {
          for (;;) {
            while (m != 0u && nq < QD) { const int q = __builtin_ctz(m); m &= m - 1u; w.hq[nq * T + t] = static_cast<unsigned short>(pc + q); ++nq; }
            if (m != 0u) break;  // the queue is full: the rest of this chunk waits for the next epoch
            if (p >= pe) { more = false; break; }
            // eight boxes per round trip (reads past the window stay inside the workgroup's LDS and are masked)
            const SpBox s0 = w.sbox[p], s1 = w.sbox[p + 1], s2 = w.sbox[p + 2], s3 = w.sbox[p + 3];
            const SpBox s4 = w.sbox[p + 4], s5 = w.sbox[p + 5], s6 = w.sbox[p + 6], s7 = w.sbox[p + 7];
            m = (passes(s0) ? 1u : 0u) | (passes(s1) ? 2u : 0u) | (passes(s2) ? 4u : 0u) | (passes(s3) ? 8u : 0u) |
                (passes(s4) ? 16u : 0u) | (passes(s5) ? 32u : 0u) | (passes(s6) ? 64u : 0u) | (passes(s7) ? 128u : 0u);
            const int left = pe - p;
            if (left < 8) m &= (1u << left) - 1u;
            pc = p;
            p += 8;
          }
        }
        const long long dc0 = MOT_CLOCK();
        drain(!more);
        c_drain += MOT_CLOCK() - dc0;
        if (!g.wave_any(more)) break;
      }
      n_hit += ninter;
    }
    nq = ninter;
    if (have && nq < nr && static_cast<double>(zc) < mn) mn = static_cast<double>(zc);
    if (have) w.eoff[j] = base | (ne << 24);
  }
  {  // OR of the lanes' flags
    int any = 0;
    for (int bit = 1; bit <= 16; bit <<= 1) any |= g.reduce_max((bad & bit) ? bit : 0);
    bad = any;
  }
  mn = g.reduce_min(mn);
  g.sync();
  SparseEnum out(1, mn);
  if (bad) out.ok = (bad & 4) ? -12 : ((bad & 8) ? -13 : ((bad & 1) ? -14 : ((bad & 2) ? -10 : -11)));
  out.c_stage = ec1 - ec0; out.c_cand = MOT_CLOCK() - ec1; out.c_csr = c_drain; out.n_cand = n_cand; out.n_hit = n_hit;  // (c_csr: the part of c_cand spent evaluating and storing the queued pairs)
  return out;
}

// ---- 1b. viable pairs from a materialised matrix (row-major, lane-owned columns: coalesced) -------------------------
template <class G, class W>
MOT_DEV SparseEnum sparse_enumerate_matrix(G& g, const W& w, int nr, int nc, const float* cost, int ld, float thresh) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  int bad = (nr > 65535) ? 1 : 0;
  double mn = 1e300;
  for (int j = t; j < nc; j += T) {
    int ne = 0;
#pragma unroll 8
    for (int i = 0; i < nr; ++i) {
      const float c = gld(cost, static_cast<size_t>(i) * ld + j);
      if (!(c > -kSpHuge && c < kSpHuge)) bad = 1;
      const double cd = static_cast<double>(c);
      if (cd < mn) mn = cd;
      if (cd < th - kSpEps || cd == th) {  // (a cost equal to the threshold: a pair of weight 0, left to the certificate)
        if (ne < kSpK) { w.strow[static_cast<size_t>(j) * kSpK + ne] = i; w.stcost[static_cast<size_t>(j) * kSpK + ne] = c; ++ne; }
        else bad |= 2;
      } else if (!(cd > th + kSpEps)) bad |= 1;
    }
    w.y[j] = ne;
  }
  bad = g.reduce_max(bad);
  mn = g.reduce_min(mn);
  g.sync();
  if (bad) return SparseEnum((bad & 1) ? 0 : -10, mn);
  return SparseEnum(sparse_build_csr(g, w, nc) ? 1 : -11, mn);
}

// ---- 2-4. matching over the viable pairs + certificate ---------------------------------------------------------------
struct SparseProf {  // diagnostics: shader cycles of the three stages, path searches and column scans
  long long c_init = 0, c_search = 0, c_cert = 0;
  int n_search = 0, n_scan = 0;
};
// 2. column duals, proposals, the columns left to insert (returns their number; the list is w.freel)
template <class G, class W>
MOT_DEV int sparse_init(G& g, const W& w, int nr, int nc, float thresh) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  for (int i = t; i < nr; i += T) { w.u[i] = 0.0; w.x[i] = kSpIntMax; w.slot[i] = 0; }
  g.sync();
  // column duals and proposals
  for (int j = t; j < nc; j += T) {
    float bc = 0.0f;
    int br = -1;
    const int ee = w.eoff[j], e0 = sp_e0(ee), e1 = e0 + sp_deg(ee);
    for (int k = e0; k < e1; ++k) {
      const int r = w.erow[k];
      const float c = w.ecost[k];
      if (br < 0 || c < bc || (c == bc && r < br)) { bc = c; br = r; }
    }
    w.v[j] = (br >= 0) ? static_cast<double>(bc) - th : 0.0;
    w.y[j] = br;
    if (br >= 0) G::atomic_min(w.x.raw(br), j);
  }
  g.sync();
  int nfree = 0;
  for (int j0 = 0; j0 < nc; j0 += T) {  // losers of a conflict, in ascending column order
    const int j = j0 + t;
    bool lose = false;
    if (j < nc) {
      const int br = w.y[j];
      if (br >= 0 && static_cast<int>(w.x[br]) != j) { lose = true; w.y[j] = -1; }
    }
    int tot;
    const int pos = g.flag_rank(lose, &tot);
    if (lose) w.freel[nfree + pos] = j;
    nfree += tot;
  }
  g.sync();
  for (int i = t; i < nr; i += T)
    if (static_cast<int>(w.x[i]) == kSpIntMax) w.x[i] = -1;
  if (t == 0) { w.ctr[1] = 0; w.ctr[3] = 1; }  // retry list of the concurrent searches, their common outcome
  g.sync();

  return nfree;
}

// 2b. (round 6) The SHORT searches, one LANE each. A free column j0 lost its best row r1 to column k (that is what "free" means after the
// proposals); nearly every search then ends within two steps of the shortest-path search: j0 takes another row that is free, or stays unmatched,
// or k moves to a free row of its own (or becomes unmatched) and j0 takes r1. The wavefront-wide search below spends ~9 k cycles on such a
// search (compare-and-swap claims, five cross-lane pushes, a lexicographic DPP minimum and a barrier per step): a third of the kernel at the
// north-star shape. Here every lane runs the first two steps of the SAME search for its own free column over the pairs of j0 and of k — same
// labels, same picks (nearest, ties to the lowest row), same dual updates — claiming every row it reads with a compare-and-swap on its slot
// word exactly as the concurrent searches do, and gives everything back untouched when a row belongs to somebody else or the search would need
// a third step; those columns stay on the free list for the search below. Whatever happens here, the certificate checks the final matching
// and duals (optimality and uniqueness), not how they were found. Returns the number of columns left on the list.
constexpr int kSpShortDeg = 8;  // pairs per column the short searches handle (a column with more is left to the wavefront's search)
template <class G, class W>
MOT_DEV int sparse_short_searches(G& g, const W& w, int nfree, float thresh) {
  if constexpr (std::remove_cv_t<std::remove_reference_t<decltype(w.u)>>::kSpace == kMemGlobal) return nfree;  // (state in global scratch: lanes of different wavefronts would need fences; rare, large problems)
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  const int mytag = 0x40000000 | (t << 8) | 0xfe;
  constexpr int KD = kSpShortDeg;
  int nleft = 0;
  for (int f0 = 0; f0 < nfree; f0 += T) {
    const int f = f0 + t;
    const bool have = f < nfree;
    const int j0 = have ? static_cast<int>(w.freel[f]) : 0;
    bool done = false;
    if (have) {
      // Every dependent LDS round trip is taken ONCE for all pairs of a column (rows, then claims, then costs and duals): a lane's search is a
      // chain of about ten round trips instead of three per pair.
      const double v0 = w.v[j0];
      const int ee = w.eoff[j0], a0 = sp_e0(ee), na = sp_deg(ee);
      bool ok = na <= KD;
      int ra[KD];
      bool mine_a[KD];
      double la[KD];
#pragma unroll
      for (int i = 0; i < KD; ++i) { ra[i] = (ok && i < na) ? static_cast<int>(w.erow[a0 + i]) : -1; mine_a[i] = false; la[i] = 1e300; }
#pragma unroll
      for (int i = 0; i < KD; ++i)
        if (ra[i] >= 0) { const int old = w.slot.atomic_cas(ra[i], 0, mytag); mine_a[i] = old == 0; ok = ok && old == 0; }
      double d1 = 1e300, d2 = 1e300;
      int r1 = kNoIdx, r2 = kNoIdx;
      if (ok) {
#pragma unroll
        for (int i = 0; i < KD; ++i)
          if (ra[i] >= 0) la[i] = (static_cast<double>(static_cast<float>(w.ecost[a0 + i])) - th) - static_cast<double>(w.u[ra[i]]) - v0;
#pragma unroll
        for (int i = 0; i < KD; ++i)
          if (ra[i] >= 0) {
            const double nd = la[i];
            const int r = ra[i];
            if (nd < d1 || (nd == d1 && r < r1)) { d2 = d1; r2 = r1; d1 = nd; r1 = r; }
            else if (nd < d2 || (nd == d2 && r < r2)) { d2 = nd; r2 = r; }
          }
      }
      double L = -v0;          // leave j0 unmatched
      int term_row = -1, term_col = j0, term_pred = j0;
      int kcol = -1;
      bool scanned = false;
      int rb[KD];
      bool mine_b[KD];
#pragma unroll
      for (int i = 0; i < KD; ++i) { rb[i] = -1; mine_b[i] = false; }
      if (ok && r1 != kNoIdx && d1 < L) {
        const int x1 = w.x[r1];
        if (x1 < 0) { L = d1; term_row = r1; term_col = -1; term_pred = j0; }
        else {
          // r1 is scanned: its column k joins the tree
          kcol = x1;
          scanned = true;
          const double D = d1;
          const double vk = w.v[kcol];
          const double cand = D - vk;
          if (cand < L) { L = cand; term_row = -1; term_col = kcol; }
          // step 2: relax k's pairs; the nearest unscanned row over j0's other rows and k's rows
          const int ek = w.eoff[kcol], b0 = sp_e0(ek), nb = sp_deg(ek);
          ok = nb <= KD;
#pragma unroll
          for (int i = 0; i < KD; ++i) { rb[i] = (ok && i < nb) ? static_cast<int>(w.erow[b0 + i]) : -1; if (rb[i] == r1) rb[i] = -1; }
#pragma unroll
          for (int i = 0; i < KD; ++i)
            if (rb[i] >= 0) { const int old = w.slot.atomic_cas(rb[i], 0, mytag); mine_b[i] = old == 0; ok = ok && (old == 0 || old == mytag); }
          double bd = d2;
          int brow = r2, bpred = j0;
          if (ok) {
            double lb[KD];
#pragma unroll
            for (int i = 0; i < KD; ++i)
              lb[i] = (rb[i] >= 0) ? D + ((static_cast<double>(static_cast<float>(w.ecost[b0 + i])) - th) - static_cast<double>(w.u[rb[i]]) - vk) : 1e300;
#pragma unroll
            for (int i = 0; i < KD; ++i)
              if (rb[i] >= 0) {
                double nd = lb[i];
                int pred = kcol;
#pragma unroll
                for (int q = 0; q < KD; ++q)  // also a row of j0: the label it already has stays unless this one is smaller
                  if (ra[q] == rb[i] && !(nd < la[q])) { nd = la[q]; pred = j0; }
                if (nd < bd || (nd == bd && rb[i] < brow)) { bd = nd; brow = rb[i]; bpred = pred; }
              }
            if (brow != kNoIdx && bd < L) {
              if (static_cast<int>(w.x[brow]) < 0) { L = bd; term_row = brow; term_col = -1; term_pred = bpred; }
              else ok = false;  // a third step: the wavefront's search takes this column
            }
          }
        }
      }
      if (ok) {
        // duals: the scanned row and its column move by (L - label), the source by L (as the search below)
        if (scanned) { const double dl = L - d1; w.u[r1] -= dl; w.v[kcol] += dl; }
        w.v[j0] += L;
        if (term_row >= 0) {
          if (term_pred == j0) { w.y[j0] = term_row; w.x[term_row] = j0; }
          else { w.y[kcol] = term_row; w.x[term_row] = kcol; w.y[j0] = r1; w.x[r1] = j0; }
        } else if (term_col != j0) { w.y[kcol] = -1; w.y[j0] = r1; w.x[r1] = j0; }
        done = true;
      }
#pragma unroll
      for (int i = 0; i < KD; ++i) {
        if (mine_a[i]) w.slot[ra[i]] = 0;
        if (mine_b[i]) w.slot[rb[i]] = 0;
      }
    }
    int tot;
    const int pos = g.flag_rank(have && !done, &tot);
    if (have && !done) w.freel[nleft + pos] = j0;  // (in place: nleft + pos <= f, and the entries of later chunks lie behind this chunk)
    nleft += tot;
  }
  g.sync();
  return nleft;
}

// 3. a shortest augmenting path per remaining column. The group may be a single wavefront of a larger workgroup (the searches
// are serial; their reductions then stay inside the wavefront). Returns 1, or -2 when a search reached too many rows.
// The rows a search has reached live in the lanes' registers — lane q is slot q: row, label, the column it was reached from,
// its current column, state — so that picking the nearest one is a register reduction and relaxing a column costs three
// LDS round trips (the column's list entry and dual; its pairs; their rows' duals and slots); labels travel between lanes
// by broadcast. Needs at least kSpSlots lanes.
//
// SHARED (round 4): several wavefronts of one workgroup search at the same time, wavefront `first` of `stride` taking the free columns
// first, first + stride, ... Two searches that reach disjoint rows do not see each other at all (a search reads and writes the duals
// and assignments of the rows it has reached, of their columns and of its own start column only), and on tracking problems they nearly
// always are disjoint: a detection is viable for a handful of tracks. A row is therefore CLAIMED (compare-and-swap on its slot word,
// tagged with the wavefront) before anything of it is read; a search that meets a row claimed by another wavefront gives every row
// back untouched — nothing is modified before a search has finished — and leaves its column on the retry list (w.arcs, counted in
// w.ctr[1]), which one wavefront works off alone afterwards. Whatever the interleaving, the result is checked like any other: the
// certificate tests optimality and uniqueness of the final matching and duals, not how they were found.
template <bool SHARED, class G, class W>
MOT_DEV int sparse_search_impl(G& g, const W& w, int nfree, float thresh, int first, int stride, bool from_retry, int* scans, long long* seg) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  int n_scan = 0;
  if (T < kSpSlots + 1) return (nfree > 0) ? -2 : 1;
  const int tag = SHARED ? ((first + 1) << 8) : 0;
  auto column_at = [&](int f) { return from_retry ? static_cast<int>(w.arcs[f]) : static_cast<int>(w.freel[f]); };
  int j_next = (first < nfree) ? column_at(first) : 0;
  for (int f = first; f < nfree; f += stride) {
    const int j0 = j_next;
    if (f + stride < nfree) j_next = column_at(f + stride);  // (global memory: fetched one search ahead)
    double L = -static_cast<double>(w.v[j0]);  // leave j0 unmatched
    int term_slot = -1, term_col = j0;          // terminal: the slot of a free row, or the column that ends unmatched
    int cur = j0;
    double D = 0.0;
    int nslots = 0;
    // this lane's slot
    int my_row = -1, my_pred = -1, my_x = -1, my_state = 0;  // state: 0 empty, 1 reached, 2 scanned
    double my_dist = 0.0;
    bool overflow = false, conflict = false;
    for (;;) {
      ++n_scan;
      const long long q0 = MOT_CLOCK();
      // relax the viable pairs of column `cur`: lane k < deg handles pair k
      const double vc = w.v[cur];
      const int ee = w.eoff[cur], e0 = sp_e0(ee), deg = sp_deg(ee);
      int er = -1, es = 0, ex = -1;
      double nd = 0.0;
      bool mine_new = false, clash = false;
      if (t < deg) {
        er = w.erow[e0 + t];
        if constexpr (SHARED) {  // the row is claimed before anything of it is read
          const int old = w.slot.atomic_cas(er, 0, tag | 0xff);
          if (old == 0) mine_new = true;
          else if ((old & ~0xff) == tag) es = old & 0xff;
          else clash = true;
        }
      }
      if constexpr (SHARED) {
        if (g.ballot(clash) != 0ull) {  // somebody else's row: everything goes back as it was
          if (mine_new) w.slot[er] = 0;
          conflict = true;
          break;
        }
        g.sync();  // (acquire: the previous owner's writes to a row just claimed are visible)
      }
      if (t < deg) {
        const double red = (static_cast<double>(static_cast<float>(w.ecost[e0 + t])) - th) - static_cast<double>(w.u[er]) - vc;
        nd = D + red;
        if constexpr (!SHARED) es = w.slot[er];
        ex = w.x[er];
      }
      const bool need = SHARED ? mine_new : (t < deg && es == 0);
      int tot;
      const int pos = g.flag_rank(need, &tot);
      if (nslots + tot > kSpSlots) { overflow = true; if (SHARED && mine_new) w.slot[er] = 0; break; }
      if (need) w.slot[er] = tag | (nslots + pos + 1);
      const long long q1 = MOT_CLOCK();
      // labels to their slots: a pair whose row already has one improves it; the others open slots nslots, nslots + 1, ...
      // (each pair lane pushes its label to the lane that owns the slot: distinct rows, distinct destinations)
      {
        const bool send = t < deg;
        const int dstl = (es != 0) ? es - 1 : nslots + pos;
        const int got = g.push_i32(1 + ((es == 0) ? 1 : 0), dstl, send);  // 0 nothing, 1 an improvement offer, 2 a new slot
        const int g_lo = g.push_i32(__builtin_bit_cast(long long, nd) & 0xffffffffll, dstl, send);
        const int g_hi = g.push_i32(static_cast<int>(__builtin_bit_cast(long long, nd) >> 32), dstl, send);
        const int g_row = g.push_i32(er, dstl, send);
        const int g_x = g.push_i32(ex, dstl, send);
        if (got != 0) {
          const double kd = __builtin_bit_cast(double, (static_cast<long long>(g_hi) << 32) | static_cast<long long>(static_cast<unsigned>(g_lo)));
          if (got == 2) { my_row = g_row; my_dist = kd; my_pred = cur; my_x = g_x; my_state = 1; }
          else if (my_state == 1 && kd < my_dist) { my_dist = kd; my_pred = cur; }
        }
      }
      nslots += tot;
      const long long q2 = MOT_CLOCK();
      // nearest reached, not yet scanned row (ties: lowest row index)
      double bd = (my_state == 1) ? my_dist : 1e300;
      int brow = (my_state == 1) ? my_row : kNoIdx;  // (kNoIdx: takes no part)
      g.reduce_lexmin(bd, brow);
      const long long q3 = MOT_CLOCK();
      if (seg) { seg[0] += q1 - q0; seg[1] += q2 - q1; seg[2] += q3 - q2; }
      if (brow == kNoIdx || !(bd < L)) break;
      const unsigned long long own = g.ballot(my_state == 1 && my_row == brow);
      const int q = __builtin_ctzll(own);
      const int xc = g.bcast_i32(my_x, q);
      if (xc < 0) { L = bd; term_slot = q; term_col = -1; break; }
      if (t == q) my_state = 2;
      cur = xc;
      D = bd;
      const double cand = D - static_cast<double>(w.v[cur]);
      if (cand < L) { L = cand; term_slot = -1; term_col = cur; }
    }
    if (overflow || conflict) {  // nothing has been modified yet: the rows go back
      if (my_state != 0) w.slot[my_row] = 0;
      g.sync();
      if (overflow) return -2;
      if (t == 0) { const int rp = w.ctr.atomic_add(1, 1); w.arcs[rp] = j0; }
      continue;
    }
    const long long q4 = MOT_CLOCK();
    // duals: scanned rows and their columns move by (L - label); the source by L
    if (my_state == 2) {
      const double dl = L - my_dist;
      w.u[my_row] -= dl;
      w.v[my_x] += dl;
    }
    if (t == 0) w.v[j0] += L;
    g.sync();
    // augment: the path is walked by everyone in step (slots are read by broadcast), one lane writes
    {
      int q = -1;
      if (term_slot >= 0) q = term_slot;
      else if (term_col != j0) {
        const int r = w.y[term_col];
        g.sync();
        if (t == 0) w.y[term_col] = -1;
        q = (static_cast<int>(w.slot[r]) & 0xff) - 1;
      }
      while (q >= 0) {
        const int r = g.bcast_i32(my_row, q), p = g.bcast_i32(my_pred, q);
        const int rn = w.y[p];
        g.sync();
        if (t == 0) { w.y[p] = r; w.x[r] = p; }
        if (p == j0) break;
        q = (static_cast<int>(w.slot[rn]) & 0xff) - 1;
      }
    }
    g.sync();  // (release: the duals and assignments are written before the rows are given back)
    if (my_state != 0) w.slot[my_row] = 0;
    g.sync();
    if (seg) seg[3] += MOT_CLOCK() - q4;
  }
  if (scans) *scans = n_scan;
  return 1;
}
template <class G, class W>
MOT_DEV int sparse_search(G& g, const W& w, int nfree, float thresh, int* scans = nullptr, long long* seg = nullptr) {
  return sparse_search_impl<false>(g, w, nfree, thresh, 0, 1, false, scans, seg);
}

// 4. certificate: 1 when w.x / w.y is the unique optimum by more than kSpEps, else -3 / -4 / -5
template <class G, class W>
MOT_DEV int sparse_certify(G& g, const W& w, int nr, int nc, float thresh) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  enum : int { kStart = 1, kEnd = 2, kReach = 4, kArrived = 8, kFreeCol = 16, kIn = 32 };
  for (int i = t; i < nr; i += T) {
    const int xc = w.x[i];
    int fl = 0;
    if (xc < 0) fl |= kStart;
    else {
      if (-static_cast<double>(w.v[xc]) <= kSpEps) fl |= kStart;  // its column may be left unmatched for free
      if (-static_cast<double>(w.u[i]) <= kSpEps) fl |= kEnd;     // the row may be left unmatched for free
    }
    w.slot[i] = fl;
  }
  if (t == 0) { w.ctr[0] = 0; w.ctr[1] = 0; }
  g.sync();
  int bad = 0;
  const int arc_cap = sparse_arc_cap(nr, nc);
  for (int j = t; j < nc; j += T) {
    const double vj = w.v[j];
    const int yj = w.y[j];
    if (yj < 0 && !(vj >= -kSpTol && vj <= kSpTol)) bad = 1;  // an unmatched column carries no dual
    if (!(vj <= kSpTol)) bad = 1;
    const int ee = w.eoff[j], e0 = sp_e0(ee), e1 = e0 + sp_deg(ee);
    for (int k = e0; k < e1; ++k) {
      const int r = w.erow[k];
      const double red = (static_cast<double>(static_cast<float>(w.ecost[k])) - th) - static_cast<double>(w.u[r]) - vj;
      if (r == yj) { if (!(red >= -kSpTol && red <= kSpTol)) bad = 1; continue; }
      if (!(red >= -kSpTol)) { bad = 1; continue; }
      if (red <= kSpEps) {
        if (yj < 0) G::atomic_or(w.slot.raw(r), kFreeCol | kEnd);
        else {
          const int a = G::atomic_add(w.ctr.raw(0), 1);
          if (a < arc_cap) { w.arcs[2 * a] = r; w.arcs[2 * a + 1] = yj; }
        }
      }
    }
  }
  for (int i = t; i < nr; i += T) {
    const double ui = w.u[i];
    if (!(ui <= kSpTol)) bad = 1;
    if (static_cast<int>(w.x[i]) < 0 && !(ui >= -kSpTol)) bad = 1;
  }
  bad = g.reduce_max(bad);
  g.sync();
  const int narcs = w.ctr[0];
  if (bad) return -3;
  if (narcs > arc_cap) return -4;
  int nonuniq = 0;
  for (int i = t; i < nr; i += T) {
    const int fl = w.slot[i];
    if ((fl & kStart) && (fl & kEnd) && static_cast<int>(w.x[i]) >= 0) nonuniq = 1;  // a matched pair of weight ~0: dropping it is free
    if ((fl & kStart) && (fl & kFreeCol)) { nonuniq = 1; SPDBG("k0 row %d x %d u %.17g\n", i, (int)w.x[i], (double)w.u[i]); }  // changes its column for free
  }
  if (narcs > 0) {
    // reachability from the rows that can start a path
    for (int a = t; a < narcs; a += T) {
      const int s = w.arcs[2 * a];
      if (static_cast<int>(w.slot[s]) & kStart) G::atomic_or(w.slot.raw(s), kReach);
    }
    g.sync();
    for (int it = 0; it <= narcs; ++it) {
      int ch = 0;
      for (int a = t; a < narcs; a += T) {
        const int s = w.arcs[2 * a], d = w.arcs[2 * a + 1];
        if ((static_cast<int>(w.slot[s]) & kReach) && !(static_cast<int>(w.slot[d]) & kArrived)) { G::atomic_or(w.slot.raw(d), kArrived | kReach); ch = 1; }
      }
      ch = g.reduce_max(ch);
      g.sync();
      if (!ch) break;
    }
    for (int a = t; a < narcs; a += T) {
      const int fl = w.slot[static_cast<int>(w.arcs[2 * a + 1])];
      if ((fl & kArrived) && (fl & kEnd)) { nonuniq = 1; SPDBG("path end row %d fl %d\n", (int)w.arcs[2 * a + 1], fl); }
    }
    // alternating cycle: peel arcs whose source has no live incoming arc; anything left lies on or behind a cycle
    // (a dead arc keeps its source as -1 - source)
    int alive_n = narcs;
    for (int it = 0; it <= narcs && alive_n > 0; ++it) {
      for (int a = t; a < narcs; a += T) {
        G::atomic_and(w.slot.raw(static_cast<int>(w.arcs[2 * a + 1])), ~kIn);
        const int s = w.arcs[2 * a];
        if (s >= 0) G::atomic_and(w.slot.raw(s), ~kIn);
      }
      g.sync();
      for (int a = t; a < narcs; a += T)
        if (static_cast<int>(w.arcs[2 * a]) >= 0) G::atomic_or(w.slot.raw(static_cast<int>(w.arcs[2 * a + 1])), kIn);
      g.sync();
      int killed = 0, live = 0;
      for (int a = t; a < narcs; a += T) {
        const int s = w.arcs[2 * a];
        if (s < 0) continue;
        if (!(static_cast<int>(w.slot[s]) & kIn)) { w.arcs[2 * a] = -1 - s; ++killed; }
        else ++live;
      }
      int tk, tl;
      g.exclusive_scan(killed, &tk);
      g.exclusive_scan(live, &tl);
      alive_n = tl;
      g.sync();
      if (tk == 0) break;
    }
    if (alive_n > 0) { nonuniq = 1; SPDBG("cycle alive %d\n", alive_n); }
  }
  nonuniq = g.reduce_max(nonuniq);
  g.sync();
  return nonuniq ? -5 : 1;
}


// 2-4 by one group. Returns 1 with w.x / w.y holding THE minimum-weight matching (unique by more than kSpEps); else a reason
// <= 0 for the exact path to take over (-2 a search reached too many rows, -3 certificate arithmetic, -4 too many tight
// pairs, -5 not unique).
template <class G, class W>
MOT_DEV int sparse_solve(G& g, const W& w, int nr, int nc, float thresh, SparseProf* prof = nullptr) {
  const long long pc0 = MOT_CLOCK();
  int nfree = sparse_init(g, w, nr, nc, thresh);
  nfree = sparse_short_searches(g, w, nfree, thresh);
  const long long pc1 = MOT_CLOCK();
  int scans = 0;
  const int rs = sparse_search(g, w, nfree, thresh, &scans);
  const long long pc2 = MOT_CLOCK();
  if (prof) { prof->c_init = pc1 - pc0; prof->c_search = pc2 - pc1; prof->n_search = nfree; prof->n_scan = scans; }
  if (rs != 1) return rs;
  const int rc = sparse_certify(g, w, nr, nc, thresh);
  if (prof) prof->c_cert = MOT_CLOCK() - pc2;
  return rc;
}

}  // namespace mot
