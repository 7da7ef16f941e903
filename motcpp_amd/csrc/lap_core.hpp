// Exact workgroup-parallel Jonker-Volgenant assignment for the thresholded rectangular problem
// that motcpp's utils::linear_assignment poses (src/utils/matching.cpp:14-60 →
// include/motcpp/association/lap_solver.hpp:251-332).
//
// What "exact" means here: the reference extends the nr x nc float cost to an (nr+nc)^2 double
// matrix (off-diagonal blocks = thresh/2, bottom-right block = 0) and runs lapjv on it; its
// result depends on lapjv's scan orders whenever costs tie — and the extension is one huge tie.
// This solver reproduces lapjv's decisions (same phases, same tie-breaks: lowest-index column
// minima, (value,index)-lexicographic top-2 in the row reductions, the cols[] permutation order
// in the shortest-path search) so x/y are identical to the reference for every input, without
// materialising the extension (its constant blocks are folded into the accessor) — memory stays
// O(nr*nc) + O(nr+nc).
//
// Parallel shape: ONE workgroup per problem; extended columns are dealt round-robin to lanes
// (column j belongs to lane j % T for the whole solve, so per-column duals need no barrier
// between an update and its next read); every O(n) scan of lapjv becomes a strided loop plus
// one wavefront/LDS reduction (grp.hpp). Sequential dependencies between rows are kept.
#pragma once
#include <type_traits>
#ifndef MOT_LAP_TIE_PER  // (the host-emulation tests lower both so that small problems take the closed-form tie runs)
#define MOT_LAP_TIE_PER 8
#endif
#ifndef MOT_LAP_TIE_MIN
#define MOT_LAP_TIE_MIN 128
#endif
#include "grp.hpp"
#include "mem.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define MOT_CLOCK() static_cast<long long>(__builtin_readcyclecounter())
#else
#define MOT_CLOCK() 0LL
#endif
// clock reads inside the shortest-path search (several per scan step; each is a scalar memory operation the wavefront waits
// for): only in builds made with -DMOT_LAP_FINE_PROF
#if defined(MOT_LAP_FINE_PROF)
#define MOT_FCLOCK() MOT_CLOCK()
#else
#define MOT_FCLOCK() 0LL
#endif

namespace mot {

// Cost functors. A functor exposes the real nr x nc block of the problem:
//   Row row(i)        row-invariant context (pointer to the matrix row, or the row's box in registers) — plain data,
//                     no pointer back to the functor, so that functor and context both stay in registers
//   at(row, j)        element (i, j) as double (the conversion of a float: every cost is a float, lap_solver.hpp:303)
//   at(i, j)          same, without a row context (column scans)
// MatrixCost reads a materialised float matrix; IouCost (lap_cost.hpp) recomputes IoU-family costs from boxes
// staged in LDS, so the N x M matrix never exists in memory.
struct MatrixCost {
  static constexpr int kRPL = 0;  // no lane-owned column cache
  static constexpr bool kMatrix = true;  // rows are loads from a float matrix: worth prefetching a sweep ahead
  static constexpr bool kPlain = false;
  const float* cost;  // nr x nc, row-major, leading dimension ld (global memory)
  int ld;
  struct Row { const float* p; };
  MOT_DEV Row row(int i) const { return Row{cost + static_cast<size_t>(i) * ld}; }
  MOT_DEV double at(const Row& r, int j) const { return static_cast<double>(gld(r.p, j)); }
  MOT_DEV double at_owned(const Row& r, int, int j) const { return at(r, j); }
  MOT_DEV float at_owned_f(const Row& r, int, int j) const { return gld(r.p, j); }
  MOT_DEV double at(int i, int j) const { return static_cast<double>(gld(cost, static_cast<size_t>(i) * ld + j)); }
};
template <class Cost, class = void>
struct is_matrix_cost { static constexpr bool value = false; };
template <class Cost>
struct is_matrix_cost<Cost, decltype(void(Cost::kMatrix))> { static constexpr bool value = Cost::kMatrix; };
template <class G, class = void>
struct has_wave_table { static constexpr bool value = false; };
template <class G>
struct has_wave_table<G, decltype(void(G::kWaveTable))> { static constexpr bool value = G::kWaveTable; };
#if !defined(__HIPCC__)
// (host emulation only: how often the search skipped a real row's sweep through the nolow flags — tests assert that the path runs)
inline long& lap_dbg_void_real() { static long c = 0; return c; }
#define MOT_LAP_DBG_VOID(n) do { if (t == 0) lap_dbg_void_real() += (n); } while (0)
#else
#define MOT_LAP_DBG_VOID(n) ((void)0)
#endif
struct LapDims {
  int nr, nc;
  double half;  // thresh / 2 (lap_solver.hpp:300)
};
// Workspace, each array n = nr + nc long. The HOT arrays are touched by every row pass and live in LDS when
// the problem fits; the COLD arrays are only used by the general shortest-path search (rare on tracking costs)
// and the phase-1 row list, and always live in global scratch.
// VS = address space of v/y, XS = of x/fr, DS = of d (mem.hpp); KS = of cols/inv (round 5: LDS for the wide matrix problems), IT = element
// type of y/cols/inv (short there: 2 B per extended row, n <= kFsMaxN), NS = of the byte-sized list lengths, YS = of the matched costs; the
// other cold arrays are always global.
template <int VS, int XS, int DS = kMemGlobal, int BS = kMemGlobal, int CS = kMemGlobal, int KS = CS, class IT = int, int NS = kMemGlobal, int YS = kMemGlobal>
struct LapWorkT {
  using idx_t = IT;
  static constexpr int kColsSpace = KS;
  static constexpr bool kLst16 = sizeof(IT) == 2;  // _find_dense's record list as 16-bit entries in lst16 (LDS) instead of lst
  // hot
  MemPtr<double, VS> v;   // column duals
  MemPtr<int, XS> x;      // row -> col (extended)
  MemPtr<IT, VS> y;       // col -> row (extended)
  MemPtr<int, XS> fr;     // free-row list (doubles as the column-hit counter in phase 1)
  // cold
  MemPtr<double, DS> d;           // shortest-path distances (LDS for the wide matrix problems: every scan step reads and writes them)
  MemPtr<int, CS> pred;   // path predecessors (CS: LDS in the all-LDS mode of the launches behind the fast path — every sweep of the
                          // shortest-path search reads inv[] / d[] and a member costs a cols[] -> d[] chain: global round trips otherwise)
  MemPtr<IT, KS> cols;    // lapjv's column permutation / phase-1 unique-row list
  MemPtr<int, CS> tmp;    // tie flags (slow path)
  MemPtr<int, CS> lst;    // compacted tie positions (slow path)
  MemPtr<double, BS> rlb; // per real row: lower bound of its reduced costs over the real columns, see "hopeless rows" (read and written by every
                          // serial round of the row reduction: LDS with the full LDS state — a global round trip per round is what a round then costs)
  MemPtr<IT, KS> inv;     // inverse of cols[] (position of a column), slow path
  MemPtr<int, CS> tie;    // tie flags by position during a scan (all zero between scans), slow path
  MemPtr<int, kMemGlobal> sa, sb, sc;  // staging of the closed-form tie runs that do not fit in registers (slow path)
  MemPtr<float, kMemGlobal> rmin; // per real row: minimum RAW cost over the real columns (phase 1, register-cached on-the-fly costs), see "void real rows"
  // optional (null: the parallel scan steps and the sparse real-row sweeps are off) — see "row lists" in lap_solve
  MemPtr<int, kMemGlobal> rl_cnt;     // [nr] entries of real row i with cost < half (may exceed kRlCap: then the row has no usable list)
  MemPtr<unsigned long long, kMemGlobal> rl_ent;  // [nr][kRlCap] their (column | cost bits << 32): one 8-byte load per entry, a row's first 16 entries in one 128-byte line
  MemPtr<int, VS == kMemAny ? kMemAny : kMemLds> fsw;  // kFsWsInts ints of fast scratch (always LDS on the device): step members + tie events
  // optional (round 5, with the row lists): what the scan steps would otherwise fetch from global memory per member and per tie event
  MemPtr<unsigned short, KS> lst16;  // (kLst16) [n] _find_dense's record list
  MemPtr<unsigned char, NS> rl_n;  // [nr] min(rl_cnt, 255), filled after phase 1
  MemPtr<float, YS> ycost;         // [nc] cost(y[j], j) while real column j is held by a real row (a random read of the N x M matrix otherwise: one
                                   // 128-byte line of a 32 MB matrix per member — what evicted a problem's working set from its XCD's L2)
  bool cyc_ext = false;      // cyc has 36 entries: [16..23] cycles inside phase 3 (step classification, dry run, apply, event sort, event replay, _find_dense, one-at-a-time sweeps, search set-up)
  long long* cyc = nullptr;  // optional profiling [16]: [0..3] cycles in phase 1a (column minima), 1b (transfer), 2, 3; [4..7] n_uniq, serial row-reduction rounds, serial augmentations, n;
                             // [8..15] shortest-path scans: parallel steps, members they consumed, real rows among them, tie events, one-at-a-time sweeps, steps refused (rounding), _find_dense calls, row lists in use
};
using LapWork = LapWorkT<kMemAny, kMemAny, kMemAny, kMemAny, kMemAny>;
MOT_HD size_t lap_hot_bytes(int n) { return static_cast<size_t>(n) * (sizeof(double) + 3 * sizeof(int)); }
MOT_HD size_t lap_cold_bytes(int n) { return static_cast<size_t>(n) * (2 * sizeof(double) + 9 * sizeof(int) + sizeof(float)); }
// Row lists (optional scratch of a matrix-cost task, mot_lap_task.rowlist): per real row the entries below thresh/2.
constexpr int kRlCap = 64;      // entries kept per row (a row with more has no list: its sweeps stay dense)
constexpr int kFsIter = 4;      // list entries a lane holds in registers during a parallel scan step (8, i.e. 256 units per step, was measured in round 5: the
                                // relaxations of so many rows overflow kKeepCap, the step falls back to ONE row, and the solve takes five times as many steps)
constexpr int kFsUnit = 16;     // the lists are dealt to the lanes in units of this many entries (kRlCap / kFsUnit <= 4 units per row)
constexpr int kFsMaxUnits = 256;    // units of one step (also bounded by kFsIter * T / kFsUnit)
constexpr int kFsMaxMembers = 256;  // real-row members of one step (each takes at least one unit)
constexpr int kEvCap = 256;     // tie events one step may produce (more: the step shrinks to one member)
constexpr int kKeepCap = 512;   // relaxations of one step that lower a distance (more: the step shrinks to one member)
constexpr int kFsHash = 1024;   // slots of the per-step column table (>= 2 * kKeepCap, a power of two)
constexpr int kFsMaxN = 8192;   // extended size up to which the TODO bitmask fits
// fast scratch (ints): members (q, row, h as 2 ints) | units (row, rank + chunk + list length) | counters | column table of a step (key, earliest member) |
// event list (q, j, k, row, cost, list length) | sorted events (q, j, k, flags, row, cost, list length) + head slots (column, event) | TODO bitmask
constexpr int kFsM = 0, kFsUnits = 4 * kFsMaxMembers, kFsCtr = kFsUnits + 2 * kFsMaxUnits, kFsKeep = kFsCtr + 8, kFsEvl = kFsKeep + 2 * kFsHash, kFsEvs = kFsEvl + 6 * kEvCap,
              kFsTodo = kFsEvs + 9 * kEvCap, kFsWsInts = kFsTodo + kFsMaxN / 32 + 8;
MOT_HD unsigned long long rl_pack(int col, float cost) {
  return static_cast<unsigned long long>(static_cast<unsigned>(col)) | (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, cost)) << 32);
}
MOT_HD int rl_col_of(unsigned long long e) { return static_cast<int>(static_cast<unsigned>(e & 0xffffffffull)); }
MOT_HD float rl_cost_of(unsigned long long e) { return __builtin_bit_cast(float, static_cast<unsigned>(e >> 32)); }
MOT_HD size_t lap_rowlist_bytes(int nr) { return static_cast<size_t>(nr > 0 ? nr : 0) * (4 + 8 * static_cast<size_t>(kRlCap)) + 160; }
template <class Work>
MOT_HD void lap_carve_rowlist(Work& w, void* base, int nr) {
  char* p = static_cast<char*>(base);
  w.rl_cnt.p = reinterpret_cast<int*>(p); p += ((sizeof(int) * static_cast<size_t>(nr > 0 ? nr : 0)) + 127) & ~size_t(127);  // (a row's entries start on a line)
  w.rl_ent.p = reinterpret_cast<unsigned long long*>(p);
}
MOT_HD size_t lap_work_bytes(int n) { return lap_hot_bytes(n) + lap_cold_bytes(n); }
template <class Work>
MOT_HD void lap_carve_hot(Work& w, void* base, int n) {
  char* p = static_cast<char*>(base);
  w.v.p = reinterpret_cast<double*>(p); p += sizeof(double) * n;
  w.x.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.y.p = reinterpret_cast<decltype(w.y.p)>(p); p += sizeof(int) * n;  // (an int's room per entry whatever the element type)
  w.fr.p = reinterpret_cast<int*>(p);
}
template <class Work>
MOT_HD void lap_carve_cold(Work& w, void* base, int n) {
  char* p = static_cast<char*>(base);
  w.d.p = reinterpret_cast<double*>(p); p += sizeof(double) * n;
  w.rlb.p = reinterpret_cast<double*>(p); p += sizeof(double) * n;
  w.pred.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.cols.p = reinterpret_cast<decltype(w.cols.p)>(p); p += sizeof(int) * n;
  w.tmp.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.lst.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.inv.p = reinterpret_cast<decltype(w.inv.p)>(p); p += sizeof(int) * n;
  w.tie.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.sa.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.sb.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.sc.p = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.rmin.p = reinterpret_cast<float*>(p);
}
MOT_HD LapWork lap_carve(void* base, int n) {  // hot then cold, contiguous
  LapWork w;
  lap_carve_hot(w, base, n);
  lap_carve_cold(w, static_cast<char*>(base) + ((lap_hot_bytes(n) + 7) & ~size_t(7)), n);
  return w;
}

// Row view of the extended matrix (lap_solver.hpp:303-315): real rows are [cost | half...],
// dummy rows are [half... | 0...].
template <class Cost>
struct ExtRow {
  typename Cost::Row r;
  bool real;
  double left, right;  // dummy row: value for j < nc; any row: value for j >= nc
  int nc;
  MOT_DEV double at(const Cost& C, int j) const {
    if (j < nc) return real ? C.at(r, j) : left;
    return right;
  }
};
template <class Cost>
MOT_DEV ExtRow<Cost> ext_row(const Cost& C, const LapDims& P, int i) {
  ExtRow<Cost> e;
  e.nc = P.nc;
  e.real = i < P.nr;
  if (e.real) { e.r = C.row(i); e.left = 0.0; e.right = P.half; }
  else { e.r = typename Cost::Row(); e.left = P.half; e.right = 0.0; }  // dummy rows have no box: nothing to fetch
  return e;
}
// same, with the row context already fetched (software prefetch one round ahead)
template <class Cost>
MOT_DEV ExtRow<Cost> ext_row_pf(const LapDims& P, int i, const typename Cost::Row& r) {
  ExtRow<Cost> e;
  e.nc = P.nc;
  e.real = i < P.nr;
  e.r = r;
  if (e.real) { e.left = 0.0; e.right = P.half; }
  else { e.left = P.half; e.right = 0.0; }
  return e;
}

// The calling lane's columns are j = t, t+T, ...; first_dummy() is the first of them in the dummy block (j >= nc).
MOT_DEV int first_dummy(int t, int T, int nc) { return t + ((nc > t) ? ((nc - t + T - 1) / T) * T : 0); }

// Visits the lane's REAL columns (j < nc) of extended row view R in ascending order and hands each reduced cost
// c = R(j) - v[j] to f(c, j).
template <class Cost, class VP, class F>
MOT_DEV void for_lane_real(const Cost& C, const ExtRow<Cost>& R, const VP& v, int t, int T, int nc, F f) {
  if (R.real) {
    if constexpr (Cost::kRPL > 0) {  // lane-owned column boxes in registers: fully unrolled, no box loads
#pragma unroll
      for (int k = 0; k < Cost::kRPL; ++k) {
        const int jj = t + k * T;
        if (jj < nc) f(C.at_owned(R.r, k, jj) - v[jj], jj);
      }
      for (int jj = t + Cost::kRPL * T; jj < nc; jj += T) f(C.at(R.r, jj) - v[jj], jj);  // columns beyond the register cache
    } else {
      for (int j = t; j < nc; j += T) f(C.at(R.r, j) - v[j], j);
    }
  } else { const double l = R.left; for (int j = t; j < nc; j += T) f(l - v[j], j); }
}
// Same for the lane's DUMMY columns (j >= nc), whose cost is the row constant `right`.
template <class VP, class F>
MOT_DEV void for_lane_dummy(double right, const VP& v, int t, int T, int nc, int n, F f) {
  for (int j = first_dummy(t, T, nc); j < n; j += T) f(right - v[j], j);
}
// Real block then dummy block: within a lane the order stays ascending, which the callers' tie rules rely on.
template <class Cost, class VP, class F>
MOT_DEV void for_lane_columns(const Cost& C, const ExtRow<Cost>& R, const VP& v, int t, int T, int nc, int n, F f) {
  for_lane_real(C, R, v, t, T, nc, f);
  for_lane_dummy(R.right, v, t, T, nc, n, f);
}

// Order-preserving map float -> int (signed compare == float compare for non-NaN values; +NaN sorts above +inf, -NaN below
// -inf) and back.
MOT_HD int f32_key(float x) {
  const int b = __builtin_bit_cast(int, x);
  return b ^ ((b >> 31) & 0x7fffffff);
}
MOT_HD float key_f32(int k) { return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7fffffff)); }

// Sparse column minima for the plain IoU-family functor (Cost::kPlain, every real column lane-owned). lapjv's column
// reduction wants, per column, the smallest cost and the lowest row attaining it. A row whose box does not intersect the
// column's costs zc = zero_cost_f(column) whatever the row, so only rows that can intersect need evaluating:
//   rows sorted by x1 (rank by counting, ties by index)      -> a0 < b2 holds exactly on a prefix [0, p_hi)
//   prefix maximum of x2 along that order                    -> a2 > b0 cannot hold before p_lo
//   candidates [p_lo, p_hi) evaluated with the full arithmetic; all other rows cost zc, the lowest of them is found by
//   walking i = 0, 1, ... until one is not a candidate (its rank is outside the range).
// The result is the lexicographic minimum of (cost, row) over all rows — what the ascending strict-< scan computes. A
// column box holding a NaN takes every row as candidate (a NaN column coordinate does not zero the intersection the way
// a NaN row coordinate does). Also leaves in W.rlb a lower bound of each row's minimum over the real columns (the
// evaluated pairs by atomic minimum, the rest by the smallest zc). Work arrays: five cold int arrays, free in phase 1.
constexpr int kSparseOwnRows = 16;  // rows per lane whose ranks are counted in registers
constexpr int kSparseMinRows = 32;
template <class G, class Cost, class Work>
MOT_DEV void sparse_column_minima(G& g, const Cost& C, const Work& W, int nr, int nc, float* vmk, int* imk, bool keep_rmin) {
  const int T = g.size(), t = g.tid();
  // The sorted tables live where the column duals and the column->row map will be written when the columns are published
  // (after this function's closing barrier) — LDS whenever the problem fits; v holds 2n ints, y holds n, n >= nr.
  using YT = decltype(W.y);  // (v and y share an address space)
  static_assert(sizeof(*W.v.raw()) == 2 * sizeof(int), "two ints per dual");
  const YT SIDX = W.y;                                                   // row at sorted position p
  const YT SKEY{reinterpret_cast<int*>(W.v.raw())};                      // key of its x1
  const YT PM{reinterpret_cast<int*>(W.v.raw()) + (nr + nc)};            // prefix maximum of the x2 keys along the sorted order
  const auto& RANK = W.inv;   // sorted position of row i
  const auto& RMK = W.tie;    // ~key of the row's smallest evaluated cost (atomic max)
  const int none = ~f32_key(3.0e38f);
  // ---- ranks: every lane counts, for its own rows (in registers), the rows that sort before them ----
  auto count_ranks = [&](auto slots_tag) {
    constexpr int kS = decltype(slots_tag)::value;
    int okey[kS], ornk[kS];
#pragma unroll
    for (int u = 0; u < kS; ++u) {
      const int i = t + u * T;
      okey[u] = (i < nr) ? f32_key(C.row_x1(i)) : 0;
      ornk[u] = 0;
    }
    for (int q = 0; q < nr; ++q) {
      const int kq = f32_key(C.row_x1(q));
#pragma unroll
      for (int u = 0; u < kS; ++u) {
        const int i = t + u * T;
        ornk[u] += ((kq < okey[u]) || (kq == okey[u] && q < i)) ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < kS; ++u) {
      const int i = t + u * T;
      if (i < nr) { SIDX[ornk[u]] = i; SKEY[ornk[u]] = okey[u]; RANK[i] = ornk[u]; RMK[i] = none; }
    }
  };
  if (nr <= 4 * T) count_ranks(std::integral_constant<int, 4>());
  else if (nr <= 8 * T) count_ranks(std::integral_constant<int, 8>());
  else count_ranks(std::integral_constant<int, kSparseOwnRows>());
  g.sync();
  // ---- prefix maximum of x2 in sorted order: a contiguous chunk per lane, one scan across the lanes ----
  {
    const int L = (nr + T - 1) / T;
    const int b = t * L, e = (b + L < nr) ? b + L : nr;
    int run = static_cast<int>(0x80000000u);
    for (int p = b; p < e; ++p) {
      const int k2 = f32_key(C.row_x2(static_cast<int>(SIDX[p])));
      if (k2 > run) run = k2;
      PM[p] = run;
    }
    const double ex = g.exclusive_scan_min(-static_cast<double>(run));
    if (ex < 1e299) {
      const int prev = static_cast<int>(-ex);
      for (int p = b; p < e; ++p)
        if (static_cast<int>(PM[p]) < prev) PM[p] = prev;
    }
  }
  g.sync();
  // ---- columns ----
  float zmin = 3.0e38f;
#pragma unroll
  for (int k = 0; k < Cost::kRPL; ++k) {
    const int jj = t + k * T;
    if (jj < nc) {
      const bool all_rows = C.owned_has_nan(k);
      int plo = 0, phi = nr;
      if (!all_rows) {
        const int kb2 = f32_key(C.owned_x2(k)), kb0 = f32_key(C.owned_x1(k));
        int lo = 0, hi = nr;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (static_cast<int>(SKEY[mid]) < kb2) lo = mid + 1; else hi = mid; }
        phi = lo;
        lo = 0; hi = phi;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (static_cast<int>(PM[mid]) > kb0) hi = mid; else lo = mid + 1; }
        plo = lo;
      }
      float vm = static_cast<float>(kLapLarge);
      int im = 0;
      for (int p = plo; p < phi; ++p) {
        const int i = SIDX[p];
        const float c = C.at_owned_f(C.row(i), k, jj);
        if (c < vm || (c == vm && i < im)) { vm = c; im = i; }
        G::atomic_max(RMK.raw(i), ~f32_key(c));
      }
      if (!all_rows) {
        int i0 = 0;
        while (i0 < nr) { const int r = RANK[i0]; if (r < plo || r >= phi) break; ++i0; }
        if (i0 < nr) {
          const float zc = C.zero_cost_f(k);
          if (zc < vm || (zc == vm && i0 < im)) { vm = zc; im = i0; }
          if (zc < zmin) zmin = zc;
        }
      }
      vmk[k] = vm; imk[k] = im;
    }
  }
  zmin = g.reduce_min_f32(zmin);
  g.sync();
  for (int i = t; i < nr; i += T) {
    const float rm = key_f32(~static_cast<int>(RMK[i]));
    const float raw = (zmin < rm) ? zmin : rm;
    W.rlb[i] = static_cast<double>(raw);
    if (keep_rmin) W.rmin[i] = raw;
  }
  // (the caller's barrier after publishing the columns orders these writes before any later use of the work arrays)
}

// Compacts {i in [0,n) : flag(i)} in ascending order into out[]; returns the count (uniform).
template <class G, class F, class Out>
MOT_DEV int compact_ascending(G& g, int n, F flag, const Out& out) {
  const int T = g.size(), t = g.tid();
  const int L = (n + T - 1) / T;
  const int b = t * L, e = (b + L < n) ? b + L : n;
  int c = 0;
  for (int i = b; i < e; ++i) c += flag(i) ? 1 : 0;
  int total;
  int pos = g.exclusive_scan(c, &total);
  for (int i = b; i < e; ++i)
    if (flag(i)) out[pos++] = i;
  g.sync();
  return total;
}

// Solves one problem. On return W.x[0..nr) / W.y[0..nc) hold extended assignments; callers map
// x >= nc / y >= nr to -1 (lap_solver.hpp:326-331). All threads of the group must call this.
template <class G, class Cost, class Work>
MOT_DEV void lap_solve(G& g, const Cost& C, const LapDims& P, const Work& W) {
  const int T = g.size(), t = g.tid();
  const int nr = P.nr, nc = P.nc, n = nr + nc;
  const double half = P.half;
  constexpr bool kLst16 = Work::kLst16;  // (then W.lst16 lies over the first (n + 1) / 2 ints of the fast scratch: n <= 2 * kFsEvl, which the LDS budget implies)

  long long c0 = MOT_CLOCK();
  long long n_carr = 0, n_paths = 0;
  long long n_fs_steps = 0, n_fs_members = 0, n_fs_sparse = 0, n_fs_events = 0, n_seq_sweeps = 0, n_fs_bad = 0, n_finds = 0;  // diagnostics
  long long cy_cls = 0, cy_dry = 0, cy_apply = 0, cy_evsort = 0, cy_evser = 0, cy_find = 0, cy_seq = 0, cy_init = 0;
  long long cy_sub[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (fine profile builds) classify: loads | reductions; relax: fetch issue | full barrier | evaluate | reductions; apply: A1 | A2 | A3
  // ---- phase 1: column reduction + reduction transfer (_ccrrt_dense, :36-72) ----
  // Row lists (matrix costs with the optional scratch): per real row the entries with cost < half, gathered by the column
  // sweep below. The shortest-path search uses them for exact sparse sweeps of real rows (see the scan of phase 3).
  const bool use_rl = is_matrix_cost<Cost>::value && W.rl_cnt.p != nullptr && W.fsw.p != nullptr && half > -1e300 && half < 1e300 &&
                      T * kFsIter >= kFsUnit && n <= kFsMaxN;
  for (int i = t; i < n; i += T) { W.x[i] = -1; W.fr[i] = 0; }
  if (use_rl) for (int i = t; i < nr; i += T) W.rl_cnt[i] = 0;
  g.sync();
  auto publish_column = [&](int j, double vm, int im) {
    W.v[j] = vm;
    W.y[j] = im;
    G::atomic_max(W.x.raw(im), j);   // x[i] = largest column whose minimum sits in row i (:47-55)
    G::atomic_add(W.fr.raw(im), 1);
  };
  int j_first = t;  // first column of this lane not handled by the register-cached sweep below
  // "Hopeless rows": lapjv's duals only ever decrease, so rlb[i] - (largest column dual after the column reduction) stays
  // a lower bound of every reduced cost of real row i over the real columns for the whole solve. A row whose bound lies
  // above the best dummy-column values can only ever pick dummy columns — such rows (every track with no detection near
  // it) are resolved in closed-form runs below, exactly like the dummy rows. Needs the row-major sweep of phase 1.
  const bool have_lb = Cost::kRPL > 0 && nc <= Cost::kRPL * T;
  // "Void real rows": the raw minimum of every real row over the real columns is kept (rmin[], on-the-fly costs with the row-major
  // phase 1). In the shortest-path search a real row whose every real cost is far enough above half cannot lower any real
  // column once a dummy row has been swept — see the search below; on tracking problems that is every unmatched track.
  const bool keep_rmin = have_lb && !is_matrix_cost<Cost>::value && half > -1e300 && half < 1e300;
  double vmax0 = 0.0;
  if constexpr (Cost::kRPL > 0) {
    // all of the lane's cached real columns advance together down the rows: one row-box fetch per row, no column loads
    // (in float: every cost IS a float and float -> double is exact and monotone, so minima and their rows are the same)
    float vmk[Cost::kRPL];
    int imk[Cost::kRPL];
#pragma unroll
    for (int k = 0; k < Cost::kRPL; ++k) { vmk[k] = static_cast<float>(kLapLarge); imk[k] = 0; }
    double neg_vmax = 1e300;
    bool swept = false;
    if constexpr (Cost::kPlain) {
      if (have_lb && nr >= kSparseMinRows && nr <= kSparseOwnRows * T) {
        sparse_column_minima(g, C, W, nr, nc, vmk, imk, keep_rmin);
        swept = true;
      }
    }
    constexpr int kRowBatch = 4;  // row contexts are fetched a batch at a time so that their load latencies overlap
    // ... and a batch ahead of their use: under load (a thousand problems in flight) a global load takes microseconds
    typename Cost::Row RN[kRowBatch];
    if (!swept) {
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) RN[u] = C.row((u < nr) ? u : nr - 1);
    }
    for (int i0 = swept ? nr : 0; i0 < nr; i0 += kRowBatch) {
      typename Cost::Row RB[kRowBatch];
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) RB[u] = RN[u];
      if (i0 + kRowBatch < nr) {
#pragma unroll
        for (int u = 0; u < kRowBatch; ++u) RN[u] = C.row((i0 + kRowBatch + u < nr) ? i0 + kRowBatch + u : nr - 1);
      }
      float rm[kRowBatch];  // this lane's share of each row's minimum
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) {
        const int i = i0 + u;
        rm[u] = 3.0e38f;
        if (i < nr) {
#pragma unroll
          for (int k = 0; k < Cost::kRPL; ++k) {
            const int jj = t + k * T;
            if (jj < nc) {
              const float c = C.at_owned_f(RB[u], k, jj);
              if (c < vmk[k]) { vmk[k] = c; imk[k] = i; }
              if (c < rm[u]) rm[u] = c;
            }
          }
        }
      }
      if (have_lb) {
#pragma unroll
        for (int u = 0; u < kRowBatch; ++u) rm[u] = g.reduce_min_f32(rm[u]);  // independent chains: they interleave
#pragma unroll
        for (int u = 0; u < kRowBatch; ++u)
          if (t == 0 && i0 + u < nr) {
            W.rlb[i0 + u] = static_cast<double>(rm[u]);
            if (keep_rmin) W.rmin[i0 + u] = rm[u];
          }
      }
    }
#pragma unroll
    for (int k = 0; k < Cost::kRPL; ++k) {
      const int jj = t + k * T;
      if (jj < nc) {
        double vm = static_cast<double>(vmk[k]);
        int im = imk[k];
        if (!(vmk[k] < static_cast<float>(kLapLarge))) { vm = kLapLarge; im = 0; }  // nothing below LARGE (:41-46)
        if (half < vm) { vm = half; im = nr; }  // rows nr.. are all `half`: only the first can win
        publish_column(jj, vm, im);
        if (-vm < neg_vmax) neg_vmax = -vm;  // (negated: the group primitive is a minimum)
      }
    }
    j_first = t + Cost::kRPL * T;
    if (have_lb) {
      vmax0 = -g.reduce_min(neg_vmax);
      g.sync();
      for (int i = t; i < nr; i += T) W.rlb[i] -= vmax0;  // raw row minimum -> lower bound of the row's reduced costs
    }
  }
  for (int j = j_first; j < nc; j += T) {  // real columns outside the register cache (or all of them without one)
    double vm = kLapLarge;
    int im = 0;
    for (int i = 0; i < nr; ++i) {
      const double c = C.at(i, j);
      if (c < vm) { vm = c; im = i; }
      if constexpr (is_matrix_cost<Cost>::value) {
        if (use_rl && c < half) {
          const int slot = G::atomic_add(W.rl_cnt.raw(i), 1);
          if (slot < kRlCap) W.rl_ent[static_cast<long>(i) * kRlCap + slot] = rl_pack(j, static_cast<float>(c));
        }
      }
    }
    if (half < vm) { vm = half; im = nr; }
    publish_column(j, vm, im);
  }
  for (int j = first_dummy(t, T, nc); j < n; j += T) {  // dummy columns
    double vm = kLapLarge;
    int im = 0;
    if (half < vm) { vm = half; im = 0; }   // rows 0..nr-1 are all `half`
    if (0.0 < vm) { vm = 0.0; im = nr; }    // rows nr.. are all 0
    publish_column(j, vm, im);
  }
  g.sync();
  long long c1 = MOT_CLOCK();
  const bool have_rn = use_rl && W.rl_n.p != nullptr;   // list lengths as bytes next to the solver state (LDS)
  const bool have_yc = use_rl && W.ycost.p != nullptr;  // the cost of every real column's matched pair, kept current from phase 3 on
  if (have_rn) for (int i = t; i < nr; i += T) { const int c = W.rl_cnt[i]; W.rl_n[i] = static_cast<unsigned char>(c > 255 ? 255 : c); }
  auto rl_len = [&](int i) -> int { return have_rn ? static_cast<int>(W.rl_n[i]) : static_cast<int>(W.rl_cnt[i]); };
  for (int j = t; j < n; j += T)
    if (W.x[W.y[j]] != j) W.y[j] = -1;
  g.sync();
  // rows that own exactly one column get their dual tightened, in ascending row order (:57-69)
  const int n_uniq = compact_ascending(g, n, [&](int i) { return W.x[i] >= 0 && W.fr[i] == 1; }, W.cols);
  int nfree = compact_ascending(g, n, [&](int i) { return W.x[i] < 0; }, W.fr);
  // Per-lane cache over the lane's DUMMY columns as seen from a real row (cost `half` everywhere): the two
  // lexicographically smallest (half - v[j], j) and the largest column attaining the minimum. It does not depend on the
  // row, so it stays valid until the dual of some dummy column is written (rare: the dummy block is one big tie).
  bool dq_ok = false;   // uniform
  Top2 dq = top2_empty();
  Top2 dqg = top2_empty();  // the same over the whole group (only kept when row lower bounds exist)
  int dq_ptr = -1;
  auto ensure_dq = [&]() {
    if (dq_ok) return;
    dq = top2_empty();
    dq_ptr = -1;
    for_lane_dummy(half, W.v, t, T, nc, n, [&](double c, int j) {
      if (c < dq.v2) {
        if (c < dq.v1) { dq.v2 = dq.v1; dq.j2 = dq.j1; dq.v1 = c; dq.j1 = j; dq_ptr = j; }
        else { dq.v2 = c; dq.j2 = j; }
      }
      if (c == dq.v1) dq_ptr = j;
    });
    if (have_lb) dqg = g.reduce_top2(dq);
    dq_ok = true;
  };
  // real row r cannot prefer any real column to the dummy columns whose value is <= bound (call with dq current)
  auto hopeless = [&](int r, double bound) { return r >= 0 && r < nr && static_cast<double>(W.rlb[r]) > bound; };
  // three-deep software pipeline over the (fixed) list of rows: the index two rounds ahead, its column and row box one
  // round ahead — each a dependent global load whose latency then overlaps a whole round (x[] and cols[] do not change here)
  int pf_i = (n_uniq > 0) ? static_cast<int>(W.cols[0]) : 0;
  int pf_j = (n_uniq > 0) ? static_cast<int>(W.x[pf_i]) : 0;
  typename Cost::Row pf_box = (n_uniq > 0 && pf_i < nr) ? C.row(pf_i) : typename Cost::Row();
  int pf_i2 = (n_uniq > 1) ? static_cast<int>(W.cols[1]) : 0;
  for (int u = 0; u < n_uniq; ++u) {
    const int i = pf_i;
    const int j = pf_j;
    const ExtRow<Cost> R = ext_row_pf<Cost>(P, i, pf_box);
    if (u + 1 < n_uniq) {
      pf_i = pf_i2;
      pf_j = W.x[pf_i];
      pf_box = (pf_i < nr) ? C.row(pf_i) : typename Cost::Row();
      if (u + 2 < n_uniq) pf_i2 = W.cols[u + 2];
    }
    double mn = kLapLarge;
    if (R.real) {
      ensure_dq();
      for_lane_real(C, R, W.v, t, T, nc, [&](double c, int j2) { if (j2 != j && c < mn) mn = c; });
      const double dm = (dq.j1 == j) ? dq.v2 : dq.v1;  // minimum over the lane's dummy columns other than j
      if (dm < mn) mn = dm;
    } else {
      for_lane_columns(C, R, W.v, t, T, nc, n, [&](double c, int j2) { if (j2 != j && c < mn) mn = c; });
    }
    mn = g.reduce_min(mn);
    if ((j % T) == t) W.v[j] -= mn;  // owner lane: next reader of v[j] is this same lane
    if (j >= nc) dq_ok = false;
  }
  g.sync();

  long long c2 = MOT_CLOCK();
  // ---- phase 2: augmenting row reduction, twice (_carr_dense, :74-113, :221-224) ----
  for (int pass = 0; pass < 2 && nfree > 0; ++pass) {
    unsigned current = 0, rr_cnt = 0;
    int new_free = 0;
    int forwarded = -1;  // row re-queued by free_rows[--current] = i0, consumed next iteration
    bool dc_valid = false;
    Top2 dc = top2_empty();
    // software prefetch: the next unread free-list entry and its row context are fetched one round ahead (entries at or
    // beyond the read position are never rewritten during a pass, so the prefetched value cannot go stale)
    int nxt = W.fr[0];
    typename Cost::Row nxt_row = (nxt < nr) ? C.row(nxt) : typename Cost::Row();
    double nxt_lb = (have_lb && nxt < nr) ? static_cast<double>(W.rlb[nxt]) : 0.0;  // (an older bound is still a bound)
    while (current < static_cast<unsigned>(nfree)) {
      // ---- runs of rows with a fixed, exactly tied pair of best columns, in closed form ----
      // A dummy row's two best columns are the cached tuple dc; a hopeless real row's are the two best dummy columns
      // dqg. When the pair ties exactly (v2 == v1: always, until some row pulls a dummy column's dual down) the round
      // changes no dual: the row takes j1 if that column is free, else j2, whose previous owner goes back to the free
      // list. A run of K consecutive such rows of the free list is therefore a chain — each row takes j2 and displaces
      // its predecessor — whose outcome is written here in one step instead of K serial rounds. (The extension turns
      // every unmatched detection into a dummy row and every unmatched track into a hopeless one.)
      bool use_d = false, use_h = false;
      int cj1 = -1, cj2 = -1;
      if (forwarded < 0 && (rr_cnt + 1u) < (current + 1u) * static_cast<unsigned>(n)) {
        const bool d_tie = dc_valid && dc.v1 == dc.v2 && dc.v2 < kLapLarge && dc.j2 != kNoIdx;
        if (have_lb) ensure_dq();
        const bool h_tie = have_lb && dqg.v1 == dqg.v2 && dqg.v2 < kLapLarge && dqg.j2 != kNoIdx;
        if (nxt >= nr) {
          if (d_tie) { use_d = true; cj1 = dc.j1; cj2 = dc.j2; use_h = h_tie && dqg.j1 == cj1 && dqg.j2 == cj2; }
        } else if (h_tie && nxt_lb > dqg.v2) {
          use_h = true; cj1 = dqg.j1; cj2 = dqg.j2;
          use_d = d_tie && dc.j1 == cj1 && dc.j2 == cj2;
        }
      }
      if (use_d || use_h) {
        const long long qr0 = MOT_FCLOCK();
        const double hb = dqg.v2;
        g.sync();  // the previous round's owner writes to y[] are visible
        const int a0 = W.y[cj1], b0 = W.y[cj2];
        const int s = (a0 < 0) ? 1 : 0;   // rows of the run that take j1 (at most the first)
        const int bi = (b0 >= 0) ? 1 : 0; // j2's owner before the run is displaced first
        const int nf0 = new_free;
        int k0 = 0, prev_last = -1, last_row = -1;
        bool open = true;
        while (open && current < static_cast<unsigned>(nfree)) {
          const unsigned idx = current + static_cast<unsigned>(t);
          const int r = (idx < static_cast<unsigned>(nfree)) ? static_cast<int>(W.fr[idx]) : -1;
          const bool ok = (use_d && r >= nr) || (use_h && hopeless(r, hb));
          int cnt = g.reduce_min_int(ok ? kNoIdx : t);  // leading lanes whose entry belongs to the run
          if (cnt > T) cnt = T;
          if (cnt == 0) break;
          const int rprev_lane = (t > 0 && t < cnt) ? static_cast<int>(W.fr[idx - 1]) : prev_last;
          const int chunk_last = g.reduce_max((t == cnt - 1) ? r : -1);
          g.sync();  // every lane has read its entries before the list is rewritten below
          if (t < cnt) {
            const int k = k0 + t;
            if (k < s) { W.x[r] = cj1; W.y[cj1] = r; }
            else {
              W.x[r] = cj2;
              if (k == s) { if (bi) W.fr[nf0] = b0; }
              else W.fr[nf0 + bi + (k - s - 1)] = rprev_lane;
            }
          }
          k0 += cnt;
          current += static_cast<unsigned>(cnt);
          prev_last = chunk_last;
          last_row = chunk_last;
          open = cnt == T;
        }
        if (k0 > 0) {
          if (k0 > s) {
            if (t == 0) W.y[cj2] = last_row;
            new_free = nf0 + bi + (k0 - s - 1);
          }
          rr_cnt += static_cast<unsigned>(k0);
          g.sync();
          if (current < static_cast<unsigned>(nfree)) {
            nxt = W.fr[current];
            nxt_row = (nxt < nr) ? C.row(nxt) : typename Cost::Row();
            nxt_lb = (have_lb && nxt < nr) ? static_cast<double>(W.rlb[nxt]) : 0.0;
          }
          if (!use_rl) { cy_sub[4] += MOT_FCLOCK() - qr0; cy_sub[8] += 1; }  // (fine-profile builds; the scan steps of phase 3 own these slots when row lists exist)
          continue;
        }
      }
      ++rr_cnt;
      ++n_carr;
      const long long qr1 = MOT_FCLOCK();
      long long qr2 = qr1, qr3 = qr1;
      const bool from_list = forwarded < 0;
      const int fi = from_list ? nxt : forwarded;
      typename Cost::Row fi_row = nxt_row;
      if (!from_list && fi < nr) fi_row = C.row(fi);
      forwarded = -1;
      ++current;
      if (from_list && current < static_cast<unsigned>(nfree)) {
        nxt = W.fr[current];
        nxt_row = (nxt < nr) ? C.row(nxt) : typename Cost::Row();
        nxt_lb = (have_lb && nxt < nr) ? static_cast<double>(W.rlb[nxt]) : 0.0;
      }
      // Dummy rows (fi >= nr) all have the same cost row, so while no dual changes their top-2 is the same tuple:
      // it is reduced once and reused until some v[j] is written (the extension makes these rounds a large share).
      const bool dummy_row = fi >= nr;
      Top2 tt;
      if (dummy_row && dc_valid) {
        tt = dc;
        g.sync();  // stands in for the reduction's barrier: last round's owner writes are visible before y/v are read
      } else {
        const ExtRow<Cost> R = ext_row_pf<Cost>(P, fi, fi_row);
        tt = top2_empty();
        auto push = [&](double c, int j) {
          if (c < tt.v2) {  // a lane visits its columns in ascending order: strict < keeps the lowest index on ties
            if (c < tt.v1) { tt.v2 = tt.v1; tt.j2 = tt.j1; tt.v1 = c; tt.j1 = j; }
            else { tt.v2 = c; tt.j2 = j; }
          }
        };
        if (!dummy_row) {
          ensure_dq();
          for_lane_real(C, R, W.v, t, T, nc, push);
          qr2 = MOT_FCLOCK();
          qr3 = qr2;
          if (have_lb) {
            // (round 5) the dummy columns' two best are known to the whole group (dqg): only real entries below them can matter — one
            // collective; the row's minimum over the real columns comes back as a float bound (a tighter rlb from here on: rlb is only ever a bound)
            float rreal;
            tt = g.reduce_top2_under(tt, dqg, &rreal);
            if (t == 0) W.rlb[fi] = static_cast<double>(rreal);
          } else {
            push(dq.v1, dq.j1);  // the lane's dummy columns come after its real ones, best first
            push(dq.v2, dq.j2);
            tt = g.reduce_top2(tt);
          }
        } else {
          for_lane_columns(C, R, W.v, t, T, nc, n, push);
          tt = g.reduce_top2(tt);
        }
        if (dummy_row) { dc = tt; dc_valid = true; }
      }
      const long long qr4 = MOT_FCLOCK();
      int j1 = tt.j1, j2 = tt.j2;
      double v1 = tt.v1, v2 = tt.v2;
      if (!(v2 < kLapLarge)) { v2 = kLapLarge; j2 = -1; }
      int i0 = W.y[j1];
      const double vj1 = W.v[j1];
      const int yj2 = (j2 >= 0) ? W.y[j2] : -1;
      g.sync();  // every lane has read y[j1], v[j1], y[j2] before any of them is rewritten
      const double v1_new = vj1 - (v2 - v1);
      const bool lowers = v1_new < vj1;
      if (rr_cnt < current * static_cast<unsigned>(n)) {
        if (lowers) { if ((j1 % T) == t) W.v[j1] = v1_new; dc_valid = false; if (j1 >= nc) dq_ok = false; }
        else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = yj2; }
        if (i0 >= 0) {
          if (lowers) { --current; if (t == 0) W.fr[current] = i0; forwarded = i0; }
          else { if (t == 0) W.fr[new_free] = i0; ++new_free; }
        }
      } else if (i0 >= 0) {
        if (t == 0) W.fr[new_free] = i0;
        ++new_free;
      }
      if ((j1 % T) == t) { W.x[fi] = j1; W.y[j1] = fi; }
      if (!use_rl) {  // (fine-profile builds)
        const long long qr5 = MOT_FCLOCK();
        if (dummy_row) { cy_sub[5] += qr5 - qr1; cy_sub[7] += 1; }
        else { cy_sub[0] += qr2 - qr1; cy_sub[1] += qr3 - qr2; cy_sub[2] += qr4 - qr3; cy_sub[3] += qr5 - qr4; cy_sub[6] += 1; }
      }
      // no barrier here: v[j1]/y[j1] are next read (a) by their owner lane in the strided
      // loop above, or (b) by everyone only after the reduce_top2 barrier of the next round.
    }
    g.sync();
    nfree = new_free;
  }

  if (have_yc) {  // (uniform) from here on y[] changes in three places only, each of which keeps this current
    for (int j = t; j < nc; j += T) {
      const int i = W.y[j];
      if (i >= 0 && i < nr) W.ycost[j] = static_cast<float>(C.at(i, j));
    }
    g.sync();
  }
  // cost of real column j's matched pair (i = y[j], a real row)
  auto match_cost = [&](int i, int j) -> double { return have_yc ? static_cast<double>(static_cast<float>(W.ycost[j])) : C.at(i, j); };
  long long c3 = MOT_CLOCK();
  // ---- phase 3: augmentation (_ca_dense, :195-211) ----
  bool da_valid = false;  // dummy-start cache: valid while no dual has changed (single-step paths never change duals)
  double da_g0 = 0.0;
  int da_ptr = -1;        // this lane's largest column that may still be a free member of the tied-minimum set
  int pf_start = (nfree > 0) ? W.fr[0] : 0;
  typename Cost::Row pf_row = (nfree > 0 && pf_start < nr) ? C.row(pf_start) : typename Cost::Row();
  for (int f = 0; f < nfree; ++f) {
    // ---- run of hopeless real starts in closed form ----
    // Such a start's minimum is the best dummy-column value, its tied set the free dummy columns attaining it, and it
    // changes no dual: a run of K of them takes the K largest members in descending order.
    if (have_lb && pf_start < nr && f + 1 < nfree) {
      ensure_dq();
      const double hb = dqg.v1;
      if (hb < kLapLarge && hopeless(pf_start, hb) && hopeless(static_cast<int>(W.fr[f + 1]), hb)) {
        g.sync();
        const int nd = n - nc;
        const int nl = compact_ascending(g, nd, [&](int q) {
          const int j = n - 1 - q;
          return (half - W.v[j]) == hb && W.y[j] < 0;
        }, W.lst);
        int m = 0;
        bool open = nl > 0;
        while (open) {
          const int idx = f + m + t;
          const int r = (idx < nfree && m + t < nl) ? static_cast<int>(W.fr[idx]) : -1;
          int cnt = g.reduce_min_int(hopeless(r, hb) ? kNoIdx : t);
          if (cnt > T) cnt = T;
          if (t < cnt) {
            const int j = n - 1 - static_cast<int>(W.lst[m + t]);
            W.y[j] = r;
            W.x[r] = j;
          }
          m += cnt;
          open = cnt == T;
        }
        g.sync();
        if (m > 0) {
          f += m;
          if (f < nfree) {
            pf_start = W.fr[f];
            pf_row = (pf_start < nr) ? C.row(pf_start) : typename Cost::Row();
          }
          --f;  // the loop increment
          continue;
        }
      }
    }
    // ---- run of dummy starts in closed form ----
    // While the dummy-start cache is valid every dummy start takes the largest free member of the same tied set and
    // changes no dual, so a run of K consecutive dummy starts takes the K largest members in descending order.
    if (da_valid && pf_start >= nr && f + 1 < nfree && static_cast<int>(W.fr[f + 1]) >= nr) {
      g.sync();
      const int nl = compact_ascending(g, n, [&](int q) {
        const int j = n - 1 - q;
        return (((j < nc) ? half : 0.0) - W.v[j]) == da_g0 && W.y[j] < 0;
      }, W.lst);  // lst[k] = n-1 - (k-th largest free tied column)
      int m = 0;
      bool open = nl > 0;
      while (open) {
        const int idx = f + m + t;
        const int r = (idx < nfree && m + t < nl) ? static_cast<int>(W.fr[idx]) : -1;
        int cnt = g.reduce_min_int((r >= nr) ? kNoIdx : t);
        if (cnt > T) cnt = T;
        if (t < cnt) {
          const int j = n - 1 - static_cast<int>(W.lst[m + t]);
          W.y[j] = r;
          W.x[r] = j;
        }
        m += cnt;
        open = cnt == T;
      }
      g.sync();
      if (m > 0) {
        f += m;
        if (f < nfree) {
          pf_start = W.fr[f];
          pf_row = (pf_start < nr) ? C.row(pf_start) : typename Cost::Row();
        }
        --f;  // the loop increment
        continue;
      }
    }
    const int start = pf_start;
    ++n_paths;
    const ExtRow<Cost> R0 = ext_row_pf<Cost>(P, start, pf_row);
    if (f + 1 < nfree) {  // next start and its row context, one round ahead (the free list is not modified in this phase)
      pf_start = W.fr[f + 1];
      pf_row = (pf_start < nr) ? C.row(pf_start) : typename Cost::Row();
    }
    const bool dummy_row = start >= nr;
    int final_j;
    if (dummy_row && da_valid) {
      // Same cost row and same duals as the last dummy start: the tied-minimum set is unchanged and only shrinks as
      // columns get assigned, so each lane just walks its own candidate pointer down to its next free tied column.
      while (da_ptr >= 0 && !((((da_ptr < nc) ? R0.left : R0.right) - W.v[da_ptr]) == da_g0 && W.y[da_ptr] < 0)) da_ptr -= T;
      if (da_ptr < 0) da_ptr = -1;
      final_j = g.reduce_max(da_ptr);
    } else {
      // One sweep: the lane's minimum and the LAST free column attaining it. After the group minimum g0 is known only
      // lanes whose own minimum equals g0 can hold a member of the tied set. (First _find_dense from cols = identity
      // leaves the tied-minimum columns in ascending order in [0,hi) and the sink test (:174-177) keeps the LAST free one.)
      double mn = 1e300;
      int cand = -1;
      auto visit = [&](double dj, int j) {
        if (dj <= mn) {
          const bool is_free = W.y[j] < 0;
          if (dj < mn) { mn = dj; cand = is_free ? j : -1; }
          else if (is_free) cand = j;
        }
      };
      if (!dummy_row) {
        ensure_dq();
        for_lane_real(C, R0, W.v, t, T, nc, visit);
        const double lane_mn = (dq.v1 < mn) ? dq.v1 : mn;
        const double g0 = g.reduce_min(lane_mn);
        if (!(mn == g0)) cand = -1;
        if (dq.v1 == g0) {
          // tied dummy columns: same pointer walk as the dummy-start cache below — the tied set among the lane's dummy
          // columns is the same for every real start while no dummy dual changes, and only shrinks as columns get assigned
          while (dq_ptr >= nc && !((half - W.v[dq_ptr]) == g0 && W.y[dq_ptr] < 0)) dq_ptr -= T;
          if (dq_ptr >= nc) cand = dq_ptr;  // dummy columns come after the lane's real ones
        }
        final_j = g.reduce_max(cand);
      } else {
        for_lane_columns(C, R0, W.v, t, T, nc, n, visit);
        const double g0 = g.reduce_min(mn);
        if (!(mn == g0)) cand = -1;
        final_j = g.reduce_max(cand);
        da_valid = true; da_g0 = g0;
        da_ptr = cand;  // this lane's largest column that may still be a free member of the tied set
      }
    }
    if (final_j < 0) {
      // ---- general path: exact emulation of find_path_dense (:157-193) ----
      da_valid = false;  // the dual update below changes v
      dq_ok = false;
      const long long qi0 = MOT_FCLOCK();
      for (int j = t; j < n; j += T) { W.cols[j] = j; W.inv[j] = j; W.tie[j] = 0; W.pred[j] = start; W.d[j] = R0.at(C, j) - W.v[j]; }
      if (use_rl) {  // the steps' column table starts empty; TODO bitmask: one bit per column that has not entered the SCAN set yet
        for (int w = t; w < kFsHash; w += T) { W.fsw[kFsKeep + w] = -1; W.fsw[kFsKeep + kFsHash + w] = kNoIdx; }
        for (int w = t; w < (n + 31) / 32; w += T) W.fsw[kFsTodo + w] = (32 * (w + 1) <= n) ? -1 : static_cast<int>((1u << (n & 31)) - 1u);
      }
      g.sync();
      cy_init += MOT_FCLOCK() - qi0;
      unsigned lo = 0, hi = 0, n_ready = 0;
      // largest h of a fully swept dummy row / of a real row's dummy-column part so far in this search (see the scan)
      // (the initial distances d[j] = E(start, j) - v[j] are a sweep of the start row with h = 0)
      double hmax_dummy_row = dummy_row ? 0.0 : -1e300, hmax_real_row = dummy_row ? -1e300 : 0.0;
      while (final_j == -1) {
        if (lo == hi) {
          n_ready = lo;
          ++n_finds;
          const long long qf0 = MOT_FCLOCK();
          // With cols[] / inv[] / d[] / the record list in LDS nothing of a _find_dense lives in global memory (the memory-staged tie runs
          // apart, which keep their full barriers): its barriers and the collectives' order LDS only — a full barrier also waits for the
          // stores to pred[] the scan steps left in flight — and the insertion point travels through an LDS word instead of tmp[0].
          const bool find_lds = kLst16 && std::remove_reference_t<decltype(W)>::kColsSpace == kMemLds && use_rl;
          auto fsync = [&]() { if (find_lds) g.sync_lds(); else g.sync(); };
          auto put_h2 = [&](int v) { if (find_lds) W.fsw[kFsTodo + kFsMaxN / 32] = v; else W.tmp[0] = v; };
          auto get_h2 = [&]() -> int { return find_lds ? static_cast<int>(W.fsw[kFsTodo + kFsMaxN / 32]) : static_cast<int>(W.tmp[0]); };
          g.lds_barriers(find_lds);
          // _find_dense (:115-127). Its outcome depends on the ORDER of cols[], which only changes at positions whose
          // value is <= the running minimum ("weak records") — and the values it reads are those of the cols[] order
          // at entry (a swap never touches a position still to be read). So: find the records in parallel (chunk
          // minima -> exclusive prefix-min -> re-scan), compact them in order, and let one lane replay only those.
          {
            const int first = static_cast<int>(lo) + 1, cnt = n - first;
            const double m0 = W.d[W.cols[lo]];
            // the list of records: 16-bit entries in LDS for the wide matrix problems (round 5: over the fast scratch's member / unit / column tables,
            // which are dead between SCAN sets — the column table is emptied again below), the task's global scratch otherwise
            auto lst_sel = [&]() -> decltype(auto) { if constexpr (kLst16) return (W.lst16); else return (W.lst); };
            const auto& LST = lst_sel();
            const int L = (cnt + T - 1) / T;
            const int b = first + t * L, e = (b + L < n) ? b + L : n;
            // (round 3) A lane's chunk of cols[] is read ONCE into registers; the records, their compaction into lst[] and the last
            // record that lowers the minimum all come out of those registers — the flag array and its three passes over memory
            // are only the fallback for chunks longer than kFindChunk.
            constexpr int kFindChunk = 16;
            int nrec = 0, last_strict_reg = -2;  // (-2: not computed here)
            if (L <= kFindChunk) {
              int cj[kFindChunk];
              double dv[kFindChunk];
#pragma unroll
              for (int u = 0; u < kFindChunk; ++u) cj[u] = (b + u < e) ? static_cast<int>(W.cols[b + u]) : 0;
              double cm = 1e300;
#pragma unroll
              for (int u = 0; u < kFindChunk; ++u) {
                dv[u] = (b + u < e) ? static_cast<double>(W.d[cj[u]]) : 1e300;
                if (dv[u] < cm) cm = dv[u];
              }
              double run = g.exclusive_scan_min(cm);
              if (m0 < run) run = m0;
              unsigned recm = 0u;
              int nl = 0, strict_u = -1;
#pragma unroll
              for (int u = 0; u < kFindChunk; ++u)
                if (b + u < e && dv[u] <= run) {
                  if (dv[u] < run) { run = dv[u]; strict_u = nl; }  // a strict record: the running minimum falls
                  recm |= 1u << u;
                  ++nl;
                }
              int base = g.exclusive_scan(nl, &nrec);
              last_strict_reg = g.reduce_max((strict_u >= 0) ? base + strict_u : -1);
#pragma unroll
              for (int u = 0; u < kFindChunk; ++u)
                if (recm & (1u << u)) LST[base++] = (b + u) - first;
              fsync();
            } else {
              double cm = 1e300;
              for (int k = b; k < e; ++k) { const double dj = W.d[W.cols[k]]; if (dj < cm) cm = dj; }
              double run = g.exclusive_scan_min(cm);
              if (m0 < run) run = m0;
              for (int k = b; k < e; ++k) {
                const double dj = W.d[W.cols[k]];
                int rec = 0;
                if (dj <= run) { rec = 1; if (dj < run) run = dj; }
                W.tmp[k] = rec;
              }
              g.sync();
              nrec = compact_ascending(g, cnt, [&](int q) { return W.tmp[first + q] != 0; }, LST);
            }
            // The replay is serial in the number of records, and on tracking problems most of them are TIES with the final
            // minimum (the dummy block is one big tie: thousands of records per call). After the last record that lowers
            // the minimum (a "strict" record: it restarts the insertion point at lo and lands there itself) the insertion
            // point is lo + 1 and every remaining record r = 0..R-1 is one swap of positions s + r and k_r (s = lo + 1,
            // k_r ascending, k_r >= s + r). Their net effect has a closed form, applied by the whole group:
            //   * position s + r ends up holding record r's column;
            //   * an item displaced from s + r0 hops to k_r0, and on from there if k_r0 = s + r1 is itself displaced at
            //     step r1 > r0, until it lands on a k beyond the last target position: with g(r) = k_r - s that is the
            //     fixed point of r -> g(r) (g is increasing and injective), found by pointer jumping over the records.
            // Only items whose position s + r0 is not some earlier record's k start such a chain ("fresh").
            const long long qf1 = MOT_FCLOCK();
            cy_sub[9] += qf1 - qf0;
            constexpr int kTiePer = MOT_LAP_TIE_PER;  // tie records per lane held in registers across the read/write barrier
            constexpr int kTieMin = MOT_LAP_TIE_MIN;  // shorter tie runs stay with the serial replay
            static_assert(kTiePer <= 32, "one bit per held record in `fresh`");
            int nser = nrec, R = 0;        // serially replayed records, then R tie records in closed form
            if (nrec >= kTieMin) {
              int last_strict = -1;
              if (last_strict_reg > -2) last_strict = last_strict_reg;
              else
              for (int r = t; r < nrec; r += T) {
                const double dr = W.d[W.cols[first + static_cast<int>(LST[r])]];
                const double dp = r ? static_cast<double>(W.d[W.cols[first + static_cast<int>(LST[r - 1])]]) : m0;
                if (dr < dp) last_strict = r;  // (records are weak: the running minimum before r is record r-1's value)
              }
              if (last_strict_reg == -2) last_strict = g.reduce_max(last_strict);
              const int ties = nrec - 1 - last_strict;
              if (ties >= kTieMin) { nser = last_strict + 1; R = ties; }  // (any length: longer runs are staged through memory)
            }
#if defined(__HIP_DEVICE_COMPILE__)
            {
              // The first wavefront replays out of registers (the others wait at the barrier below). A chunk of 64 records
              // (position, column, distance) is gathered in parallel, one per lane, together with a 64-entry window of cols[] at the insertion point; the
              // serial walk then reads them with v_readlane and keeps the windows current with lane-selects, so that an
              // iteration costs ~30 instructions instead of three dependent global loads. (A later record's position is
              // never touched by an earlier step: h2 <= lo + r < k_r.) The stores to cols[] / inv[] still go to memory.
              if (t < 64) {
                unsigned h2 = lo + 1;
                double mind = m0;
                // Two windows of cols[]: win_lo at lo (where a new minimum restarts the insertion point) and win_ch at the
                // insertion point the chunk started with. Within a chunk h2 takes at most 64 values from there, or after a
                // restart at most 64 from lo, so one of them always covers it and the walk has no loads — a load in the loop body would make
                // every step wait for the previous step's stores (stores count in vmcnt on gfx9).
                const bool in_lo = lo + static_cast<unsigned>(t) < static_cast<unsigned>(n);
                int win_lo = in_lo ? static_cast<int>(W.cols[lo + t]) : 0;
                for (int r0 = 0; r0 < nser; r0 += 64) {
                  const int r = r0 + t;
                  int rk = 0, rj = 0;
                  double rd = 0.0;
                  if (r < nser) { rk = first + static_cast<int>(LST[r]); rj = W.cols[rk]; rd = W.d[rj]; }
                  const unsigned wb = h2;  // (the first chunk starts at lo + 1: its 64th record may insert at lo + 64)
                  const int win_ch0 = (wb + static_cast<unsigned>(t) < static_cast<unsigned>(n)) ? static_cast<int>(W.cols[wb + t]) : 0;
                  int win_ch = win_ch0;
                  const int cc = (nser - r0 < 64) ? nser - r0 : 64;
                  for (int q = 0; q < cc; ++q) {
                    const int k = __builtin_amdgcn_readlane(rk, q), j = __builtin_amdgcn_readlane(rj, q);
                    const double dj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rd), q), __builtin_amdgcn_readlane(__double2loint(rd), q));
                    if (dj < mind) { h2 = lo; mind = dj; }
                    const unsigned off_ch = h2 - wb, off_lo = h2 - lo;
                    const int jh_ch = __builtin_amdgcn_readlane(win_ch, static_cast<int>(off_ch & 63u));
                    const int jh_lo = __builtin_amdgcn_readlane(win_lo, static_cast<int>(off_lo & 63u));
                    const int jh = (off_ch < 64u) ? jh_ch : jh_lo;
                    if (t == 0) {
                      W.cols[k] = jh; W.inv[jh] = k;
                      W.cols[h2] = j; W.inv[j] = static_cast<int>(h2);
                    }
                    const unsigned ut = static_cast<unsigned>(t), uk = static_cast<unsigned>(k);
                    if (ut == uk - wb) win_ch = jh;  // (offsets >= 64 match no lane)
                    if (ut == off_ch) win_ch = j;
                    if (ut == uk - lo) win_lo = jh;
                    if (ut == off_lo) win_lo = j;
                    ++h2;
                  }
                }
                if (t == 0) put_h2(static_cast<int>(h2));  // (tmp[0] is not a record flag: those start at first >= 1)
              }
            }
#else
            if (t == 0) {  // groups without wavefronts (tests/emu): lane 0 replays
              unsigned h2 = lo + 1;
              double mind = m0;
              for (int r = 0; r < nser; ++r) {
                const int k = first + LST[r];
                const int j = W.cols[k];
                const double dj = W.d[j];
                if (dj < mind) { h2 = lo; mind = dj; }
                const int jh = W.cols[h2];
                W.cols[k] = jh; W.inv[jh] = k;
                W.cols[h2] = j; W.inv[j] = static_cast<int>(h2);
                ++h2;
              }
              put_h2(static_cast<int>(h2));
            }
#endif
            fsync();  // the replaying lane's stores to cols[] / inv[] / tmp[0] are visible to everyone
            const long long qf2 = MOT_FCLOCK();
            cy_sub[10] += qf2 - qf1;
            unsigned h2;
            if (R == 0) {
              h2 = static_cast<unsigned>(get_h2());
            } else if (R > kTiePer * T) {
              // the same closed form with the per-record values staged through memory instead of registers (any run length)
              g.lds_barriers(false);  // (its scratch is global memory: full barriers, also inside the collectives)
              const int a = nser, s = first;
              for (int r = t; r < R; r += T) W.tmp[r] = 0;
              g.sync();
              for (int r = t; r < R; r += T) { const int gq = LST[a + r]; if (gq < R && gq != r) W.tmp[gq] = 1; }
              g.sync();
              for (int r = t; r < R; r += T) { const int gq = LST[a + r]; W.sc[r] = (gq != r && static_cast<int>(W.tmp[r]) == 0) ? 1 : 0; }  // fresh
              g.sync();
              for (int r = t; r < R; r += T) { const int gq = LST[a + r]; W.tmp[r] = (gq < R) ? gq : r; }
              g.sync();
              for (;;) {  // pointer jumping, in place
                int changed = 0;
                for (int r = t; r < R; r += T) {
                  const int p1 = W.tmp[r];
                  const int p2 = W.tmp[p1];
                  if (p2 != p1) { W.tmp[r] = p2; changed = 1; }
                }
                if (g.reduce_max(changed) == 0) break;
              }
              g.sync();
              for (int r = t; r < R; r += T) {
                const int gq = LST[a + r];
                W.sa[r] = W.cols[s + gq];
                if (static_cast<int>(W.sc[r])) {
                  W.sb[r] = W.cols[s + r];
                  W.sc[r] = 1 + s + static_cast<int>(LST[a + static_cast<int>(W.tmp[r])]);  // 1 + destination of the displaced item
                }
              }
              g.sync();  // every read of the old order is done
              for (int r = t; r < R; r += T) {
                const int jr = W.sa[r], dd = W.sc[r];
                W.cols[s + r] = jr; W.inv[jr] = s + r;
                if (dd) { const int og = W.sb[r]; W.cols[dd - 1] = og; W.inv[og] = dd - 1; }
              }
              g.sync();
              g.lds_barriers(find_lds);
              h2 = static_cast<unsigned>(s + R);
            } else {
              // The scratch of the closed form (fresh flags, then the pointer-jumping array: chains of displaced items can be as long as the
              // run, i.e. ~log2(R) rounds of two dependent reads and a reduction each) lives in LDS when the fast scratch is there and the run
              // fits (the event tables are free during a find): its phases then meet at LDS-only barriers.
              auto tie_closed_form = [&](const auto& TMP, const bool in_lds) -> unsigned {
                auto sync_t = [&]() { if (in_lds) g.sync_lds(); else g.sync(); };
                const int a = nser;  // first tie record; the serial part left the insertion point at s = lo + 1 = first
                const int s = first;
                int gr[kTiePer];
  #pragma unroll
                for (int q = 0; q < kTiePer; ++q) {
                  const int r = t + q * T;
                  gr[q] = (r < R) ? static_cast<int>(LST[a + r]) : 0;  // g(r) = k_r - s
                  if (r < R) TMP[r] = 0;
                }
                sync_t();
  #pragma unroll
                for (int q = 0; q < kTiePer; ++q) {
                  const int r = t + q * T;
                  if (r < R && gr[q] < R && gr[q] != r) TMP[gr[q]] = 1;  // position s + g is record r's k: not fresh
                }
                sync_t();
                unsigned fresh = 0;
  #pragma unroll
                for (int q = 0; q < kTiePer; ++q) {
                  const int r = t + q * T;
                  if (r < R && gr[q] != r && TMP[r] == 0) fresh |= 1u << q;
                }
                sync_t();
  #pragma unroll
                for (int q = 0; q < kTiePer; ++q) {
                  const int r = t + q * T;
                  if (r < R) TMP[r] = (gr[q] < R) ? gr[q] : r;
                }
                sync_t();
                g.lds_barriers(in_lds);
                for (;;) {  // pointer jumping, in place: a racing reader sees an older or a newer node of the same chain
                  int changed = 0;
  #pragma unroll
                  for (int q = 0; q < kTiePer; ++q) {
                    const int r = t + q * T;
                    if (r < R) {
                      const int p1 = TMP[r];
                      const int p2 = TMP[p1];
                      if (p2 != p1) { TMP[r] = p2; changed = 1; }
                    }
                  }
                  if (g.reduce_max(changed) == 0) break;
                }
                g.lds_barriers(find_lds);
                int jr[kTiePer], orig[kTiePer], dst[kTiePer];
  #pragma unroll
                for (int q = 0; q < kTiePer; ++q) {
                  const int r = t + q * T;
                  jr[q] = 0; orig[q] = 0; dst[q] = 0;
                  if (r < R) {
                    jr[q] = W.cols[s + gr[q]];
                    if (fresh & (1u << q)) {
                      orig[q] = W.cols[s + r];
                      dst[q] = s + static_cast<int>(LST[a + static_cast<int>(TMP[r])]);
                    }
                  }
                }
                if (in_lds && find_lds) g.sync_lds(); else g.sync();  // every read of the old order is done
  #pragma unroll
                for (int q = 0; q < kTiePer; ++q) {
                  const int r = t + q * T;
                  if (r < R) {
                    W.cols[s + r] = jr[q]; W.inv[jr[q]] = s + r;
                    if (fresh & (1u << q)) { W.cols[dst[q]] = orig[q]; W.inv[orig[q]] = dst[q]; }
                  }
                }
                if (in_lds && find_lds) g.sync_lds(); else g.sync();
                return static_cast<unsigned>(s + R);
              };
              constexpr int kTmpLds = kFsTodo - kFsEvl;  // ints of the event tables
              if (use_rl && R <= kTmpLds) {
                const std::remove_cv_t<std::remove_reference_t<decltype(W.fsw)>> TL{W.fsw.raw(kFsEvl)};
                h2 = tie_closed_form(TL, true);
              } else {
                h2 = tie_closed_form(W.tmp, false);
              }
            }
            cy_sub[11] += MOT_FCLOCK() - qf2;
            int best = -1;
            for (unsigned k = lo + static_cast<unsigned>(t); k < h2; k += static_cast<unsigned>(T))
              if (static_cast<int>(W.y[static_cast<int>(W.cols[k])]) < 0) best = static_cast<int>(k);  // the LAST free member (:174-177)
            best = g.reduce_max(best);
            if (use_rl)
              for (unsigned k = lo + static_cast<unsigned>(t); k < h2; k += static_cast<unsigned>(T)) {
                const int j = W.cols[k];
                W.fsw.atomic_and(kFsTodo + (j >> 5), ~(1 << (j & 31)));
              }
            if constexpr (kLst16) {
              if (use_rl) {  // the record list lay over the steps' column table: empty again (every read of the list is behind a barrier by now)
                g.sync_lds();
                for (int w = t; w < kFsHash; w += T) { W.fsw[kFsKeep + w] = -1; W.fsw[kFsKeep + kFsHash + w] = kNoIdx; }
              }
            }
            hi = h2;
            final_j = (best >= 0) ? static_cast<int>(W.cols[best]) : -1;
          }
          fsync();
          g.lds_barriers(false);
          cy_find += MOT_FCLOCK() - qf0;
        }
        if (final_j == -1) {
          // _scan_dense (:129-155) on local copies of lo/hi, written back only on normal exit
          unsigned slo = lo, shi = hi;
          bool returned = false;
          // software pipeline: the next member of the SCAN set, its row, distance and row box are fetched while the
          // current one is swept (cols[] below shi, y[] and the d[] of SCAN members do not change during a sweep)
          // On wavefront hardware the upcoming members are read 64 at a time, member slo + l by lane l of every wavefront, and
          // handed out with v_readlane: the dependent cols[] -> y[] -> d[] loads (global memory for the wide problems) are
          // paid once per 64 members instead of once per sweep. Members that join the set later are picked up by the next fill.
          constexpr bool kWin = has_wave_table<G>::value;
          constexpr bool kColsLds = std::remove_reference_t<decltype(W)>::kColsSpace == kMemLds;
          int win_j = 0, win_i = 0;
          double win_d = 0.0;
          unsigned win_base = 0, win_cnt = 0;
          (void)win_j; (void)win_i; (void)win_d; (void)win_base; (void)win_cnt;
          int pq_j, pq_i;
          double pq_d;
          auto member = [&](unsigned pos) {  // pq_* = member `pos` of the SCAN set (pos < shi)
            if constexpr (kWin) {
              if (pos - win_base >= win_cnt) {
                win_base = pos;
                win_cnt = (shi - pos < 64u) ? (shi - pos) : 64u;
                const unsigned idx = pos + static_cast<unsigned>(g.wave_lane());
                if (idx < shi) {
                  win_j = W.cols[idx];
                  win_i = W.y[win_j];
                  win_d = W.d[win_j];
                }
              }
              const int l = static_cast<int>(pos - win_base);
              pq_j = G::wave_get(win_j, l);
              pq_i = G::wave_get(win_i, l);
              pq_d = G::wave_get(win_d, l);
            } else {
              pq_j = W.cols[pos];
              pq_i = W.y[pq_j];
              pq_d = W.d[pq_j];
            }
          };
          member(slo);
          constexpr int kSweep = 4;
          float pf[kSweep] = {0.f, 0.f, 0.f, 0.f}, pf_next[kSweep] = {0.f, 0.f, 0.f, 0.f};
          float pf_h = 0.f, pf_next_h = 0.f;
          int pf_row = -1, pf_next_row = -1;  // the matrix row pf[] / pf_next[] hold (-1: none)
          int pf_col = -1, pf_next_col = -1;  // the column pf_h / pf_next_h is the element of
          (void)pf; (void)pf_next; (void)pf_row; (void)pf_next_row; (void)pf_h; (void)pf_next_h; (void)pf_col; (void)pf_next_col;
          // ---- parallel scan steps (row lists present) ----
          // lapjv sweeps the members of the SCAN set one at a time; all of them sit at the same distance `mind`. Two exact facts
          // make most sweeps cheap and independent of each other:
          //  (1) a dummy row whose h does not exceed the largest h of a dummy row swept before changes nothing (above);
          //  (2) for a REAL row i with h_i <= that same bound, every real column j with cost(i,j) >= half is a no-op as well:
          //      d[j] <= fl(fl(half - v[j]) - hmax_dummy_row) <= fl(fl(cost(i,j) - v[j]) - h_i) (rounding is monotone) — so its
          //      sweep only needs the row's list of entries below half (and, while h_i <= hmax_real_row, no dummy column).
          // Up to T consecutive members of those two kinds are handled by one step: each lane classifies one member; the
          // real rows' list entries are relaxed together: distances by an atomic minimum (every value is >= mind >= 0, so the bit
          // patterns order like the doubles); of the relaxations that attain a column's new distance the EARLIEST member is
          // the predecessor (a later equal value does not replace it in lapjv either) — they meet in a small table keyed by
          // the column; values that tie with `mind` become events and are applied in lapjv's order: by member, then by position
          // in cols[] at that member's turn. A step stops in front of the first member of any other kind, which takes the
          // one-at-a-time sweep below. If a relaxed value falls below mind (possible only through rounding) the rest of the
          // SCAN set is swept one member at a time.
          // Memory traffic of a step: the duals, distances, column->row map, the TODO bitmask and every table of the step are
          // in LDS (lds mode 4), and the barriers inside a step order LDS only, so that its few global accesses — the rows' lists,
          // the positions of tie columns, the stores to cols[] / inv[] / pred[] — overlap instead of costing a round trip each;
          // ONE full barrier per step makes the previous step's stores visible. The members a step appends are classified by
          // the next step from the event table (row, cost and list length were fetched with the event), not from memory.
          // units of list entries one step holds (kFsIter passes of T lanes, kFsUnit entries per unit); a row is steppable when its units fit
          const int units_cap = use_rl ? ((((kFsIter * T) / kFsUnit) < kFsMaxUnits) ? (kFsIter * T) / kFsUnit : kFsMaxUnits) : 0;
          bool fast_ok = units_cap > 0;
          constexpr int kMQ = kFsM, kMROW = kFsM + kFsMaxMembers, kMH = kFsM + 2 * kFsMaxMembers, kUROW = kFsUnits, kUINF = kFsUnits + kFsMaxUnits;
          constexpr int kCEV = kFsCtr, kCDONE = kFsCtr + 2, kCSINK = kFsCtr + 3;
          constexpr int kHK = kFsKeep, kHQ = kFsKeep + kFsHash;
          constexpr int kEQ = kFsEvl, kEJ = kFsEvl + kEvCap, kEK = kFsEvl + 2 * kEvCap, kEI = kFsEvl + 3 * kEvCap, kEC = kFsEvl + 4 * kEvCap, kEN = kFsEvl + 5 * kEvCap;
          constexpr int kSQ = kFsEvs, kSJ = kFsEvs + kEvCap, kSK = kFsEvs + 2 * kEvCap, kSF = kFsEvs + 3 * kEvCap, kSI = kFsEvs + 4 * kEvCap,
                        kSC = kFsEvs + 5 * kEvCap, kSN = kFsEvs + 6 * kEvCap, kHC = kFsEvs + 7 * kEvCap, kHE = kFsEvs + 8 * kEvCap;
          unsigned qpos = 0u, qlen = 0u;   // SCAN positions [qpos, qpos + qlen) were appended by the last step: their members are in the sorted event table
          bool need_full = false;          // stores to cols[] that the next classification reads from memory are still in flight
          bool stepped = false;            // a step ran in this SCAN set (its stores need a full barrier before anyone else reads them)
          // the matched cost of the member this lane will classify in the NEXT step, requested as soon as this step knows how many members it takes:
          // the one global load on a step's classification path (the costs do not fit next to the LDS state), a round trip every wavefront waited for
          unsigned pfc_idx = 0xffffffffu;
          float pfc_cost = 0.f;
          auto fast_step_body = [&](double mind_b) -> int {  // 0: nothing done, 1: members consumed, 2: a sink was reached (final_j set)
            const long long q0 = MOT_FCLOCK();
            if (need_full) { if constexpr (kColsLds) g.sync_lds(); else g.sync(); need_full = false; }
            // one member per lane. (Tried in round 5: four consecutive members per lane — two steps in three of an OC-SORT 4096 x 2048 problem end
            // after T members because its SCAN sets hold thousands of dummy rows; 16 % fewer steps, each so much longer that the solve took 13 %
            // more cycles: a step costs its critical path, and the classification is on it.)
            const unsigned idx = slo + static_cast<unsigned>(t);
            int cls = 0, mi = -1, mcnt = 0;  // 0 stop, 1 void dummy row, 2 real row with a list
            double hh = 0.0;
            if (idx < shi) {
              const unsigned qi = idx - qpos;
              int mj;
              double md, cij = half;
              if (qi < qlen) {  // appended by the last step: d == mind by construction
                mj = W.fsw[kSJ + qi]; mi = W.fsw[kSI + qi]; md = mind_b;
                mcnt = W.fsw[kSN + qi];
                if (mi < nr && mj < nc) cij = static_cast<double>(__builtin_bit_cast(float, static_cast<int>(W.fsw[kSC + qi])));
              } else {
                mj = W.cols[idx];
                mi = W.y[mj];
                md = W.d[mj];
                if (mi < nr) {
                  mcnt = rl_len(mi);
                  if (mj < nc) cij = (have_yc && pfc_idx == idx) ? static_cast<double>(pfc_cost) : match_cost(mi, mj);
                }
              }
              if (md == mind_b) {
                if (mi >= nr) {
                  hh = ((mj < nc) ? half : 0.0) - W.v[mj] - md;
                  if (hh <= hmax_dummy_row) cls = 1;
                } else {
                  hh = cij - W.v[mj] - md;
                  if (hh <= hmax_dummy_row && hh <= hmax_real_row && mcnt <= kRlCap && mcnt <= units_cap * kFsUnit) cls = 2;
                }
              }
            }
            const long long qa = MOT_FCLOCK();
            cy_sub[0] += qa - q0;
            // Real rows with a list take part in the step through their list entries, dealt to the lanes in UNITS of kFsUnit entries (a row of
            // mcnt entries takes ceil(mcnt / kFsUnit) units; round 5 — a row used to reserve kRlCap slots whatever its length, so that a
            // step of 512 lanes held 64 rows with nine slots in ten empty). ONE collective gives the number of leading members the step can take
            // (a lane beyond the SCAN set has cls 0: it stops the count) and every real row among them its rank and its first unit.
            const bool spc = cls == 2 && mcnt > 0;  // (a real row without entries below half relaxes nothing)
            const int nu = spc ? ((mcnt + kFsUnit - 1) / kFsUnit) : 0;
            const auto sc = g.scan_until_stop(spc ? (1 | (nu << 12)) : 0, cls == 0);
            int cnt = sc.cnt;
            if (cnt == 0) { cy_cls += MOT_FCLOCK() - q0; return 0; }
            if (have_yc) {  // (the step may still end earlier than cnt — the units cap, the one-row fallback: then the prefetched index does not match and the load is repeated)
              const unsigned nidx = slo + static_cast<unsigned>(cnt) + static_cast<unsigned>(t);
              pfc_idx = 0xffffffffu;
              if (nidx < shi) {
                const int nj = W.cols[nidx];
                if (nj < nc) { pfc_cost = W.ycost[nj]; pfc_idx = nidx; }
              }
            }
            const bool sp = spc && t < cnt;
            const int rank = sc.base & 0xfff, ubase = sc.base >> 12;
            int ns = sc.tot & 0xfff, U = sc.tot >> 12;
            if (U > units_cap) {  // the step ends in front of the first member whose units do not fit any more (a prefix property: unit offsets only grow)
              const int key = g.reduce_min_int((sp && ubase + nu > units_cap) ? ((t << 20) | (rank << 11) | ubase) : kNoIdx);
              cnt = key >> 20; ns = (key >> 11) & 0x1ff; U = key & 0x7ff;
            }
            if (ns == 0) { slo += static_cast<unsigned>(cnt); ++n_fs_steps; n_fs_members += cnt; cy_cls += MOT_FCLOCK() - q0; return 1; }
            if (sp && rank < ns) {
              const long long hb = __builtin_bit_cast(long long, hh);
              W.fsw[kMQ + rank] = t;
              W.fsw[kMROW + rank] = mi;
              W.fsw[kMH + 2 * rank] = static_cast<int>(hb & 0xffffffffll);
              W.fsw[kMH + 2 * rank + 1] = static_cast<int>(hb >> 32);
#pragma unroll
              for (int k = 0; k < kRlCap / kFsUnit; ++k)
                if (k < nu) { W.fsw[kUROW + ubase + k] = mi; W.fsw[kUINF + ubase + k] = rank | (k << 10) | (mcnt << 12); }
            }
            if (t == 0) W.fsw[kCEV] = 0;
            g.sync_lds();
            const int niter = (kFsUnit * U + T - 1) / T;  // (uniform) passes of T lanes over the step's entries: <= kFsIter
            const long long q1 = MOT_FCLOCK();
            cy_cls += q1 - q0;
            cy_sub[1] += q1 - qa;
            // every list entry of the step's real rows, held in registers
            int ej[kFsIter], eq[kFsIter], er[kFsIter], erk[kFsIter];
            float cc[kFsIter];
            double ec[kFsIter];
            unsigned keep = 0u;
            int bad = 0, nt = 0, nk = 0;
            // (loads first, arithmetic after — and every load unconditional, from an address that is valid whatever the lane's slot holds: a load
            // inside a divergent branch is waited for inside that branch, one round trip per entry instead of one for all of them)
            auto fetch = [&]() {
              int urow[kFsIter], uinf[kFsIter];
#pragma unroll
              for (int it = 0; it < kFsIter; ++it) {
                urow[it] = 0; uinf[it] = 0;
                if (it < niter) {
                  const int u = (t + it * T) / kFsUnit;
                  const int uu = (u < U) ? u : 0;
                  urow[it] = W.fsw[kUROW + uu];
                  uinf[it] = W.fsw[kUINF + uu];
                }
              }
              unsigned long long ent[kFsIter];
#pragma unroll
              for (int it = 0; it < kFsIter; ++it) {
                ent[it] = 0ull;
                if (it < niter) {
                  const int e = ((uinf[it] >> 10) & 3) * kFsUnit + ((t + it * T) % kFsUnit);
                  ent[it] = W.rl_ent[static_cast<long>(urow[it]) * kRlCap + e];
                }
              }
#pragma unroll
              for (int it = 0; it < kFsIter; ++it) {
                ej[it] = -1; cc[it] = 0.f; er[it] = urow[it]; erk[it] = uinf[it] & 0x3ff;
                if (it < niter) {
                  const int u = (t + it * T) / kFsUnit;
                  const int e = ((uinf[it] >> 10) & 3) * kFsUnit + ((t + it * T) % kFsUnit);
                  if (u < U && e < (uinf[it] >> 12)) { ej[it] = rl_col_of(ent[it]); cc[it] = rl_cost_of(ent[it]); }
                }
              }
            };
            auto evaluate = [&](int ns_now) {
              keep = 0u; nt = 0; nk = 0;
              int tw[kFsIter], hlo[kFsIter], hhi[kFsIter], mq[kFsIter];
              double vv[kFsIter], dd[kFsIter];
#pragma unroll
              for (int it = 0; it < kFsIter; ++it) {
                tw[it] = 0; hlo[it] = 0; hhi[it] = 0; mq[it] = 0; vv[it] = 0.0; dd[it] = 0.0;
                if (it < niter) {
                  const int jc = (ej[it] >= 0) ? ej[it] : 0;
                  tw[it] = W.fsw[kFsTodo + (jc >> 5)];
                  vv[it] = W.v[jc];
                  dd[it] = W.d[jc];
                  hlo[it] = W.fsw[kMH + 2 * erk[it]];
                  hhi[it] = W.fsw[kMH + 2 * erk[it] + 1];
                  mq[it] = W.fsw[kMQ + erk[it]];
                }
              }
#pragma unroll
              for (int it = 0; it < kFsIter; ++it) {
                const int j = ej[it];
                eq[it] = 0; ec[it] = 0.0;
                if (it < niter && j >= 0 && erk[it] < ns_now && ((tw[it] >> (j & 31)) & 1)) {
                  const long long hb = (static_cast<long long>(hhi[it]) << 32) | static_cast<long long>(static_cast<unsigned>(hlo[it]));
                  const double cred = static_cast<double>(cc[it]) - vv[it] - __builtin_bit_cast(double, hb);
                  if (!(cred >= mind_b)) bad = 1;
                  if (cred < dd[it]) {
                    keep |= 1u << it;
                    eq[it] = mq[it]; ec[it] = cred;
                    ++nk;
                    if (cred == mind_b) ++nt;
                  }
                }
              }
            };
            fetch();
            const long long qb = MOT_FCLOCK();
            cy_sub[2] += qb - q1;
            // the step's barrier, with the list loads in flight: the previous step's stores to cols[] / inv[] are visible from here on. With
            // cols[] / inv[] in LDS it orders LDS only; what is left in global memory is pred[] (written by the steps, read when the search
            // is over): wait_vm() below makes every wavefront's stores to it land before the next step's — which sit behind this barrier.
            if constexpr (!kColsLds) g.sync();  // (LDS state: the barriers of this step's classification already stand between the previous step's stores and every read below)
            stepped = true;
            // the columns in the first TODO positions (where tie events will swap their columns to) are requested now, one per lane, and
            // consumed by the event sort three phases later: a global round trip off the step's critical path
            const int hcol_pref = (t < kEvCap && shi + static_cast<unsigned>(t) < static_cast<unsigned>(n)) ? static_cast<int>(W.cols[shi + static_cast<unsigned>(t)]) : 0;
            const long long qc = MOT_FCLOCK();
            cy_sub[3] += qc - qb;
            if constexpr (kColsLds) g.wait_vm();  // (the list entries have to be here anyway)
            evaluate(ns);
            const long long qd = MOT_FCLOCK();
            cy_sub[4] += qd - qc;
            // one reduction for the three counts (sums of nk and nt <= kFsIter * T <= 4096: 13 bits each; "bad" once per wavefront above them)
            int packed3, bad_any;
            if constexpr (has_wave_table<G>::value) {
              const bool wb = G::wave_ballot(bad != 0) != 0ull;
              packed3 = g.reduce_sum(nk | (nt << 13) | ((wb && g.wave_lane() == 0) ? (1 << 26) : 0));
              bad_any = packed3 >> 26;
            } else {
              bad_any = g.reduce_max(bad);
              packed3 = g.reduce_sum(nk | (nt << 13));
            }
            if (bad_any) { fast_ok = false; ++n_fs_bad; return 0; }
            const int packed_all = ((packed3 >> 13) & 0x1fff) << 16 | (packed3 & 0x1fff);
            if ((packed_all >> 16) > kEvCap || (packed_all & 0xffff) > kKeepCap) {  // more than the tables hold: this step takes one real row only (<= kRlCap entries)
              ns = 1;
              cnt = static_cast<int>(W.fsw[kMQ]) + 1;
              evaluate(1);
            }
            const long long q2 = MOT_FCLOCK();
            cy_dry += q2 - q1;
            cy_sub[5] += q2 - qd;
            // 1. distances
#pragma unroll
            for (int it = 0; it < kFsIter; ++it)
              if (keep & (1u << it)) mem_atomic_min_f64_nonneg<std::remove_reference_t<decltype(W.d)>::kSpace>(W.d.raw(ej[it]), ec[it]);
            if constexpr (std::remove_reference_t<decltype(W.d)>::kSpace == kMemGlobal) g.sync(); else g.sync_lds();  // (orders the accesses to d[])
            const long long qe = MOT_FCLOCK();
            cy_sub[6] += qe - q2;
            // 2. the earliest member among the relaxations that attain a column's new distance
            int es[kFsIter];
            unsigned ach = 0u, tie = 0u;
            int tk[kFsIter], ti[kFsIter], tn[kFsIter];
            float tc[kFsIter];
#pragma unroll
            for (int it = 0; it < kFsIter; ++it) {
              es[it] = 0; tk[it] = 0; ti[it] = 0; tn[it] = 0; tc[it] = 0.f;
              if ((keep & (1u << it)) && static_cast<double>(W.d[ej[it]]) == ec[it]) {
                int slot = static_cast<int>((static_cast<unsigned>(ej[it]) * 2654435761u) >> 22) & (kFsHash - 1);
                for (;;) {
                  const int old = W.fsw.atomic_cas(kHK + slot, -1, ej[it]);
                  if (old == -1 || old == ej[it]) break;
                  slot = (slot + 1) & (kFsHash - 1);
                }
                W.fsw.atomic_min(kHQ + slot, eq[it]);
                es[it] = slot;
                ach |= 1u << it;
                if (ec[it] == mind_b) {
                  // a tie with mind: if this relaxation wins it becomes an event. What the event needs from memory — the column's
                  // position in cols[] (the order of the events), and for the member it appends the row's list length and its cost
                  // at the column — is requested now and arrives behind the barrier
                  tie |= 1u << it;
                  const int j = ej[it], i = W.y[j];
                  tk[it] = W.inv[j];
                  ti[it] = i;
                  if (i >= 0 && i < nr) {
                    tn[it] = rl_len(i);
                    if (j < nc) tc[it] = have_yc ? static_cast<float>(W.ycost[j]) : static_cast<float>(C.at(i, j));
                  }
                }
              }
            }
            g.sync_lds();
            const long long qg = MOT_FCLOCK();
            cy_sub[7] += qg - qe;
            // 3. predecessors; the winners that tie with mind become events
            unsigned win = 0u;
#pragma unroll
            for (int it = 0; it < kFsIter; ++it) {
              if ((ach & (1u << it)) && static_cast<int>(W.fsw[kHQ + es[it]]) == eq[it]) {
                win |= 1u << it;
                W.pred[ej[it]] = er[it];
              } else {
                tie &= ~(1u << it);
              }
            }
#pragma unroll
            for (int it = 0; it < kFsIter; ++it)
              if (tie & (1u << it)) {
                const int e = W.fsw.atomic_add(kCEV, 1);
                W.fsw[kEQ + e] = eq[it]; W.fsw[kEJ + e] = ej[it]; W.fsw[kEK + e] = tk[it];
                W.fsw[kEI + e] = ti[it]; W.fsw[kEC + e] = __builtin_bit_cast(int, tc[it]); W.fsw[kEN + e] = tn[it];
              }
            g.sync_lds();
#pragma unroll
            for (int it = 0; it < kFsIter; ++it)
              if (win & (1u << it)) { W.fsw[kHK + es[it]] = -1; W.fsw[kHQ + es[it]] = kNoIdx; }  // the table is empty again for the next step
            const int nev = W.fsw[kCEV];
            const long long q3 = MOT_FCLOCK();
            cy_apply += q3 - q2;
            cy_sub[8] += q3 - qg;
            ++n_fs_steps; n_fs_members += cnt; n_fs_sparse += ns; n_fs_events += nev;
            qlen = 0u;
            if (nev == 0) { slo += static_cast<unsigned>(cnt); return 1; }
            // tie events in lapjv's order. Sorted by (member, position at the start of the step); positions only change for
            // columns that sit in the first nev TODO positions ("head slots": a tie swaps its column with the first TODO one).
            // What can stop the events from being independent swaps of (event column, head column) pairs in sorted order: an event column that sits in a
            // head slot (one of the first nev TODO positions). Event e (sorted index) with its column in head slot o != e is harmless for the
            // prefix [0, max(e, o)): up to there no turn touches that slot or that event. So (round 5) the longest conflict-free prefix P = the
            // smallest max(e, o) over such events is applied in parallel and only the events behind it take the serial walk — which used to start
            // at event 0 whenever any event column sat in a head slot (one step in eight, 40 k cycles each). The first event whose column is free
            // (a sink) ends everything: one reduction finds the smaller of the two bounds (a sink wins a tie: the prefix in front of it is applied,
            // then the search is over).
            int lane_key = kNoIdx;
            for (int e = t; e < nev; e += T) {
              const int hcol = (e == t && t < kEvCap) ? hcol_pref : static_cast<int>(W.cols[shi + static_cast<unsigned>(e)]);
              const int q = W.fsw[kEQ + e], j = W.fsw[kEJ + e], k = W.fsw[kEK + e], i = W.fsw[kEI + e];
              const int fl = (i < 0) ? 1 : 0;
              int rk = 0;
              for (int o = 0; o < nev; ++o) {
                const int oq = W.fsw[kEQ + o], ok = W.fsw[kEK + o];
                rk += (oq < q || (oq == q && ok < k)) ? 1 : 0;
              }
              W.fsw[kSQ + rk] = q; W.fsw[kSJ + rk] = j; W.fsw[kSK + rk] = k; W.fsw[kSF + rk] = fl;
              W.fsw[kSI + rk] = i; W.fsw[kSC + rk] = W.fsw[kEC + e]; W.fsw[kSN + rk] = W.fsw[kEN + e];
              W.fsw[kHC + e] = hcol;
              W.fsw[kHE + e] = -1;
              const int off = k - static_cast<int>(shi);
              if (fl && (rk << 1) < lane_key) lane_key = rk << 1;
              if (off < nev && off != rk) { const int cb = ((rk > off ? rk : off) << 1) | 1; if (cb < lane_key) lane_key = cb; }
            }
            const int key = g.reduce_min_int(lane_key);  // (its barrier also orders the sorted table)
            const int P = (key == kNoIdx) ? nev : (key >> 1);
            const bool conflict = key != kNoIdx && (key & 1) != 0;
            const long long q4 = MOT_FCLOCK();
            cy_evsort += q4 - q3;
            int done, sink = -1;
            if (conflict) {  // which event's column sits in which head slot
              for (int e = t; e < nev; e += T) {
                const int off = static_cast<int>(W.fsw[kSK + e]) - static_cast<int>(shi);
                if (off < nev) W.fsw[kHE + off] = e;
              }
              g.sync_lds();
            }
            for (int r = t; r < P; r += T) {  // the conflict-free prefix: independent swaps
              const int j = W.fsw[kSJ + r], bp = W.fsw[kSK + r], hc = W.fsw[kHC + r];
              const int hpos = static_cast<int>(shi) + r;
              if (!conflict) { W.cols[bp] = hc; W.inv[hc] = bp; }
              else {
                if (bp != hpos) {
                  // the head column moves to the event column's old position; when it is itself the column of a later event (he >= P) or that
                  // position is a later head slot (off >= P), the tables the serial walk reads are kept current instead
                  const int he = W.fsw[kHE + r], off = bp - static_cast<int>(shi);
                  if (off < nev) { W.fsw[kHC + off] = hc; W.fsw[kHE + off] = he; }
                  else { W.cols[bp] = hc; W.inv[hc] = bp; }
                  if (he >= 0) W.fsw[kSK + he] = bp;
                }
                W.fsw[kSF + r] = static_cast<int>(W.fsw[kSF + r]) | 2;
              }
              W.cols[hpos] = j; W.inv[j] = hpos;
              W.fsw.atomic_and(kFsTodo + (j >> 5), ~(1 << (j & 31)));
            }
            if (!conflict) {
              done = P;
              if (key != kNoIdx) sink = W.fsw[kSJ + P];
              qpos = shi; qlen = static_cast<unsigned>(done);  // the appended members, in order, are the first `done` sorted events
            } else {
              g.sync_lds();
              // one lane, everything through LDS: groups without wavefronts (tests/emu), and a member with more than 64 events
              auto serial_replay = [&](int r0, int e0, int& dn, int& sk) {
                for (int r = r0; r < nev; ++r) {
                  while (static_cast<int>(W.fsw[kSF + e0]) & 2) ++e0;
                  const int q = W.fsw[kSQ + e0];
                  int best = e0, bp = W.fsw[kSK + e0];
                  for (int e = e0 + 1; e < nev && static_cast<int>(W.fsw[kSQ + e]) == q; ++e) {
                    const int pk = W.fsw[kSK + e];
                    if (!(static_cast<int>(W.fsw[kSF + e]) & 2) && pk < bp) { best = e; bp = pk; }
                  }
                  const int j = W.fsw[kSJ + best], fl = W.fsw[kSF + best];
                  if (fl & 1) { sk = j; break; }
                  W.fsw[kSF + best] = fl | 2;
                  const int hpos = static_cast<int>(shi) + r;
                  if (bp != hpos) {
                    const int hc = W.fsw[kHC + r], he = W.fsw[kHE + r];
                    const int off = bp - static_cast<int>(shi);
                    if (off < nev) { W.fsw[kHC + off] = hc; W.fsw[kHE + off] = he; }
                    else { W.cols[bp] = hc; W.inv[hc] = bp; }
                    if (he >= 0) W.fsw[kSK + he] = bp;
                  }
                  W.cols[hpos] = j; W.inv[j] = hpos;
                  W.fsw[kFsTodo + (j >> 5)] = static_cast<int>(W.fsw[kFsTodo + (j >> 5)]) & ~(1 << (j & 31));
                  ++dn;
                }
              };
              bool by_wave = false;
              if constexpr (has_wave_table<G>::value) {
                by_wave = true;
                if (t < 64) {
                  int dn = P, sk = -1;  // (the prefix [0, P) is applied)
                  if (nev <= 64) {
                    // the first wavefront replays out of registers: lane e holds sorted event e and head slot e, the serial walk reads
                    // them with v_readlane and finds the next event with one ballot and one DPP minimum (an LDS round trip per field
                    // and event otherwise)
                    const bool in = t < nev;
                    const int vq = in ? static_cast<int>(W.fsw[kSQ + t]) : kNoIdx, vj = in ? static_cast<int>(W.fsw[kSJ + t]) : 0;
                    int vk = in ? static_cast<int>(W.fsw[kSK + t]) : kNoIdx, vf = in ? static_cast<int>(W.fsw[kSF + t]) : 2;
                    int hc = in ? static_cast<int>(W.fsw[kHC + t]) : 0, he = in ? static_cast<int>(W.fsw[kHE + t]) : -1;
                    for (int r = P; r < nev; ++r) {
                      const int e0 = __builtin_ctzll(G::wave_ballot(!(vf & 2)));  // first event not applied yet, in sorted order
                      const int q = G::wave_get(vq, e0);
                      const int cand = (!(vf & 2) && vq == q) ? vk : kNoIdx;      // its member's events: the one lowest in cols[] NOW is next
                      const int bp = G::wave_min_i32(cand);
                      const int best = __builtin_ctzll(G::wave_ballot(cand == bp));
                      const int j = G::wave_get(vj, best), fl = G::wave_get(vf, best);
                      if (fl & 1) { sk = j; break; }
                      if (t == best) vf |= 2;
                      const int hpos = static_cast<int>(shi) + r;
                      if (bp != hpos) {
                        const int hcr = G::wave_get(hc, r), her = G::wave_get(he, r);
                        const int off = bp - static_cast<int>(shi);
                        if (off < nev) { if (t == off) { hc = hcr; he = her; } }
                        else if (t == 0) { W.cols[bp] = hcr; W.inv[hcr] = bp; }
                        if (her >= 0 && t == her) vk = bp;
                      }
                      if (t == 0) {
                        W.cols[hpos] = j; W.inv[j] = hpos;
                        W.fsw[kFsTodo + (j >> 5)] = static_cast<int>(W.fsw[kFsTodo + (j >> 5)]) & ~(1 << (j & 31));
                      }
                      ++dn;
                    }
                  } else {
                    // (round 5) more than 64 events: the same walk member by member — a member's events are contiguous in the sorted table, lane l
                    // holds the l-th of them; the head slots stay in LDS (one uniform read per event: the wavefront's own LDS accesses run in
                    // order, so a slot rewritten by one event is seen by the next). This used to fall to the one-lane loop above: ~3 k cycles
                    // per event, and the steps that take this path are the ones with many events.
                    int e0 = P;
                    bool fb = false;
                    while (e0 < nev && sk < 0) {
                      const int q = W.fsw[kSQ + e0];
                      const int e = e0 + t;
                      const bool in = e < nev && static_cast<int>(W.fsw[kSQ + (e < nev ? e : e0)]) == q;
                      const int cm = __builtin_popcountll(G::wave_ballot(in));
                      if (cm == 64 && e0 + 64 < nev && static_cast<int>(W.fsw[kSQ + e0 + 64]) == q) { fb = true; break; }
                      const int vj = in ? static_cast<int>(W.fsw[kSJ + e]) : 0;
                      int vk = in ? static_cast<int>(W.fsw[kSK + e]) : kNoIdx, vf = in ? static_cast<int>(W.fsw[kSF + e]) : 2;
                      for (int a = 0; a < cm; ++a) {
                        const int cand = !(vf & 2) ? vk : kNoIdx;
                        const int bp = G::wave_min_i32(cand);
                        const int best = __builtin_ctzll(G::wave_ballot(cand == bp));
                        const int j = G::wave_get(vj, best), fl = G::wave_get(vf, best);
                        if (fl & 1) { sk = j; break; }
                        if (t == best) vf |= 2;
                        const int hpos = static_cast<int>(shi) + dn;
                        if (bp != hpos) {
                          const int hcr = W.fsw[kHC + dn], her = W.fsw[kHE + dn];
                          const int off = bp - static_cast<int>(shi);
                          if (off < nev) { if (t == 0) { W.fsw[kHC + off] = hcr; W.fsw[kHE + off] = her; } }
                          else if (t == 0) { W.cols[bp] = hcr; W.inv[hcr] = bp; }
                          if (her >= 0) {
                            if (t == her - e0) vk = bp;             // (an event of this member, held by a lane; harmless for a lane that holds none)
                            if (t == 0) W.fsw[kSK + her] = bp;      // (and for the members still to come)
                          }
                        }
                        if (t == 0) {
                          W.cols[hpos] = j; W.inv[j] = hpos;
                          W.fsw[kFsTodo + (j >> 5)] = static_cast<int>(W.fsw[kFsTodo + (j >> 5)]) & ~(1 << (j & 31));
                        }
                        ++dn;
                      }
                      if (in) W.fsw[kSF + e] = vf;  // (the applied flags, for the one-lane continuation)
                      e0 += cm;
                    }
                    if (fb && t == 0) serial_replay(dn, e0, dn, sk);
                  }
                  if (t == 0) { W.fsw[kCDONE] = dn; W.fsw[kCSINK] = sk; }
                }
              }
              if (!by_wave && t == 0) {
                int dn = P, sk = -1;
                serial_replay(P, P, dn, sk);
                W.fsw[kCDONE] = dn;
                W.fsw[kCSINK] = sk;
              }
              g.sync_lds();
              done = W.fsw[kCDONE]; sink = W.fsw[kCSINK];
              ++n_fs_bad;  // (diagnostics: counted with the refused steps)
              need_full = true;  // (the order the events were applied in is not the sorted one: the next step reads cols[] from memory)
            }
            cy_evser += MOT_FCLOCK() - q4;
            if (sink >= 0) { final_j = sink; return 2; }
            shi += static_cast<unsigned>(done);
            slo += static_cast<unsigned>(cnt);
            return 1;
          };
          auto fast_step = [&](double mind_b) -> int {
            g.lds_barriers(true);
            const int r = fast_step_body(mind_b);
            g.lds_barriers(false);
            return r;
          };
          // Void real rows (round 4, on-the-fly costs). Let H = hmax_dummy_row: a dummy row has been swept with h = H (or the search
          // started from one: its initial distances are that sweep with h = 0), so every TODO real column j has
          // d[j] <= D_j = fl(fl(half - v[j]) - H), and d only falls. A real row i with h_i relaxes column j to
          // cred = fl(fl(c_ij - v[j]) - h_i), monotone in c_ij, and c_ij >= rmin_i (a NaN cost compares false: no-op). Then
          //  (a) h_i <= H and rmin_i >= half: cred >= D_j >= d[j] by monotone rounding — fact (2) of the scan steps with an empty list;
          //  (b) otherwise, in exact arithmetic cred - D_j >= g = (rmin_i - half) - (h_i - H) (v[j] cancels) and the four roundings
          //      involved err by at most 4u(|rmin_i| + |half| + 2|v[j]| + |h_i| + |H|), u = 2^-53: with g above 2^-46 times that sum
          //      (max |v| over the real columns taken once per search: the duals do not change inside one) cred > d[j] strictly.
          // Either way `cred < d[j]` is false for every real column: the sweep changes nothing and is skipped. On a tracking problem
          // these are the unmatched tracks (costs 1 against half = thresh / 2, h = half - v[dummy column]): hundreds per tied set.
          double vabs_max = -1.0;
          auto void_cols = [&](int ri, double hi) {
            const double rmn = static_cast<double>(W.rmin[ri]);
            if (hi <= hmax_dummy_row) return rmn >= half;
            if (!(hmax_dummy_row > -1e299)) return false;
            const double gap = (rmn - half) - (hi - hmax_dummy_row);
            const double mag = (rmn < 0 ? -rmn : rmn) + (half < 0 ? -half : half) + 2.0 * vabs_max + (hi < 0 ? -hi : hi) +
                               (hmax_dummy_row < 0 ? -hmax_dummy_row : hmax_dummy_row);
            return gap > mag * 1.4210854715202004e-14;  // 2^-46
          };
          if (keep_rmin) {  // (uniform; one pass over the hot duals per search that gets here)
            double vm = 0.0;
            for (int j = t; j < nc; j += T) { const double a = W.v[j]; const double aa = a < 0 ? -a : a; if (aa > vm) vm = aa; }
            vabs_max = -g.reduce_min(-vm);
            if (!(vabs_max >= 0.0) || !(vabs_max < 1e299)) vabs_max = 1e300;  // (NaN / inf duals: rule (b) never fires)
          }
          const double mind_set = pq_d;  // every member of a SCAN set sits at the same distance
          bool pq_valid = true;
          while (slo != shi) {
            if (fast_ok) {
              const int fr = fast_step(mind_set);
              if (fr == 2) { returned = true; break; }
              if (fr == 1) { pq_valid = false; continue; }
            }
            if (!pq_valid) {
              if (stepped) g.sync();  // the steps' stores to cols[] are visible
              member(slo);
              pq_valid = true;
            }
            // Runs of dummy-row members whose sweep is void (h <= hmax_dummy_row, see below) leave the SCAN set together:
            // each lane classifies one member ahead, one reduction counts the leading void ones. With more detections
            // than tracks the tied sets are hundreds of such rows (one per unmatched detection).
            // (round 4) So do the real rows that sit on a dummy column (every unmatched track: h = half - v[j] - d needs no cost
            // evaluation) when their real columns are void (void_cols below) and, with h <= hmax_real_row, their dummy columns too.
            auto void_real = [&](int mi, int mj, double md) {
              if (!keep_rmin || mi >= nr || mj < nc) return false;
              const double hv = half - W.v[mj] - md;
              return hv <= hmax_real_row && void_cols(mi, hv);
            };
            if ((pq_i >= nr && (((pq_j < nc) ? half : 0.0) - W.v[pq_j] - pq_d) <= hmax_dummy_row) || void_real(pq_i, pq_j, pq_d)) {
              const unsigned idx = slo + static_cast<unsigned>(t);
              bool ok = false;
              if (idx < shi) {
                const int mj = W.cols[idx];
                const int mi = W.y[mj];
                if (mi >= nr) ok = (((mj < nc) ? half : 0.0) - W.v[mj] - W.d[mj]) <= hmax_dummy_row;
                else ok = void_real(mi, mj, W.d[mj]);
              }
              int cnt = g.reduce_min_int(ok ? kNoIdx : t);
              const int avail = (shi - slo < static_cast<unsigned>(T)) ? static_cast<int>(shi - slo) : T;
              if (cnt > avail) cnt = avail;
              if (pq_i < nr) MOT_LAP_DBG_VOID(1);
              slo += static_cast<unsigned>(cnt);  // cnt >= 1: lane 0 looked at the current member
              if (slo != shi) member(slo);
              continue;
            }
            ++n_seq_sweeps;
            const long long qs0 = MOT_FCLOCK();
            const int jq = pq_j;
            const int i = pq_i;
            const double mind = pq_d;
            const ExtRow<Cost> R = ext_row(C, P, i);
            ++slo;
            const bool fetched = slo != shi;
            if (fetched) member(slo);
            if constexpr (is_matrix_cost<Cost>::value) {
              // the next member's matrix row (the lane's first kSweep columns) is requested now and consumed one sweep later:
              // a row of a matrix tens of MB large misses L2, and the sweep below would otherwise wait for it load by load
              if (fetched && pq_i < nr && pq_i != pf_row && !(have_yc && rl_len(pq_i) <= kRlCap)) {  // (a row with a list is nearly always swept through it: no random line of the matrix for those)
                const float* rp = C.row(pq_i).p;
#pragma unroll
                for (int k = 0; k < kSweep; ++k) {
                  const int j = t + k * T;
                  pf_next[k] = (j < nc) ? gld(rp, j) : 0.f;
                }
                pf_next_h = (pq_j < nc) ? gld(rp, pq_j) : 0.f;  // the member's own element, head of the next sweep
                pf_next_row = pq_i;
                pf_next_col = pq_j;
              }
            }
            double h;
            if constexpr (is_matrix_cost<Cost>::value) {
              const bool have_h = pf_row == i && pf_col == jq && jq < nc;
              h = (have_h ? static_cast<double>(pf_h) : ((have_yc && i < nr && jq < nc) ? match_cost(i, jq) : R.at(C, jq))) - W.v[jq] - mind;
            } else {
              h = R.at(C, jq) - W.v[jq] - mind;
            }
            g.sync();
            // The relaxation sweep, by OWNER lanes (coalesced d[], conflict-free v[], register-cached boxes) rather than
            // by cols[] position; inv[] tells whether a column is still TODO and where it sits for the order-dependent
            // parts. Rows of one kind share their cost over part of the columns — dummy rows over all of them, real rows
            // over the dummy block — where cred = base(j) - v[j] - h; a row whose h does not exceed the largest h swept so
            // far cannot undercut what that sweep left (d only falls), so that part of its sweep is skipped. lapjv scans
            // every member of a tied set one by one, and the extension makes those sets hundreds of such rows long.
            bool sweep_real = true, sweep_dummy = true;
            if (i >= nr) {
              if (h <= hmax_dummy_row) { sweep_real = false; sweep_dummy = false; }
              else hmax_dummy_row = h;
            } else {
              if (h <= hmax_real_row) sweep_dummy = false;
              else hmax_real_row = h;
              // no entry below half and a dummy row swept with at least this h: nothing among the real columns can be lowered
              if (keep_rmin && void_cols(i, h)) { sweep_real = false; MOT_LAP_DBG_VOID(1); }
            }
            int first_sink = kNoIdx, any_tie = 0;
            auto relax_pre = [&](double red, int j, int k, double dj) {  // k = inv[j], dj = d[j]
              if (k < static_cast<int>(shi)) return;  // already SCAN/READY
              const double cred = red - h;
              if (cred < dj) {
                W.d[j] = cred;
                W.pred[j] = i;
                if (cred == mind) {
                  W.tie[k] = 1;
                  any_tie = 1;
                  if (W.y[j] < 0 && k < first_sink) first_sink = k;
                }
              }
            };
            auto relax = [&](double red, int j) { relax_pre(red, j, W.inv[j], W.d[j]); };
            if constexpr (is_matrix_cost<Cost>::value) {
              // kSweep of the lane's columns at a time: their cost, inv[] and d[] loads are all in flight before the first
              // relaxation (a lane owns its columns, so nothing it stores below aliases what it preloaded; inv[] does not
              // change during a sweep)
              auto sweep = [&](int jb, int je, auto load, auto value) {  // load: the float to fetch for column j; value: its cost
                for (int j0 = jb; j0 < je; j0 += kSweep * T) {
                  float cv[kSweep];
                  double dd[kSweep];
                  int kk[kSweep];
#pragma unroll
                  for (int k = 0; k < kSweep; ++k) {
                    const int j = j0 + k * T;
                    if (j < je) { cv[k] = load(j, k, j0 == jb); kk[k] = W.inv[j]; dd[k] = W.d[j]; }
                  }
#pragma unroll
                  for (int k = 0; k < kSweep; ++k) {
                    const int j = j0 + k * T;
                    if (j < je) relax_pre(value(cv[k]) - W.v[j], j, kk[k], dd[k]);
                  }
                }
              };
              auto no_load = [](int, int, bool) { return 0.f; };
              if (sweep_real && R.real && use_rl && h <= hmax_dummy_row && rl_len(i) <= kRlCap) {
                // fact (2) above: the columns at or above half cannot be lowered by this row — its list is the whole sweep
                const int ne = rl_len(i);
                for (int e = t; e < ne; e += T) {
                  const unsigned long long ent = W.rl_ent[static_cast<long>(i) * kRlCap + e];
                  const int j = rl_col_of(ent);
                  relax_pre(static_cast<double>(rl_cost_of(ent)) - W.v[j], j, W.inv[j], W.d[j]);
                }
              } else if (sweep_real) {
                if (R.real) {
                  const float* rp = R.r.p;
                  const bool have = pf_row == i;
                  sweep(t, nc, [&](int j, int k, bool first) { return (first && have) ? pf[k] : gld(rp, j); },
                        [](float c) { return static_cast<double>(c); });
                } else {
                  const double l = R.left;
                  sweep(t, nc, no_load, [l](float) { return l; });
                }
              }
              if (sweep_dummy) {
                const double r = R.right;
                sweep(first_dummy(t, T, nc), n, no_load, [r](float) { return r; });
              }
              if (pf_next_row >= 0) {  // rotate the prefetched row in for the next sweep
#pragma unroll
                for (int k = 0; k < kSweep; ++k) pf[k] = pf_next[k];
                pf_h = pf_next_h;
                pf_row = pf_next_row;
                pf_col = pf_next_col;
                pf_next_row = -1;
              }
            } else {
              if (sweep_real) for_lane_real(C, R, W.v, t, T, nc, relax);
              if (sweep_dummy) for_lane_dummy(R.right, W.v, t, T, nc, n, relax);
            }
            // one reduction for both outcomes: the first sink position, or "ties but no sink", or nothing
            const int key = g.reduce_min_int((first_sink != kNoIdx) ? first_sink : (any_tie ? kNoIdx - 1 : kNoIdx));
            if (key < kNoIdx - 1) {
              final_j = W.cols[key];
              returned = true;
              g.sync();
              break;
            }
            const int base = static_cast<int>(shi);
            int nt = 0;
            if (key == kNoIdx - 1) {  // ties are rare on real-valued costs: skip the ordered compaction without them
              g.sync();
              nt = compact_ascending(g, n - base, [&](int q) { return W.tie[base + q] != 0; }, W.lst);
              if (t == 0) {
                for (int q = 0; q < nt; ++q) {  // ties join the SCAN set in ascending k (:146-147)
                  const int k = base + W.lst[q];
                  W.tie[k] = 0;
                  const int j = W.cols[k];
                  const int js = W.cols[shi + q];
                  W.cols[k] = js; W.inv[js] = k;
                  W.cols[shi + q] = j; W.inv[j] = static_cast<int>(shi) + q;
                  if (use_rl) W.fsw[kFsTodo + (j >> 5)] = static_cast<int>(W.fsw[kFsTodo + (j >> 5)]) & ~(1 << (j & 31));
                }
              }
            }
            shi += static_cast<unsigned>(nt);
            g.sync();
            cy_seq += MOT_FCLOCK() - qs0;
            if (!fetched && slo != shi) member(slo);  // the SCAN set was empty until this sweep's ties joined it
          }
          if (stepped) g.sync();  // the steps' stores to cols[] / inv[] / pred[] are visible to what follows
          if (!returned) { lo = slo; hi = shi; }
        }
      }
      {
        const double mind = W.d[W.cols[lo]];
        g.sync();
        for (unsigned k = t; k < n_ready; k += T) {
          const int j = W.cols[k];
          W.v[j] += W.d[j] - mind;
        }
      }
      g.sync();
      if (t == 0) {  // augment along pred (:202-207)
        int i = -1, j = final_j;
        while (i != start) {
          i = W.pred[j];
          W.y[j] = i;
          if (have_yc && i < nr && j < nc) W.ycost[j] = static_cast<float>(C.at(i, j));
          const int nx = W.x[i];
          W.x[i] = j;
          j = nx;
        }
      }
      g.sync();
    } else {
      // single-step path: n_ready == 0 so no dual changes (:183-189); y[final_j] = start, x[start] = final_j
      // written by the owner lane of final_j: y[final_j] is next read by that same lane (or after
      // a barrier); x[] is only read again by lane 0 behind the general path's barriers.
      if ((final_j % T) == t) {
        W.y[final_j] = start; W.x[start] = final_j;
        if (have_yc && start < nr && final_j < nc) W.ycost[final_j] = static_cast<float>(R0.at(C, final_j));
      }
    }
  }
  if (W.cyc && t == 0) {
    const long long c4 = MOT_CLOCK();
    W.cyc[0] = c1 - c0; W.cyc[1] = c2 - c1; W.cyc[2] = c3 - c2; W.cyc[3] = c4 - c3;
    W.cyc[4] = n_uniq; W.cyc[5] = n_carr; W.cyc[6] = n_paths; W.cyc[7] = n;
    W.cyc[8] = n_fs_steps; W.cyc[9] = n_fs_members; W.cyc[10] = n_fs_sparse; W.cyc[11] = n_fs_events; W.cyc[12] = n_seq_sweeps; W.cyc[13] = n_fs_bad;
    W.cyc[14] = n_finds; W.cyc[15] = use_rl ? 1 : 0;
    if (W.cyc_ext) { for (int k = 0; k < 12; ++k) W.cyc[24 + k] = cy_sub[k]; W.cyc[16] = cy_cls; W.cyc[17] = cy_dry; W.cyc[18] = cy_apply; W.cyc[19] = cy_evsort; W.cyc[20] = cy_evser; W.cyc[21] = cy_find; W.cyc[22] = cy_seq; W.cyc[23] = cy_init; }
  }
}

}  // namespace mot
