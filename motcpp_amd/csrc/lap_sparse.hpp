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
//      is; materialised matrix: one coalesced sweep);
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
#pragma once
#include "cost_math.hpp"
#include "grp.hpp"
#include "lap_core.hpp"
#include "mem.hpp"

namespace mot {

constexpr int kSpK = 8;            // viable pairs kept per column (more: fall back)
constexpr int kSpBuckets = 256;    // x1 buckets of the row boxes
constexpr int kSpSlots = 64;       // rows one path search may reach (more: fall back)
constexpr int kSpArcs = 256;       // eps-tight non-matching pairs the certificate may hold (more: fall back)
constexpr double kSpEps = 1e-9;    // tie margin
constexpr double kSpTol = 1e-11;   // tolerated violation of dual feasibility / complementary slackness (fp64 rounding)
constexpr int kSpIntMax = 0x7fffffff;
constexpr float kSpHuge = 1.0e30f; // boxes / costs beyond this magnitude are left to the exact path

// Workspace. HS = address space of the hot arrays (LDS when the problem fits, else global scratch).
template <int HS>
struct SparseWorkT {
  MemPtr<double, HS> u, v;       // row / column duals
  MemPtr<int, HS> x, y;          // row -> column, column -> row (-1: unmatched)
  MemPtr<int, HS> slot;          // per row: 1 + search slot during a path search; flag bits during the certificate
  MemPtr<double, HS> sdist;      // search slots [kSpSlots]: distance label,
  MemPtr<int, HS> srow, spred, sstate;  // row, column it was reached from, 1 reached / 2 scanned
  MemPtr<int, HS> bstart, bcur, bmax;   // x1 buckets: [B+1] start, [B] fill cursor, [B] prefix maximum of the x2 keys
  MemPtr<int, HS> ctr;           // [4] counters (arc count, flags)
  MemPtr<int, kMemGlobal> erow;    // [nc][kSpK] viable pairs of a column: row (-1 ends the list) ...
  MemPtr<float, kMemGlobal> ecost; // ... and cost
  MemPtr<int, kMemGlobal> freel;   // [nc] columns still to insert
  MemPtr<float, kMemGlobal> sbox;  // [nr][4] row boxes in bucket order
  MemPtr<int, kMemGlobal> sidx;    // [nr] their row indices
  MemPtr<int, kMemGlobal> arcs;    // [kSpArcs][2] eps-tight pair: (row, owner of its column)
};
MOT_HD size_t sparse_hot_bytes(int nr, int nc) {
  return static_cast<size_t>(nr) * 16 + static_cast<size_t>(nc) * 12 + kSpSlots * 20 + (3 * kSpBuckets + 1 + 4) * 4 + 16;
}
MOT_HD size_t sparse_cold_bytes(int nr, int nc) {
  return static_cast<size_t>(nc) * (8 * kSpK + 4) + static_cast<size_t>(nr) * 20 + kSpArcs * 8 + 64;
}
// Scratch of one task (mot_lap_work_bytes): the exact solver's hot + cold arrays and staged boxes, or — they are never live
// at the same time — the fast path's lists; the last 16 bytes hold the task's status word (1: finished by the fast path).
MOT_HD size_t lap_task_scratch_bytes(int n, int m) {
  const size_t rn = n > 0 ? n : 0, rm = m > 0 ? m : 0, nm = rn + rm;
  return ((lap_hot_bytes(static_cast<int>(nm)) + 15) & ~size_t(15)) + ((lap_cold_bytes(static_cast<int>(nm)) + 15) & ~size_t(15)) + 4 * (5 * rn + 6 * rm) + 256 + 8192;
}
template <class W>
MOT_HD void sparse_carve_hot(W& w, void* base, int nr, int nc) {
  char* p = static_cast<char*>(base);
  w.u.p = reinterpret_cast<double*>(p); p += 8 * static_cast<size_t>(nr);
  w.v.p = reinterpret_cast<double*>(p); p += 8 * static_cast<size_t>(nc);
  w.sdist.p = reinterpret_cast<double*>(p); p += 8 * kSpSlots;
  w.x.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(nr);
  w.slot.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(nr);
  w.y.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(nc);
  w.srow.p = reinterpret_cast<int*>(p); p += 4 * kSpSlots;
  w.spred.p = reinterpret_cast<int*>(p); p += 4 * kSpSlots;
  w.sstate.p = reinterpret_cast<int*>(p); p += 4 * kSpSlots;
  w.bstart.p = reinterpret_cast<int*>(p); p += 4 * (kSpBuckets + 1);
  w.bcur.p = reinterpret_cast<int*>(p); p += 4 * kSpBuckets;
  w.bmax.p = reinterpret_cast<int*>(p); p += 4 * kSpBuckets;
  w.ctr.p = reinterpret_cast<int*>(p);
}
template <class W>
MOT_HD void sparse_carve_cold(W& w, void* base, int nr, int nc) {
  char* p = static_cast<char*>(base);
  w.sbox.p = reinterpret_cast<float*>(p); p += 16 * static_cast<size_t>(nr);
  w.sidx.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(nr);
  w.erow.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(kSpK) * nc;
  w.ecost.p = reinterpret_cast<float*>(p); p += 4 * static_cast<size_t>(kSpK) * nc;
  w.freel.p = reinterpret_cast<int*>(p); p += 4 * static_cast<size_t>(nc);
  w.arcs.p = reinterpret_cast<int*>(p);
}

// Outcome of the enumeration of viable pairs.
struct SparseEnum {
  int ok;          // 0: something the fast path does not decide was seen (tie with the threshold, NaN, overflow ...)
  double mincost;  // minimum cost over ALL pairs (MOT_LAP_GATE_MIN)
};

// ---- 1a. viable pairs from boxes ---------------------------------------------------------------------------------
// Box source of one problem: planes [4][ld] with an optional gather index, as in mot_iou_task.
struct SparseBoxes {
  const float* p; int ld; const int* idx;
  MOT_DEV void load(int i, float b[4]) const {
    const int gi = idx ? gld(idx, i) : i;
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = gld(p, static_cast<size_t>(k) * ld + gi);
  }
};
MOT_DEV bool sp_finite4(const float b[4]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) ok = ok && (b[k] > -kSpHuge) && (b[k] < kSpHuge);  // false for NaN
  return ok;
}
// EvalFn(row index, row box, row area, column box, column area, column confidence, column index) -> float cost with the
// cost kernel's arithmetic; zc(conf) = cost of a pair that does not intersect.
template <class G, class W, class EvalFn, class ZeroFn>
MOT_DEV SparseEnum sparse_enumerate_boxes(G& g, const W& w, int nr, int nc, const SparseBoxes& A, const SparseBoxes& Bx,
                                          const float* bconf, const int* bidx, float thresh, EvalFn eval, ZeroFn zero_cost) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  int bad = 0;
  // ---- rows into x1 buckets ----
  float xlo = 3.0e38f, xhi = -3.0e38f;
  for (int i = t; i < nr; i += T) {
    float a[4];
    A.load(i, a);
    if (!sp_finite4(a)) bad = 1;
    if (a[0] < xlo) xlo = a[0];
    if (a[0] > xhi) xhi = a[0];
  }
  for (int b = t; b <= kSpBuckets; b += T) w.bstart[b] = 0;
  for (int b = t; b < kSpBuckets; b += T) w.bmax[b] = static_cast<int>(0x80000000u);
  xlo = static_cast<float>(g.reduce_min(static_cast<double>(xlo)));
  xhi = -static_cast<float>(g.reduce_min(-static_cast<double>(xhi)));
  bad = g.reduce_max(bad);
  if (bad) return SparseEnum{0, 0.0};
  const float span = xhi - xlo;
  const float scale = (span > 0.0f) ? static_cast<float>(kSpBuckets) / span : 0.0f;
  // monotone non-decreasing in x (so a0 < b2 implies bucket(a0) <= bucket(b2)); any x maps into [0, B)
  auto bucket = [&](float x) {
    const float f = (x - xlo) * scale;
    int b = (f > 0.0f) ? ((f < static_cast<float>(kSpBuckets)) ? static_cast<int>(f) : kSpBuckets - 1) : 0;
    return (b < kSpBuckets) ? b : kSpBuckets - 1;
  };
  g.sync();
  for (int i = t; i < nr; i += T) {
    float a[4];
    A.load(i, a);
    G::atomic_add(w.bstart.raw(bucket(a[0]) + 1), 1);
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
  for (int i = t; i < nr; i += T) {
    float a[4];
    A.load(i, a);
    const int b = bucket(a[0]);
    const int pos = G::atomic_add(w.bcur.raw(b), 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) w.sbox[4 * static_cast<size_t>(pos) + k] = a[k];
    w.sidx[pos] = i;
    G::atomic_max(w.bmax.raw(b), f32_key(a[2]));
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
  double mn = 1e300;
  for (int j = t; j < nc; j += T) {
    float b[4];
    Bx.load(j, b);
    const int gj = bidx ? gld(bidx, j) : j;
    const float conf = bconf ? gld(bconf, gj) : 0.0f;
    int ne = 0;
    if (!sp_finite4(b) || !(conf > -kSpHuge && conf < kSpHuge)) bad = 1;
    const float barea = (b[2] - b[0]) * (b[3] - b[1]);
    const float zc = zero_cost(conf);
    if (!(static_cast<double>(zc) > th + kSpEps)) bad = 1;  // a non-intersecting pair would be viable (or zc is NaN)
    int ninter = 0;
    if (!bad) {
      const int bhi = bucket(b[2]);
      const int kb0 = f32_key(b[0]);
      int lo = 0, hi = bhi + 1;  // first bucket whose prefix maximum of x2 exceeds b0
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (static_cast<int>(w.bmax[mid]) > kb0) hi = mid; else lo = mid + 1; }
      const int ps = w.bstart[lo], pe = w.bstart[bhi + 1];
      for (int p = ps; p < pe; ++p) {
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = w.sbox[4 * static_cast<size_t>(p) + k];
        // w > 0 and h > 0 of iou_pair (finite boxes): anything else has inter == 0 and costs zc
        if (!(a[0] < b[2] && a[2] > b[0] && a[1] < b[3] && a[3] > b[1])) continue;
        const int i = w.sidx[p];
        const float aarea = (a[2] - a[0]) * (a[3] - a[1]);
        const float c = eval(i, a, aarea, b, barea, conf, j);
        ++ninter;
        if (!(c > -kSpHuge && c < kSpHuge)) { bad = 1; break; }
        const double cd = static_cast<double>(c);
        if (cd < mn) mn = cd;
        if (cd < th - kSpEps) {
          if (ne < kSpK) { w.erow[static_cast<size_t>(j) * kSpK + ne] = i; w.ecost[static_cast<size_t>(j) * kSpK + ne] = c; ++ne; }
          else { bad = 1; break; }
        } else if (!(cd > th + kSpEps)) { bad = 1; break; }  // tie with the threshold
      }
    }
    if (ninter < nr && static_cast<double>(zc) < mn) mn = static_cast<double>(zc);
    if (ne < kSpK) w.erow[static_cast<size_t>(j) * kSpK + ne] = -1;
  }
  bad = g.reduce_max(bad);
  mn = g.reduce_min(mn);
  g.sync();
  return SparseEnum{bad ? 0 : 1, mn};
}

// ---- 1b. viable pairs from a materialised matrix (row-major, lane-owned columns: coalesced) -------------------------
template <class G, class W>
MOT_DEV SparseEnum sparse_enumerate_matrix(G& g, const W& w, int nr, int nc, const float* cost, int ld, float thresh) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  int bad = 0;
  double mn = 1e300;
  for (int j = t; j < nc; j += T) {
    int ne = 0;
    for (int i = 0; i < nr && !bad; ++i) {
      const float c = gld(cost, static_cast<size_t>(i) * ld + j);
      if (!(c > -kSpHuge && c < kSpHuge)) { bad = 1; break; }
      const double cd = static_cast<double>(c);
      if (cd < mn) mn = cd;
      if (cd < th - kSpEps) {
        if (ne < kSpK) { w.erow[static_cast<size_t>(j) * kSpK + ne] = i; w.ecost[static_cast<size_t>(j) * kSpK + ne] = c; ++ne; }
        else bad = 1;
      } else if (!(cd > th + kSpEps)) bad = 1;
    }
    if (ne < kSpK) w.erow[static_cast<size_t>(j) * kSpK + ne] = -1;
  }
  bad = g.reduce_max(bad);
  mn = g.reduce_min(mn);
  g.sync();
  return SparseEnum{bad ? 0 : 1, mn};
}

// ---- 2-4. matching over the viable pairs + certificate ---------------------------------------------------------------
// Returns 1 with w.x / w.y holding THE minimum-weight matching (unique by more than kSpEps); else a reason <= 0 for the
// exact path to take over (-2 a search reached too many rows, -3 certificate arithmetic, -4 too many tight pairs, -5 not unique).
template <class G, class W>
MOT_DEV int sparse_solve(G& g, const W& w, int nr, int nc, float thresh) {
  const int T = g.size(), t = g.tid();
  const double th = static_cast<double>(thresh);
  for (int i = t; i < nr; i += T) { w.u[i] = 0.0; w.x[i] = kSpIntMax; w.slot[i] = 0; }
  g.sync();
  // column duals and proposals
  for (int j = t; j < nc; j += T) {
    float bc = 0.0f;
    int br = -1;
    for (int k = 0; k < kSpK; ++k) {
      const int r = w.erow[static_cast<size_t>(j) * kSpK + k];
      if (r < 0) break;
      const float c = w.ecost[static_cast<size_t>(j) * kSpK + k];
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
    int lose = 0;
    if (j < nc) {
      const int br = w.y[j];
      if (br >= 0 && static_cast<int>(w.x[br]) != j) { lose = 1; w.y[j] = -1; }
    }
    int tot;
    const int pos = g.exclusive_scan(lose, &tot);
    if (lose) w.freel[nfree + pos] = j;
    nfree += tot;
  }
  g.sync();
  for (int i = t; i < nr; i += T)
    if (static_cast<int>(w.x[i]) == kSpIntMax) w.x[i] = -1;
  g.sync();

  // ---- shortest augmenting path per remaining column ----
  for (int f = 0; f < nfree; ++f) {
    const int j0 = w.freel[f];
    double L = -static_cast<double>(w.v[j0]);  // leave j0 unmatched
    int term_row = -1, term_col = j0;           // terminal: a free row, or the column that ends unmatched
    int cur = j0;
    double D = 0.0;
    int nslots = 0;
    bool overflow = false;
    for (;;) {
      // relax the viable pairs of column `cur`
      {
        const double vc = w.v[cur];
        int need = 0, er = -1;
        double nd = 0.0;
        if (t < kSpK) {
          er = w.erow[static_cast<size_t>(cur) * kSpK + t];
          // (entries after the -1 terminator are stale: a lane is valid only if every earlier entry is)
        }
        // validity prefix: the list ends at the first negative row
        int firstneg = g.reduce_min_int((t < kSpK && er < 0) ? t : kSpIntMax);
        const bool valid = t < kSpK && t < firstneg;
        if (valid) {
          const double red = (static_cast<double>(static_cast<float>(w.ecost[static_cast<size_t>(cur) * kSpK + t])) - th) -
                             static_cast<double>(w.u[er]) - vc;
          nd = D + red;
          const int s = w.slot[er];
          if (s == 0) need = 1;
          else if (static_cast<int>(w.sstate[s - 1]) == 1 && nd < static_cast<double>(w.sdist[s - 1])) { w.sdist[s - 1] = nd; w.spred[s - 1] = cur; }
        }
        int tot;
        const int pos = g.exclusive_scan(need, &tot);
        if (nslots + tot > kSpSlots) { overflow = true; break; }
        if (need) {
          const int q = nslots + pos;
          w.srow[q] = er; w.sdist[q] = nd; w.spred[q] = cur; w.sstate[q] = 1; w.slot[er] = q + 1;
        }
        nslots += tot;
      }
      g.sync();
      // nearest reached, not yet scanned row (ties: lowest row index)
      Top2 tt = top2_empty();
      for (int q = t; q < nslots; q += T)
        if (static_cast<int>(w.sstate[q]) == 1) top2_push(tt, static_cast<double>(w.sdist[q]), static_cast<int>(w.srow[q]));
      tt = g.reduce_top2(tt);
      if (tt.j1 == kNoIdx || !(tt.v1 < L)) break;
      const int row = tt.j1;
      const int q = static_cast<int>(w.slot[row]) - 1;
      const int xc = w.x[row];
      if (xc < 0) { L = tt.v1; term_row = row; term_col = -1; break; }
      g.sync();
      if (t == 0) w.sstate[q] = 2;
      cur = xc;
      D = tt.v1;
      const double cand = D - static_cast<double>(w.v[cur]);
      if (cand < L) { L = cand; term_row = -1; term_col = cur; }
      g.sync();
    }
    if (overflow) return -2;
    g.sync();
    // duals: scanned rows and their columns move by (L - label); the source by L
    for (int q = t; q < nslots; q += T)
      if (static_cast<int>(w.sstate[q]) == 2) {
        const int r = w.srow[q];
        const double dl = L - static_cast<double>(w.sdist[q]);
        w.u[r] -= dl;
        w.v[static_cast<int>(w.x[r])] += dl;
      }
    if (t == 0) w.v[j0] += L;
    g.sync();
    // augment (one lane walks the path)
    if (t == 0) {
      int r = -1;
      if (term_row >= 0) r = term_row;
      else if (term_col != j0) { r = w.y[term_col]; w.y[term_col] = -1; }
      while (r >= 0) {
        const int q = static_cast<int>(w.slot[r]) - 1;
        const int p = w.spred[q];
        const int rn = w.y[p];
        w.y[p] = r;
        w.x[r] = p;
        if (p == j0) break;
        r = rn;
      }
    }
    g.sync();
    for (int q = t; q < nslots; q += T) w.slot[static_cast<int>(w.srow[q])] = 0;
    g.sync();
  }

  // ---- certificate ----
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
  for (int j = t; j < nc; j += T) {
    const double vj = w.v[j];
    const int yj = w.y[j];
    if (yj < 0 && !(vj >= -kSpTol && vj <= kSpTol)) bad = 1;  // an unmatched column carries no dual
    if (!(vj <= kSpTol)) bad = 1;
    for (int k = 0; k < kSpK; ++k) {
      const int r = w.erow[static_cast<size_t>(j) * kSpK + k];
      if (r < 0) break;
      const double red = (static_cast<double>(static_cast<float>(w.ecost[static_cast<size_t>(j) * kSpK + k])) - th) - static_cast<double>(w.u[r]) - vj;
      if (r == yj) { if (!(red >= -kSpTol && red <= kSpTol)) bad = 1; continue; }
      if (!(red >= -kSpTol)) { bad = 1; continue; }
      if (red <= kSpEps) {
        if (yj < 0) G::atomic_or(w.slot.raw(r), kFreeCol | kEnd);
        else {
          const int a = G::atomic_add(w.ctr.raw(0), 1);
          if (a < kSpArcs) { w.arcs[2 * a] = r; w.arcs[2 * a + 1] = yj; }
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
  if (narcs > kSpArcs) return -4;
  int nonuniq = 0;
  for (int i = t; i < nr; i += T) {
    const int fl = w.slot[i];
    if ((fl & kStart) && (fl & kFreeCol)) nonuniq = 1;  // changes its column for free
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
      if ((fl & kArrived) && (fl & kEnd)) nonuniq = 1;
    }
    // alternating cycle: peel arcs whose source has no live incoming arc; anything left lies on or behind a cycle
    int alive_n = narcs;
    // arcs[2a] >= 0 marks a live arc (dead: source stored as -1 - source)
    for (int it = 0; it <= narcs && alive_n > 0; ++it) {
      for (int a = t; a < narcs; a += T) {
        const int d = w.arcs[2 * a + 1];
        const int s = w.arcs[2 * a];
        (void)s;
        G::atomic_and(w.slot.raw(d), ~kIn);
      }
      for (int a = t; a < narcs; a += T) {  // (sources may not be destinations of any arc: clear theirs too)
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
    if (alive_n > 0) nonuniq = 1;
  }
  nonuniq = g.reduce_max(nonuniq);
  g.sync();
  return nonuniq ? -5 : 1;
}

}  // namespace mot
