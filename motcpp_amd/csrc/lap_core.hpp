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
#include "grp.hpp"

namespace mot {

struct LapProblem {
  const float* cost;  // nr x nc, row-major, leading dimension ld
  int ld, nr, nc;
  double half;        // thresh / 2 (lap_solver.hpp:300)
};
// Workspace, each array n = nr + nc long. May live in LDS or in global memory.
struct LapWork {
  double* v;   // column duals
  double* d;   // shortest-path distances
  int* x;      // row -> col (extended)
  int* y;      // col -> row (extended)
  int* fr;     // free-row list
  int* pred;   // path predecessors / column-hit counters in phase 1
  int* cols;   // lapjv's column permutation / scratch list
  int* tmp;    // tie flags (slow path)
  int* lst;    // compacted tie positions (slow path)
};
MOT_HD size_t lap_work_bytes(int n) { return static_cast<size_t>(n) * (2 * sizeof(double) + 7 * sizeof(int)); }
MOT_HD LapWork lap_carve(void* base, int n) {
  LapWork w;
  char* p = static_cast<char*>(base);
  w.v = reinterpret_cast<double*>(p); p += sizeof(double) * n;
  w.d = reinterpret_cast<double*>(p); p += sizeof(double) * n;
  w.x = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.y = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.fr = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.pred = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.cols = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.tmp = reinterpret_cast<int*>(p); p += sizeof(int) * n;
  w.lst = reinterpret_cast<int*>(p);
  return w;
}

// Row view of the extended matrix (lap_solver.hpp:303-315): real rows are [cost | half...],
// dummy rows are [half... | 0...].
struct ExtRow {
  const float* p;
  double left, right;  // constants used when p == nullptr (left: j < nc) / for j >= nc
  int nc;
  MOT_DEV double at(int j) const {
    if (j < nc) return p ? static_cast<double>(p[j]) : left;
    return right;
  }
};
MOT_DEV ExtRow ext_row(const LapProblem& P, int i) {
  ExtRow r;
  r.nc = P.nc;
  if (i < P.nr) { r.p = P.cost + static_cast<size_t>(i) * P.ld; r.left = 0.0; r.right = P.half; }
  else { r.p = nullptr; r.left = P.half; r.right = 0.0; }
  return r;
}

// Compacts {i in [0,n) : flag(i)} in ascending order into out[]; returns the count (uniform).
template <class G, class F>
MOT_DEV int compact_ascending(G& g, int n, F flag, int* out) {
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
template <class G>
MOT_DEV void lap_solve(G& g, const LapProblem& P, const LapWork& W) {
  const int T = g.size(), t = g.tid();
  const int nr = P.nr, nc = P.nc, n = nr + nc;
  const double half = P.half;

  // ---- phase 1: column reduction + reduction transfer (_ccrrt_dense, :36-72) ----
  for (int i = t; i < n; i += T) { W.x[i] = -1; W.pred[i] = 0; }
  g.sync();
  for (int j = t; j < n; j += T) {
    double vm = kLapLarge;
    int im = 0;
    if (j < nc) {
      const float* cp = P.cost + j;
      for (int i = 0; i < nr; ++i) {
        const double c = static_cast<double>(cp[static_cast<size_t>(i) * P.ld]);
        if (c < vm) { vm = c; im = i; }
      }
      if (half < vm) { vm = half; im = nr; }  // rows nr.. are all `half`: only the first can win
    } else {
      if (half < vm) { vm = half; im = 0; }   // rows 0..nr-1 are all `half`
      if (0.0 < vm) { vm = 0.0; im = nr; }    // rows nr.. are all 0
    }
    W.v[j] = vm;
    W.y[j] = im;
    G::atomic_max(&W.x[im], j);   // x[i] = largest column whose minimum sits in row i (:47-55)
    G::atomic_add(&W.pred[im], 1);
  }
  g.sync();
  for (int j = t; j < n; j += T)
    if (W.x[W.y[j]] != j) W.y[j] = -1;
  g.sync();
  // rows that own exactly one column get their dual tightened, in ascending row order (:57-69)
  const int n_uniq = compact_ascending(g, n, [&](int i) { return W.x[i] >= 0 && W.pred[i] == 1; }, W.cols);
  int nfree = compact_ascending(g, n, [&](int i) { return W.x[i] < 0; }, W.fr);
  for (int u = 0; u < n_uniq; ++u) {
    const int i = W.cols[u];
    const int j = W.x[i];
    const ExtRow R = ext_row(P, i);
    double mn = kLapLarge;
    for (int j2 = t; j2 < n; j2 += T) {
      if (j2 == j) continue;
      const double c = R.at(j2) - W.v[j2];
      if (c < mn) mn = c;
    }
    mn = g.reduce_min(mn);
    if ((j % T) == t) W.v[j] -= mn;  // owner lane: next reader of v[j] is this same lane
  }
  g.sync();

  // ---- phase 2: augmenting row reduction, twice (_carr_dense, :74-113, :221-224) ----
  for (int pass = 0; pass < 2 && nfree > 0; ++pass) {
    unsigned current = 0, rr_cnt = 0;
    int new_free = 0;
    int forwarded = -1;  // row re-queued by free_rows[--current] = i0, consumed next iteration
    while (current < static_cast<unsigned>(nfree)) {
      ++rr_cnt;
      const int fi = (forwarded >= 0) ? forwarded : W.fr[current];
      forwarded = -1;
      ++current;
      const ExtRow R = ext_row(P, fi);
      Top2 tt = top2_empty();
      for (int j = t; j < n; j += T) top2_push(tt, R.at(j) - W.v[j], j);
      tt = g.reduce_top2(tt);
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
        if (lowers) { if ((j1 % T) == t) W.v[j1] = v1_new; }
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
      // no barrier here: v[j1]/y[j1] are next read (a) by their owner lane in the strided
      // loop above, or (b) by everyone only after the reduce_top2 barrier of the next round.
    }
    g.sync();
    nfree = new_free;
  }

  // ---- phase 3: augmentation (_ca_dense, :195-211) ----
  for (int f = 0; f < nfree; ++f) {
    const int start = W.fr[f];
    const ExtRow R0 = ext_row(P, start);
    double mn = 1e300;
    for (int j = t; j < n; j += T) {
      const double dj = R0.at(j) - W.v[j];
      if (dj < mn) mn = dj;
    }
    const double g0 = g.reduce_min(mn);
    // First _find_dense from cols = identity leaves the tied-minimum columns in ascending
    // order in [0,hi) and the sink test (:174-177) keeps the LAST free one.
    int cand = -1;
    for (int j = t; j < n; j += T)
      if ((R0.at(j) - W.v[j]) == g0 && W.y[j] < 0 && j > cand) cand = j;
    int final_j = g.reduce_max(cand);
    if (final_j < 0) {
      // ---- general path: exact emulation of find_path_dense (:157-193) ----
      for (int j = t; j < n; j += T) { W.cols[j] = j; W.pred[j] = start; W.d[j] = R0.at(j) - W.v[j]; }
      g.sync();
      unsigned lo = 0, hi = 0, n_ready = 0;
      while (final_j == -1) {
        if (lo == hi) {
          n_ready = lo;
          if (t == 0) {  // _find_dense (:115-127), sequential: order of cols[] matters
            unsigned h2 = lo + 1;
            double mind = W.d[W.cols[lo]];
            for (unsigned k = h2; k < static_cast<unsigned>(n); ++k) {
              const int j = W.cols[k];
              const double dj = W.d[j];
              if (dj <= mind) {
                if (dj < mind) { h2 = lo; mind = dj; }
                W.cols[k] = W.cols[h2];
                W.cols[h2++] = j;
              }
            }
            int fj = -1;
            for (unsigned k = lo; k < h2; ++k) {
              const int j = W.cols[k];
              if (W.y[j] < 0) fj = j;
            }
            W.tmp[0] = static_cast<int>(h2);
            W.tmp[1] = fj;
          }
          g.sync();
          hi = static_cast<unsigned>(W.tmp[0]);
          final_j = W.tmp[1];
          g.sync();
        }
        if (final_j == -1) {
          // _scan_dense (:129-155) on local copies of lo/hi, written back only on normal exit
          unsigned slo = lo, shi = hi;
          bool returned = false;
          while (slo != shi) {
            const int jq = W.cols[slo++];
            const int i = W.y[jq];
            const double mind = W.d[jq];
            const ExtRow R = ext_row(P, i);
            const double h = R.at(jq) - W.v[jq] - mind;
            g.sync();
            int first_sink = kNoIdx;
            for (int k = static_cast<int>(shi) + t; k < n; k += T) {
              const int j = W.cols[k];
              const double cred = R.at(j) - W.v[j] - h;
              int flag = 0;
              if (cred < W.d[j]) {
                W.d[j] = cred;
                W.pred[j] = i;
                if (cred == mind) {
                  flag = 1;
                  if (W.y[j] < 0 && k < first_sink) first_sink = k;
                }
              }
              W.tmp[k] = flag;
            }
            first_sink = g.reduce_min_int(first_sink);
            if (first_sink != kNoIdx) {
              final_j = W.cols[first_sink];
              returned = true;
              g.sync();
              break;
            }
            const int base = static_cast<int>(shi);
            const int nt = compact_ascending(g, n - base, [&](int q) { return W.tmp[base + q] != 0; }, W.lst);
            if (t == 0) {
              for (int q = 0; q < nt; ++q) {  // ties join the SCAN set in ascending k (:146-147)
                const int k = base + W.lst[q];
                const int j = W.cols[k];
                W.cols[k] = W.cols[shi + q];
                W.cols[shi + q] = j;
              }
            }
            shi += static_cast<unsigned>(nt);
            g.sync();
          }
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
      if ((final_j % T) == t) { W.y[final_j] = start; W.x[start] = final_j; }
    }
  }
}

}  // namespace mot
