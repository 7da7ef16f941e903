// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed by the product
// (motcpp_amd/, include/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it, and only as the checker / timed CPU baseline.
//
// CPU restatement (scalar C++17, fp32 state, fp64 assignment, no FMA contraction:
// build with -O2 -ffp-contract=off, mirroring /root/reference/CMakeLists.txt:231-236)
// of motcpp's association primitives:
//   box conversions     include/motcpp/utils/ops.hpp:15-114,188-211
//   iou_batch           include/motcpp/utils/iou.hpp:63-100
//   iou_distance        src/utils/matching.cpp:62-65, include/motcpp/utils/matching.hpp:128-182
//   fuse_score          src/utils/matching.cpp:130-143
//   embedding_distance  src/utils/matching.cpp:67-92 (cosine)
//   lapjv + extension   include/motcpp/association/lap_solver.hpp:36-332
//   linear_assignment   src/utils/matching.cpp:14-60
// Parity status: LAP/IoU pinned by the reference's own known answers
// (tests/test_matching.cpp, tests/test_iou.cpp — see tests/test_oracle_known_answers.py).
// The reference LAP header cannot be compiled here (needs Eigen, absent; no stand-ins allowed).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

// Dense row-major float matrix (the reference uses column-major Eigen::MatrixXf; the
// element values, not the storage order, are what parity is defined on).
struct Mat {
  int r = 0, c = 0;
  std::vector<float> a;
  Mat() = default;
  Mat(int rows, int cols, float fill = 0.0f) : r(rows), c(cols), a(static_cast<size_t>(rows) * cols, fill) {}
  float& operator()(int i, int j) { return a[static_cast<size_t>(i) * c + j]; }
  float operator()(int i, int j) const { return a[static_cast<size_t>(i) * c + j]; }
  int size() const { return r * c; }
};

using Box = std::array<float, 4>;

// ---- ops.hpp -------------------------------------------------------------------------
inline Box xyxy2xywh(const Box& b) {  // ops.hpp:15-22
  float w = b[2] - b[0];
  float h = b[3] - b[1];
  float xc = b[0] + w * 0.5f;
  float yc = b[1] + h * 0.5f;
  return {xc, yc, w, h};
}
inline Box xywh2xyxy(const Box& b) {  // ops.hpp:27-34
  float hw = b[2] * 0.5f, hh = b[3] * 0.5f;
  return {b[0] - hw, b[1] - hh, b[0] + hw, b[1] + hh};
}
inline Box xywh2tlwh(const Box& b) {  // ops.hpp:39-44
  return {b[0] - b[2] * 0.5f, b[1] - b[3] * 0.5f, b[2], b[3]};
}
inline Box tlwh2xyah(const Box& b) {  // ops.hpp:79-85
  float xc = b[0] + b[2] * 0.5f;
  float yc = b[1] + b[3] * 0.5f;
  float a = (b[3] > 0.0f) ? (b[2] / b[3]) : 0.0f;
  return {xc, yc, a, b[3]};
}
inline Box xyah2xywh(const Box& b) {  // ops.hpp:110-114
  return {b[0], b[1], b[2] * b[3], b[3]};
}
inline Box xyxy2xysr(const Box& b) {  // ops.hpp:188-197
  float w = b[2] - b[0];
  float h = b[3] - b[1];
  float xc = b[0] + w * 0.5f;
  float yc = b[1] + h * 0.5f;
  float s = w * h;
  float r = (h > 1e-6f) ? (w / h) : 0.0f;
  return {xc, yc, s, r};
}
inline Box xysr2xyxy(const Box& b) {  // ops.hpp:202-211 (NaN when s*r < 0: used for track deletion)
  float w = std::sqrt(b[2] * b[3]);
  float h = b[2] / w;
  return {b[0] - w * 0.5f, b[1] - h * 0.5f, b[0] + w * 0.5f, b[1] + h * 0.5f};
}

// ---- iou.hpp:63-100 ------------------------------------------------------------------
// Boxes are rows of A (N x >=4) and B (M x >=4); only columns 0..3 are read.
inline Mat iou_batch(const Mat& A, const Mat& B) {
  const int N = A.r, M = B.r;
  Mat out(N, M, 0.0f);
  if (N == 0 || M == 0) return out;
  std::vector<float> area1(N), area2(M);
  for (int i = 0; i < N; ++i) area1[i] = (A(i, 2) - A(i, 0)) * (A(i, 3) - A(i, 1));
  for (int j = 0; j < M; ++j) area2[j] = (B(j, 2) - B(j, 0)) * (B(j, 3) - B(j, 1));
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < M; ++j) {
      float xx1 = std::max(A(i, 0), B(j, 0));
      float yy1 = std::max(A(i, 1), B(j, 1));
      float xx2 = std::min(A(i, 2), B(j, 2));
      float yy2 = std::min(A(i, 3), B(j, 3));
      float w = std::max(0.0f, xx2 - xx1);
      float h = std::max(0.0f, yy2 - yy1);
      float inter = w * h;
      float uni = area1[i] + area2[j] - inter;
      out(i, j) = (uni > 0.0f) ? (inter / uni) : 0.0f;
    }
  }
  return out;
}

// ---- iou.hpp:122-366 — the other association measures (AssociationFunction, iou.hpp:371-414) -------------
// Elementwise float expressions in the reference's operation order. `atan` is taken as the correctly rounded float
// (through double) — the reference calls libm's atanf (<1 ulp, libm dependent); same convention as acos in OC-SORT.
enum AssoKind { ASSO_IOU = 0, ASSO_HMIOU = 1, ASSO_GIOU = 2, ASSO_CIOU = 3, ASSO_DIOU = 4, ASSO_CENTROID = 5 };
inline float emax(float a, float b) { return (a < b) ? b : a; }  // Eigen cwiseMax / std::max
inline float emin(float a, float b) { return (b < a) ? b : a; }  // Eigen cwiseMin / std::min
inline float atan_f32(float x) { return static_cast<float>(std::atan(static_cast<double>(x))); }

inline Mat hmiou_batch(const Mat& A, const Mat& B) {  // :122-150
  const int N = A.r, M = B.r;
  Mat out(N, M, 0.0f);
  if (N == 0 || M == 0) return out;
  const Mat iou = iou_batch(A, B);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float iy1 = emax(A(i, 1), B(j, 1)), iy2 = emin(A(i, 3), B(j, 3));
      const float ih = emax(iy2 - iy1, 0.0f);
      const float uy1 = emin(A(i, 1), B(j, 1)), uy2 = emax(A(i, 3), B(j, 3));
      const float uh = emax(uy2 - uy1, 1e-10f);
      const float o = ih / uh;
      out(i, j) = iou(i, j) * o;
    }
  return out;
}
inline Mat giou_batch(const Mat& A, const Mat& B) {  // :155-193
  const int N = A.r, M = B.r;
  Mat out(N, M, 0.0f);
  if (N == 0 || M == 0) return out;
  const Mat iou = iou_batch(A, B);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float xxc1 = emin(A(i, 0), B(j, 0)), yyc1 = emin(A(i, 1), B(j, 1));
      const float xxc2 = emax(A(i, 2), B(j, 2)), yyc2 = emax(A(i, 3), B(j, 3));
      const float wc = xxc2 - xxc1, hc = yyc2 - yyc1;
      const float area_enclose = wc * hc;
      const float area1 = (A(i, 2) - A(i, 0)) * (A(i, 3) - A(i, 1));
      const float area2 = (B(j, 2) - B(j, 0)) * (B(j, 3) - B(j, 1));
      const float intersection = iou(i, j) * (area1 + area2) / (iou(i, j) + 1e-10f);
      const float union_area = area1 + area2 - intersection;
      float g = iou(i, j) - (area_enclose - union_area) / (area_enclose + 1e-10f);
      out(i, j) = (g + 1.0f) / 2.0f;
    }
  return out;
}
inline Mat ciou_batch(const Mat& A, const Mat& B) {  // :198-256
  const int N = A.r, M = B.r;
  Mat out(N, M, 0.0f);
  if (N == 0 || M == 0) return out;
  const float epsilon = 1e-7f;
  const Mat iou = iou_batch(A, B);
  const float pi_squared = static_cast<float>(M_PI * M_PI);
  const float k = 4.0f / pi_squared;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float cx1 = (A(i, 0) + A(i, 2)) / 2.0f, cy1 = (A(i, 1) + A(i, 3)) / 2.0f;
      const float cx2 = (B(j, 0) + B(j, 2)) / 2.0f, cy2 = (B(j, 1) + B(j, 3)) / 2.0f;
      const float ddx = cx1 - cx2, ddy = cy1 - cy2;
      const float inner = ddx * ddx + ddy * ddy;
      const float xxc1 = emin(A(i, 0), B(j, 0)), yyc1 = emin(A(i, 1), B(j, 1));
      const float xxc2 = emax(A(i, 2), B(j, 2)), yyc2 = emax(A(i, 3), B(j, 3));
      const float ox = xxc2 - xxc1, oy = yyc2 - yyc1;
      const float outer = ox * ox + oy * oy + epsilon;
      const float w1 = A(i, 2) - A(i, 0), h1 = A(i, 3) - A(i, 1);
      const float w2 = B(j, 2) - B(j, 0), h2 = B(j, 3) - B(j, 1);
      const float ad = atan_f32(w2 / (h2 + epsilon)) - atan_f32(w1 / (h1 + epsilon));
      const float v = k * (ad * ad);
      const float S = 1.0f - iou(i, j);
      const float alpha = v / (S + v + epsilon);
      const float c = iou(i, j) - inner / outer + alpha * v;
      out(i, j) = (c + 1.0f) / 2.0f;
    }
  return out;
}
inline Mat diou_batch(const Mat& A, const Mat& B) {  // :261-298
  const int N = A.r, M = B.r;
  Mat out(N, M, 0.0f);
  if (N == 0 || M == 0) return out;
  const Mat iou = iou_batch(A, B);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float cx1 = (A(i, 0) + A(i, 2)) / 2.0f, cy1 = (A(i, 1) + A(i, 3)) / 2.0f;
      const float cx2 = (B(j, 0) + B(j, 2)) / 2.0f, cy2 = (B(j, 1) + B(j, 3)) / 2.0f;
      const float ddx = cx1 - cx2, ddy = cy1 - cy2;
      const float inner = ddx * ddx + ddy * ddy;
      const float xxc1 = emin(A(i, 0), B(j, 0)), yyc1 = emin(A(i, 1), B(j, 1));
      const float xxc2 = emax(A(i, 2), B(j, 2)), yyc2 = emax(A(i, 3), B(j, 3));
      const float ox = xxc2 - xxc1, oy = yyc2 - yyc1;
      const float outer = ox * ox + oy * oy;
      const float d = iou(i, j) - inner / (outer + 1e-10f);
      out(i, j) = (d + 1.0f) / 2.0f;
    }
  return out;
}
inline Mat centroid_batch(const Mat& A, const Mat& B, int frame_width, int frame_height) {  // :303-334
  const int N = A.r, M = B.r;
  Mat out(N, M, 0.0f);
  if (N == 0 || M == 0) return out;
  const float norm = static_cast<float>(std::sqrt(static_cast<double>(frame_width * frame_width + frame_height * frame_height)));
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float cx1 = (A(i, 0) + A(i, 2)) / 2.0f, cy1 = (A(i, 1) + A(i, 3)) / 2.0f;
      const float cx2 = (B(j, 0) + B(j, 2)) / 2.0f, cy2 = (B(j, 1) + B(j, 3)) / 2.0f;
      const float dx = cx1 - cx2, dy = cy1 - cy2;
      const float dist = std::sqrt(dx * dx + dy * dy);
      out(i, j) = 1.0f - dist / norm;
    }
  return out;
}
// AssociationFunction::operator(), iou.hpp:371-414 (the oriented-box modes are out of scope)
inline Mat asso_batch(int kind, const Mat& A, const Mat& B, int frame_width, int frame_height) {
  switch (kind) {
    case ASSO_HMIOU: return hmiou_batch(A, B);
    case ASSO_GIOU: return giou_batch(A, B);
    case ASSO_CIOU: return ciou_batch(A, B);
    case ASSO_DIOU: return diou_batch(A, B);
    case ASSO_CENTROID: return centroid_batch(A, B, frame_width, frame_height);
    default: return iou_batch(A, B);
  }
}

// matching.cpp:62-65 — 1 - IoU. (The pointer-list templates, matching.hpp:134-136,163-165,
// return Ones(m,n) for an empty side; callers below handle that case where it matters.)
inline Mat iou_distance(const Mat& A, const Mat& B) {
  Mat d = iou_batch(A, B);
  for (float& v : d.a) v = 1.0f - v;
  return d;
}

// matching.cpp:130-143 — 1 - (1 - cost) * conf_j
inline Mat fuse_score(const Mat& cost, const std::vector<float>& conf) {
  if (cost.size() == 0) return cost;
  Mat out(cost.r, cost.c);
  for (int i = 0; i < cost.r; ++i)
    for (int j = 0; j < cost.c; ++j) {
      float sim = 1.0f - cost(i, j);
      float fused = sim * conf[j];
      out(i, j) = 1.0f - fused;
    }
  return out;
}

// matching.cpp:67-92 — cosine distance max(0, 1 - a.b / (|a||b| + 1e-10)).
// Reduction order: Eigen's dot()/norm() order is unspecified (SIMD-dependent) and Eigen is
// absent here, so the oracle fixes a canonical order: a k-ordered fmaf chain (one rounding
// per term), which is also what the gfx950 fp32 MFMA computes bit-for-bit. |difference to
// any other fp32 order| <~ 1e-6 relative, inside the 1e-4 budget.
inline int& arith_mode();  // orc_kf.hpp
inline float dot_chain(const float* a, const float* b, int d) {
  if (arith_mode() != 0) {  // four lane sums (mul, then add: no fused operation), combined pairwise, then the tail — an SSE-style reduction
    float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, l3 = 0.0f;
    int k = 0;
    for (; k + 4 <= d; k += 4) {
      const float p0 = a[k] * b[k], p1 = a[k + 1] * b[k + 1], p2 = a[k + 2] * b[k + 2], p3 = a[k + 3] * b[k + 3];
      l0 += p0; l1 += p1; l2 += p2; l3 += p3;
    }
    float s = (l0 + l2) + (l1 + l3);
    for (; k < d; ++k) { const float p = a[k] * b[k]; s += p; }
    return s;
  }
  float s = 0.0f;
  for (int k = 0; k < d; ++k) s = std::fmaf(a[k], b[k], s);
  return s;
}
inline Mat embedding_distance_cosine(const Mat& T, const Mat& D) {
  const int n = T.r, m = D.r;
  Mat out(n, m, 0.0f);
  if (n == 0 || m == 0) return out;
  const int d = T.c;
  std::vector<float> tn(n), dn(m);
  for (int i = 0; i < n; ++i) tn[i] = std::sqrt(dot_chain(&T.a[(size_t)i * d], &T.a[(size_t)i * d], d));
  for (int j = 0; j < m; ++j) dn[j] = std::sqrt(dot_chain(&D.a[(size_t)j * d], &D.a[(size_t)j * d], d));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      float dp = (d > 0) ? dot_chain(&T.a[(size_t)i * d], &D.a[(size_t)j * d], d) : 0.0f;
      float sim = dp / (tn[i] * dn[j] + 1e-10f);
      out(i, j) = std::max(0.0f, 1.0f - sim);
    }
  return out;
}

// ---- lap_solver.hpp ------------------------------------------------------------------
// Jonker-Volgenant on the (n+m)^2 extension (lap_solver.hpp:289-332): real block = cost,
// both off-diagonal blocks = thresh/2, bottom-right block = 0. The extension is evaluated
// through an accessor instead of being materialised; values are identical.
struct LapResult {
  std::vector<std::array<int, 2>> matches;  // sorted by row
  std::vector<int> unmatched_a, unmatched_b;
  std::vector<int> x, y;  // row->col / col->row over the real block, -1 when unmatched
};

// Work counters of the last lapjv_rect() call (instrumentation only; results are unaffected).
struct LapStats {
  long n = 0, free_after_colred = 0, unique_rows = 0, carr_iters = 0, paths = 0, finds = 0,
       find_records = 0, scan_rows = 0, scan_ties = 0;
};
inline LapStats& lap_stats() { static thread_local LapStats s; return s; }

namespace lapdetail {
constexpr double kLarge = 1000000.0;  // lap_solver.hpp:24

struct Ext {
  const float* c;
  int nr, nc, ld;
  double half;
  int n() const { return nr + nc; }
  double at(int i, int j) const {
    if (i < nr && j < nc) return static_cast<double>(c[static_cast<size_t>(i) * ld + j]);
    if (i >= nr && j >= nc) return 0.0;
    return half;
  }
};

// lap_solver.hpp:36-72 — column reduction + reduction transfer
inline int column_reduce(const Ext& E, std::vector<int>& free_rows, std::vector<int>& x,
                         std::vector<int>& y, std::vector<double>& v) {
  const int n = E.n();
  for (int i = 0; i < n; ++i) { x[i] = -1; v[i] = kLarge; y[i] = 0; }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const double c = E.at(i, j);
      if (c < v[j]) { v[j] = c; y[j] = i; }
    }
  std::vector<char> uniq(n, 1);
  for (int j = n - 1; j >= 0; --j) {
    const int i = y[j];
    if (x[i] < 0) x[i] = j;
    else { uniq[i] = 0; y[j] = -1; }
  }
  int nfree = 0;
  for (int i = 0; i < n; ++i) {
    if (x[i] < 0) { free_rows[nfree++] = i; continue; }
    if (!uniq[i]) continue;
    ++lap_stats().unique_rows;
    const int j = x[i];
    double mn = kLarge;
    for (int j2 = 0; j2 < n; ++j2) {
      if (j2 == j) continue;
      const double c = E.at(i, j2) - v[j2];
      if (c < mn) mn = c;
    }
    v[j] -= mn;
  }
  return nfree;
}

// lap_solver.hpp:74-113 — augmenting row reduction
inline int augmenting_row_reduce(const Ext& E, int nfree, std::vector<int>& free_rows,
                                 std::vector<int>& x, std::vector<int>& y, std::vector<double>& v) {
  const unsigned n = static_cast<unsigned>(E.n());
  unsigned current = 0, rr_cnt = 0;
  int new_free = 0;
  while (current < static_cast<unsigned>(nfree)) {
    ++rr_cnt;
    ++lap_stats().carr_iters;
    const int fi = free_rows[current++];
    int j1 = 0, j2 = -1;
    double v1 = E.at(fi, 0) - v[0], v2 = kLarge;
    for (unsigned j = 1; j < n; ++j) {
      const double c = E.at(fi, static_cast<int>(j)) - v[j];
      if (c < v2) {
        if (c >= v1) { v2 = c; j2 = static_cast<int>(j); }
        else { v2 = v1; v1 = c; j2 = j1; j1 = static_cast<int>(j); }
      }
    }
    int i0 = y[j1];
    const double v1_new = v[j1] - (v2 - v1);
    const bool lowers = v1_new < v[j1];
    if (rr_cnt < current * n) {
      if (lowers) v[j1] = v1_new;
      else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = y[j2]; }
      if (i0 >= 0) {
        if (lowers) free_rows[--current] = i0;
        else free_rows[new_free++] = i0;
      }
    } else if (i0 >= 0) {
      free_rows[new_free++] = i0;
    }
    x[fi] = j1;
    y[j1] = fi;
  }
  return new_free;
}

}  // namespace lapdetail

// find_path_dense/_find_dense/_scan_dense/_ca_dense (lap_solver.hpp:115-211) are restated
// inline below. One subtlety is kept exactly: _scan_dense works on local copies of lo/hi and
// writes them back (*plo/*phi) only on its normal exit; when it returns a sink column from
// inside the k-loop (:145) the caller's lo/hi keep their pre-call values, so the dual update
// after the path search reads d[cols[lo]] with lo == n_ready.
inline void lapjv_rect(const float* cost, int nr, int nc, int ld, double thresh,
                       std::vector<int>& x_out, std::vector<int>& y_out) {
  using namespace lapdetail;
  Ext E{cost, nr, nc, ld, thresh / 2.0};
  const int n = E.n();
  std::vector<int> x(n), y(n), free_rows(n);
  std::vector<double> v(n);
  lap_stats() = LapStats();
  lap_stats().n = n;
  int nfree = column_reduce(E, free_rows, x, y, v);
  lap_stats().free_after_colred = nfree;
  for (int pass = 0; nfree > 0 && pass < 2; ++pass)  // lap_solver.hpp:220-224
    nfree = augmenting_row_reduce(E, nfree, free_rows, x, y, v);
  if (nfree > 0) {  // _ca_dense :195-211
    std::vector<int> pred(n), cols(n);
    std::vector<double> d(n);
    for (int f = 0; f < nfree; ++f) {
      const int start = free_rows[f];
      ++lap_stats().paths;
      // shortest path with exact lo/hi write-back semantics of find_path_dense
      const unsigned un = static_cast<unsigned>(n);
      unsigned lo = 0, hi = 0, n_ready = 0;
      int final_j = -1;
      for (unsigned j = 0; j < un; ++j) {
        cols[j] = static_cast<int>(j);
        pred[j] = start;
        d[j] = E.at(start, static_cast<int>(j)) - v[j];
      }
      while (final_j == -1) {
        if (lo == hi) {
          n_ready = lo;
          ++lap_stats().finds;
          hi = lo + 1;
          double mind = d[cols[lo]];
          for (unsigned k = hi; k < un; ++k) {
            const int j = cols[k];
            if (d[j] <= mind) {
              ++lap_stats().find_records;
              if (d[j] < mind) { hi = lo; mind = d[j]; }
              cols[k] = cols[hi];
              cols[hi++] = j;
            }
          }
          for (unsigned k = lo; k < hi; ++k) {
            const int j = cols[k];
            if (y[j] < 0) final_j = j;
          }
        }
        if (final_j == -1) {
          unsigned slo = lo, shi = hi;  // locals of _scan_dense
          bool returned = false;
          while (slo != shi) {
            int j = cols[slo++];
            ++lap_stats().scan_rows;
            const int i = y[j];
            const double mind = d[j];
            const double h = E.at(i, j) - v[j] - mind;
            for (unsigned k = shi; k < un; ++k) {
              j = cols[k];
              const double cred = E.at(i, j) - v[j] - h;
              if (cred < d[j]) {
                d[j] = cred;
                pred[j] = i;
                if (cred == mind) {
                  if (y[j] < 0) { final_j = j; returned = true; break; }
                  ++lap_stats().scan_ties;
                  cols[k] = cols[shi];
                  cols[shi++] = j;
                }
              }
            }
            if (returned) break;
          }
          if (!returned) { lo = slo; hi = shi; }  // *plo/*phi only written on normal exit
        }
      }
      {
        const double mind = d[cols[lo]];
        for (unsigned k = 0; k < n_ready; ++k) {
          const int j = cols[k];
          v[j] += d[j] - mind;
        }
      }
      // augment along pred (lap_solver.hpp:202-207)
      int i = -1, j = final_j;
      while (i != start) {
        i = pred[j];
        y[j] = i;
        std::swap(j, x[i]);
      }
    }
  }
  bool dump_it = std::getenv("ORC_LAP_DUMP") != nullptr;
  if (dump_it && std::getenv("ORC_LAP_DUMP_TIES")) {  // only problems holding a cost within 1e-9 of the threshold
    dump_it = false;
    for (int i = 0; i < nr && !dump_it; ++i)
      for (int j = 0; j < nc; ++j)
        if (std::fabs(static_cast<double>(cost[static_cast<size_t>(i) * ld + j]) - thresh) <= 1e-9) { dump_it = true; break; }
  }
  if (const char* dd = dump_it ? std::getenv("ORC_LAP_DUMP") : nullptr) {  // instrumentation only: the problem as raw floats, for offline replay
    static int seq = 0;
    char path[512];
    std::snprintf(path, sizeof(path), "%s/lap_%04d_%dx%d.bin", dd, seq++, nr, nc);
    if (FILE* f = std::fopen(path, "wb")) {
      const float th = static_cast<float>(thresh);
      std::fwrite(&th, 4, 1, f);
      for (int i = 0; i < nr; ++i) std::fwrite(cost + static_cast<size_t>(i) * ld, 4, nc, f);
      std::fclose(f);
    }
  }
  if (std::getenv("ORC_LAP_TRACE")) {  // instrumentation only
    const LapStats& s = lap_stats();
    std::fprintf(stderr, "lapjv %dx%d: free_after_colred %ld uniq %ld carr %ld paths %ld finds %ld records %ld scan_rows %ld ties %ld\n", nr, nc,
                 s.free_after_colred, s.unique_rows, s.carr_iters, s.paths, s.finds, s.find_records, s.scan_rows, s.scan_ties);
  }
  x_out.assign(nr, -1);
  y_out.assign(nc, -1);
  for (int i = 0; i < nr; ++i) x_out[i] = (x[i] >= nc) ? -1 : x[i];  // :326-331
  for (int j = 0; j < nc; ++j) y_out[j] = (y[j] >= nr) ? -1 : y[j];
}

// matching.cpp:14-60 + lap_solver.hpp:251-286
inline LapResult linear_assignment(const Mat& cost, float thresh) {
  LapResult R;
  const int n = cost.r, m = cost.c;
  if (n == 0 || m == 0) {
    for (int i = 0; i < n; ++i) R.unmatched_a.push_back(i);
    for (int j = 0; j < m; ++j) R.unmatched_b.push_back(j);
    R.x.assign(n, -1);
    R.y.assign(m, -1);
    return R;
  }
  lapjv_rect(cost.a.data(), n, m, m, static_cast<double>(thresh), R.x, R.y);
  for (int i = 0; i < n; ++i) {
    if (R.x[i] < 0) R.unmatched_a.push_back(i);
    else R.matches.push_back({i, R.x[i]});
  }
  for (int j = 0; j < m; ++j)
    if (R.y[j] < 0) R.unmatched_b.push_back(j);
  return R;
}

}  // namespace orc
