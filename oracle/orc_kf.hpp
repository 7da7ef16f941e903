// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// CPU restatement of motcpp's three Kalman filters, fp32, dense small-matrix loops in the
// same expression order as the reference's Eigen expressions, every inner product summed
// in k order, no FMA contraction:
//   KalmanFilterXYSR            src/motion/kalman_filters/xysr_kf.cpp:10-112
//   BaseKalmanFilter + XYAH     src/motion/kalman_filter.cpp:29-112, kalman_filters/xyah_kf.cpp:14-62
//   KalmanFilterXYWH            include/motcpp/motion/kalman_filters/xywh_kf.hpp:41-135
// Parity status: the reference's tests pin only XYSR, loosely (tests/test_kalman_filter.cpp).
// Eigen (un-vendored dependency, any 3.3+) is absent here, so the rounding order inside its
// GEMM / LLT / PartialPivLU kernels is restated from its documented algorithms
// (unblocked Cholesky, column-axpy forward / row-dot backward substitution, partial-pivot
// Doolittle LU) — "parity unpinned" at the last-bit level, 1e-4 relative by contract.
#pragma once
#include <cmath>
#include <utility>

namespace orc {

// Arithmetic mode of the restatement (process-wide; test infrastructure for tests/test_oracle_arith_modes.py).
// The reference's floats come out of Eigen (absent here), whose summation orders are not part of its interface: mode 0 is the order this
// oracle — and the gfx950 kernels, bit for bit — use; mode 1 re-derives every Eigen-dependent quantity in a DIFFERENT but equally
// plausible order, so that a test can show that assignments and track ids do not hinge on the choice:
//   small matrix products   k-ordered mul+add            -> k-ordered with fused multiply-add (an -mfma build of the reference)
//   4x4 Cholesky            left-looking (sums, then one subtraction) -> right-looking (the trailing block updated column by column)
//   triangular solves       column-axpy forward          -> row-dot forward
//   4x4 inverse (XYWH)      partial-pivot LU             -> cofactor expansion (what Eigen does for FIXED-size 4x4)
//   dot / norm (cosine)     k-ordered fmaf chain         -> four lane sums, mul then add, pairwise combined (Eigen's SSE reduction)
inline int& arith_mode() { static int m = 0; return m; }

template <int R, int C>
struct SMat {
  float a[R][C];
  float* operator[](int i) { return a[i]; }
  const float* operator[](int i) const { return a[i]; }
  static SMat zero() {
    SMat m;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) m.a[i][j] = 0.0f;
    return m;
  }
  static SMat identity() {
    SMat m = zero();
    for (int i = 0; i < (R < C ? R : C); ++i) m.a[i][i] = 1.0f;
    return m;
  }
};

template <int R, int K, int C>
inline SMat<R, C> mul(const SMat<R, K>& A, const SMat<K, C>& B) {
  SMat<R, C> o;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      float s = A[i][0] * B[0][j];
      if (arith_mode() == 0) { for (int k = 1; k < K; ++k) s += A[i][k] * B[k][j]; }
      else { for (int k = 1; k < K; ++k) s = std::fmaf(A[i][k], B[k][j], s); }
      o[i][j] = s;
    }
  return o;
}
template <int R, int C>
inline SMat<C, R> transpose(const SMat<R, C>& A) {
  SMat<C, R> o;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) o[j][i] = A[i][j];
  return o;
}
template <int R, int C>
inline SMat<R, C> add(const SMat<R, C>& A, const SMat<R, C>& B) {
  SMat<R, C> o;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) o[i][j] = A[i][j] + B[i][j];
  return o;
}
template <int R, int C>
inline SMat<R, C> sub(const SMat<R, C>& A, const SMat<R, C>& B) {
  SMat<R, C> o;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) o[i][j] = A[i][j] - B[i][j];
  return o;
}

// Unblocked lower Cholesky (Eigen llt_inplace<Lower>::unblocked): diagonal first, then the
// column below it; inner sums are formed first and subtracted once.
template <int N>
inline bool cholesky(SMat<N, N>& A) {
  if (arith_mode() != 0) {  // right-looking: every column's outer product is subtracted from the trailing block as soon as it exists
    for (int k = 0; k < N; ++k) {
      float x = A[k][k];
      if (!(x > 0.0f)) return false;
      x = std::sqrt(x);
      A[k][k] = x;
      for (int i = k + 1; i < N; ++i) A[i][k] = A[i][k] / x;
      for (int j = k + 1; j < N; ++j)
        for (int i = j; i < N; ++i) A[i][j] -= A[i][k] * A[j][k];
    }
    return true;
  }
  for (int k = 0; k < N; ++k) {
    float x = A[k][k];
    if (k > 0) {
      float s = A[k][0] * A[k][0];
      for (int j = 1; j < k; ++j) s += A[k][j] * A[k][j];
      x -= s;
    }
    if (!(x > 0.0f)) return false;
    x = std::sqrt(x);
    A[k][k] = x;
    for (int i = k + 1; i < N; ++i) {
      float t = A[i][k];
      if (k > 0) {
        float s = A[i][0] * A[k][0];
        for (int j = 1; j < k; ++j) s += A[i][j] * A[k][j];
        t -= s;
      }
      A[i][k] = t / x;
    }
  }
  return true;
}
// Solve (L L^T) z = b in place: forward substitution column-axpy style, backward row-dot style.
template <int N>
inline void chol_solve(const SMat<N, N>& L, float* b) {
  if (arith_mode() != 0) {  // forward substitution row-dot style
    for (int i = 0; i < N; ++i) {
      float s = 0.0f;
      for (int j = 0; j < i; ++j) s += L[i][j] * b[j];
      b[i] = (b[i] - s) / L[i][i];
    }
  } else
  for (int i = 0; i < N; ++i) {
    b[i] /= L[i][i];
    for (int r = i + 1; r < N; ++r) b[r] -= b[i] * L[r][i];
  }
  for (int i = N - 1; i >= 0; --i) {
    if (i < N - 1) {
      float s = L[i + 1][i] * b[i + 1];
      for (int j = i + 2; j < N; ++j) s += L[j][i] * b[j];
      b[i] -= s;
    }
    b[i] /= L[i][i];
  }
}

// Partial-pivot LU inverse of a 4x4 (Eigen's dynamic-size inverse() = partialPivLu().inverse()).
inline SMat<4, 4> inverse_lu4(const SMat<4, 4>& S) {
  if (arith_mode() != 0) {  // cofactor expansion
    SMat<4, 4> inv;
    auto det3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
      return S[r0][c0] * (S[r1][c1] * S[r2][c2] - S[r1][c2] * S[r2][c1]) - S[r0][c1] * (S[r1][c0] * S[r2][c2] - S[r1][c2] * S[r2][c0]) +
             S[r0][c2] * (S[r1][c0] * S[r2][c1] - S[r1][c1] * S[r2][c0]);
    };
    float cof[4][4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        int r[3], c[3], a = 0, b = 0;
        for (int q = 0; q < 4; ++q) { if (q != i) r[a++] = q; if (q != j) c[b++] = q; }
        const float m = det3(r[0], r[1], r[2], c[0], c[1], c[2]);
        cof[i][j] = ((i + j) & 1) ? -m : m;
      }
    const float det = S[0][0] * cof[0][0] + S[0][1] * cof[0][1] + S[0][2] * cof[0][2] + S[0][3] * cof[0][3];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) inv[i][j] = cof[j][i] / det;
    return inv;
  }
  SMat<4, 4> lu = S;
  int perm[4] = {0, 1, 2, 3};
  for (int k = 0; k < 4; ++k) {
    int p = k;
    float best = std::fabs(lu[k][k]);
    for (int i = k + 1; i < 4; ++i) {
      float v = std::fabs(lu[i][k]);
      if (v > best) { best = v; p = i; }
    }
    if (p != k) {
      for (int j = 0; j < 4; ++j) std::swap(lu[k][j], lu[p][j]);
      std::swap(perm[k], perm[p]);
    }
    for (int i = k + 1; i < 4; ++i) lu[i][k] /= lu[k][k];
    for (int i = k + 1; i < 4; ++i)
      for (int j = k + 1; j < 4; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
  SMat<4, 4> inv;
  for (int c = 0; c < 4; ++c) {
    float b[4];
    for (int i = 0; i < 4; ++i) b[i] = (perm[i] == c) ? 1.0f : 0.0f;
    for (int i = 0; i < 4; ++i)  // unit-lower forward, column-axpy
      for (int r = i + 1; r < 4; ++r) b[r] -= b[i] * lu[r][i];
    for (int i = 3; i >= 0; --i) {  // upper backward, column-axpy
      b[i] /= lu[i][i];
      for (int r = 0; r < i; ++r) b[r] -= b[i] * lu[r][i];
    }
    for (int i = 0; i < 4; ++i) inv[i][c] = b[i];
  }
  return inv;
}

// ---------------------------------------------------------------------------------------
// KalmanFilterXYSR — xysr_kf.cpp:10-112. State [x,y,s,r,vx,vy,vs], observation [x,y,s,r].
struct KfXYSR {
  SMat<7, 7> F, P, Q;
  SMat<4, 7> H;
  SMat<4, 4> R;
  float x[7];
  KfXYSR() {  // :10-69
    F = SMat<7, 7>::identity();
    F[0][4] = 1.0f; F[1][5] = 1.0f; F[2][6] = 1.0f;
    H = SMat<4, 7>::zero();
    for (int i = 0; i < 4; ++i) H[i][i] = 1.0f;
    for (float& v : x) v = 0.0f;
    P = SMat<7, 7>::identity();
    for (int i = 0; i < 7; ++i) P[i][i] *= 10.0f;
    for (int i = 4; i < 7; ++i)
      for (int j = 4; j < 7; ++j) P[i][j] *= 100.0f;
    Q = SMat<7, 7>::identity();
    Q[4][4] = 0.01f; Q[5][5] = 0.01f; Q[6][6] = 0.0001f;
    R = SMat<4, 4>::identity();
    for (int i = 2; i < 4; ++i)
      for (int j = 2; j < 4; ++j) R[i][j] *= 10.0f;
  }
  void predict() {  // :71-77
    float nx[7];
    for (int i = 0; i < 7; ++i) {
      float s = F[i][0] * x[0];
      for (int k = 1; k < 7; ++k) s += F[i][k] * x[k];
      nx[i] = s;
    }
    for (int i = 0; i < 7; ++i) x[i] = nx[i];
    P = add(mul(mul(F, P), transpose(F)), Q);
  }
  void update(const float z[4]) {  // :79-112 (a size-0 z is the caller's no-op, :80-82)
    float y[4];
    for (int i = 0; i < 4; ++i) {
      float s = H[i][0] * x[0];
      for (int k = 1; k < 7; ++k) s += H[i][k] * x[k];
      y[i] = z[i] - s;
    }
    SMat<7, 4> Ht = transpose(H);
    SMat<4, 4> S = add(mul(mul(H, P), Ht), R);
    SMat<4, 4> L = S;
    SMat<4, 4> Sinv;
    if (cholesky(L)) {
      for (int c = 0; c < 4; ++c) {
        float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        b[c] = 1.0f;
        chol_solve(L, b);
        for (int i = 0; i < 4; ++i) Sinv[i][c] = b[i];
      }
    } else {
      Sinv = inverse_lu4(S);  // stand-in for the pseudo-inverse fallback (:98-104); never hit for SPD S
    }
    SMat<7, 4> K = mul(mul(P, Ht), Sinv);
    for (int i = 0; i < 7; ++i) {
      float s = K[i][0] * y[0];
      for (int k = 1; k < 4; ++k) s += K[i][k] * y[k];
      x[i] = x[i] + s;
    }
    SMat<7, 7> IKH = sub(SMat<7, 7>::identity(), mul(K, H));
    SMat<7, 7> first = mul(mul(IKH, P), transpose(IKH));
    SMat<7, 7> second = mul(mul(K, R), transpose(K));
    P = add(first, second);
  }
  // apply_affine_correction :114-141 — x[:2] = m x[:2] + t, x[4:6] = m x[4:6], and the position, velocity and
  // position-velocity blocks of P become m B m^T (the lower-left block is the transpose of the new upper-right one).
  void apply_affine_correction(const float m[2][2], const float t[2]) {
    const float cx = x[0], cy = x[1], vx = x[4], vy = x[5];
    x[0] = (m[0][0] * cx + m[0][1] * cy) + t[0];
    x[1] = (m[1][0] * cx + m[1][1] * cy) + t[1];
    x[4] = m[0][0] * vx + m[0][1] * vy;
    x[5] = m[1][0] * vx + m[1][1] * vy;
    auto block = [&](int r0, int c0) {
      float A[2][2], T[2][2];
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) A[i][j] = P[r0 + i][c0 + j];
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) T[i][j] = m[i][0] * A[0][j] + m[i][1] * A[1][j];
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) P[r0 + i][c0 + j] = T[i][0] * m[j][0] + T[i][1] * m[j][1];
    };
    block(0, 0);
    block(4, 4);
    block(0, 4);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) P[4 + i][j] = P[j][4 + i];
  }
};

// ---------------------------------------------------------------------------------------
// 8-state filters share this shape: mean[8], cov[8][8], F = I + shift(4), H = [I 0].
struct State8 {
  float mean[8];
  SMat<8, 8> cov;
};

inline SMat<8, 8> motion8() {
  SMat<8, 8> F = SMat<8, 8>::identity();
  for (int i = 0; i < 4; ++i) F[i][4 + i] = 1.0f;  // dt = 1
  return F;
}
inline SMat<4, 8> obs8() {
  SMat<4, 8> H = SMat<4, 8>::zero();
  for (int i = 0; i < 4; ++i) H[i][i] = 1.0f;
  return H;
}
inline void predict8(State8& s, const float std8[8]) {
  const SMat<8, 8> F = motion8();
  float nm[8];
  for (int i = 0; i < 8; ++i) {
    float acc = F[i][0] * s.mean[0];
    for (int k = 1; k < 8; ++k) acc += F[i][k] * s.mean[k];
    nm[i] = acc;
  }
  SMat<8, 8> Q = SMat<8, 8>::zero();
  for (int i = 0; i < 8; ++i) Q[i][i] = std8[i] * std8[i];
  s.cov = add(mul(mul(F, s.cov), transpose(F)), Q);
  for (int i = 0; i < 8; ++i) s.mean[i] = nm[i];
}

// KalmanFilterXYAH (ByteTrack) — kalman_filter.cpp + xyah_kf.cpp
struct KfXYAH {
  static constexpr float wp = 1.0f / 20.0f;   // kalman_filter.cpp:13
  static constexpr float wv = 1.0f / 160.0f;  // kalman_filter.cpp:14
  static State8 initiate(const float m[4]) {  // kalman_filter.cpp:29-42, xyah_kf.cpp:14-30
    State8 s;
    for (int i = 0; i < 4; ++i) { s.mean[i] = m[i]; s.mean[4 + i] = 0.0f; }
    const float h = m[3];
    const float sd[8] = {2.0f * wp * h, 2.0f * wp * h, 1e-2f, 2.0f * wp * h,
                         10.0f * wv * h, 10.0f * wv * h, 1e-5f, 10.0f * wv * h};
    s.cov = SMat<8, 8>::zero();
    for (int i = 0; i < 8; ++i) s.cov[i][i] = sd[i] * sd[i];
    return s;
  }
  static void predict(State8& s) {  // kalman_filter.cpp:44-58, xyah_kf.cpp:32-49
    const float h = s.mean[3];
    const float sd[8] = {wp * h, wp * h, 1e-2f, wp * h, wv * h, wv * h, 1e-5f, wv * h};
    predict8(s, sd);
  }
  static void update(State8& s, const float z[4], float confidence = 0.0f) {  // kalman_filter.cpp:60-112
    const float h = s.mean[3];
    float sd[4] = {wp * h, wp * h, 1e-1f, wp * h};  // xyah_kf.cpp:51-62
    for (float& v : sd) v = v * (1.0f - confidence);  // NSA, kalman_filter.cpp:67
    const SMat<4, 8> H = obs8();
    const SMat<8, 4> Ht = transpose(H);
    float pm[4];
    for (int i = 0; i < 4; ++i) {
      float acc = H[i][0] * s.mean[0];
      for (int k = 1; k < 8; ++k) acc += H[i][k] * s.mean[k];
      pm[i] = acc;
    }
    SMat<4, 4> Rn = SMat<4, 4>::zero();
    for (int i = 0; i < 4; ++i) Rn[i][i] = sd[i] * sd[i];
    SMat<4, 4> S = add(mul(mul(H, s.cov), Ht), Rn);
    SMat<4, 4> L = S;
    SMat<8, 4> PHt = mul(s.cov, Ht);
    SMat<8, 4> K;
    if (cholesky(L)) {  // row-wise solves, :103-105
      for (int i = 0; i < 8; ++i) {
        float b[4] = {PHt[i][0], PHt[i][1], PHt[i][2], PHt[i][3]};
        chol_solve(L, b);
        for (int c = 0; c < 4; ++c) K[i][c] = b[c];
      }
    } else {  // :86-94 fallback (pseudo-inverse); LU inverse stands in, unreachable for SPD S
      K = mul(PHt, inverse_lu4(S));
    }
    float inn[4];
    for (int i = 0; i < 4; ++i) inn[i] = z[i] - pm[i];
    for (int i = 0; i < 8; ++i) {
      float acc = K[i][0] * inn[0];
      for (int k = 1; k < 4; ++k) acc += K[i][k] * inn[k];
      s.mean[i] = s.mean[i] + acc;
    }
    s.cov = sub(s.cov, mul(mul(K, S), transpose(K)));
  }
};

// KalmanFilterXYWH (BoT-SORT) — xywh_kf.hpp:41-135
struct KfXYWH {
  static constexpr float wp = 1.0f / 20.0f;
  static constexpr float wv = 1.0f / 160.0f;
  static State8 initiate(const float m[4]) {  // :41-62
    State8 s;
    for (int i = 0; i < 4; ++i) { s.mean[i] = m[i]; s.mean[4 + i] = 0.0f; }
    const float h = m[3];
    s.cov = SMat<8, 8>::zero();
    for (int i = 0; i < 8; ++i) {
      const float sd = (i < 4) ? 2.0f * wp * h : 10.0f * wv * h;
      s.cov[i][i] = sd * sd;
    }
    return s;
  }
  static void predict(State8& s) {  // :70-94
    const float h = s.mean[3];
    float sd[8];
    for (int i = 0; i < 8; ++i) sd[i] = (i < 4) ? wp * h : wv * h;
    predict8(s, sd);
  }
  static void update(State8& s, const float z[4]) {  // :103-135
    const float h = s.mean[3];
    const SMat<4, 8> H = obs8();
    const SMat<8, 4> Ht = transpose(H);
    SMat<4, 4> Rn = SMat<4, 4>::zero();
    for (int i = 0; i < 4; ++i) { const float sd = wp * h; Rn[i][i] = sd * sd; }
    float pm[4];
    for (int i = 0; i < 4; ++i) {
      float acc = H[i][0] * s.mean[0];
      for (int k = 1; k < 8; ++k) acc += H[i][k] * s.mean[k];
      pm[i] = acc;
    }
    SMat<4, 4> S = add(mul(mul(H, s.cov), Ht), Rn);
    SMat<8, 4> K = mul(mul(s.cov, Ht), inverse_lu4(S));
    float inn[4];
    for (int i = 0; i < 4; ++i) inn[i] = z[i] - pm[i];
    for (int i = 0; i < 8; ++i) {
      float acc = K[i][0] * inn[0];
      for (int k = 1; k < 4; ++k) acc += K[i][k] * inn[k];
      s.mean[i] = s.mean[i] + acc;
    }
    s.cov = sub(s.cov, mul(mul(K, S), transpose(K)));
  }
};

// ---- gating distances (StrongSORT's motion gate; "parity unpinned": the reference holds no known answer for them) ----
// BaseKalmanFilter::gating_distance, kalman_filter.cpp:148-176, for the XYAH filter: project() (:60-75, confidence 0), the
// leading dim x dim block of S, d = z - H x. "maha" (metric 0): z = LLT(S_sub).solve(d) — the full solve, so the value is
// |S^-1 d|^2, not d^T S^-1 d — summed in component order; a failed factorisation or "gaussian" (metric 1): |d|^2.
// meas: n rows of 4 (xyah). The dynamic-size triangular solves are restated with chol_solve's operation order.
inline void gating_xyah(const State8& s, const float* meas, int n, bool only_position, int metric, float* out) {
  const float h = s.mean[3];
  float sd[4] = {KfXYAH::wp * h, KfXYAH::wp * h, 1e-1f, KfXYAH::wp * h};
  for (float& v : sd) v = v * (1.0f - 0.0f);
  const SMat<4, 8> H = obs8();
  SMat<4, 4> Rn = SMat<4, 4>::zero();
  for (int i = 0; i < 4; ++i) Rn[i][i] = sd[i] * sd[i];
  const SMat<4, 4> S = add(mul(mul(H, s.cov), transpose(H)), Rn);
  float pm[4];
  for (int i = 0; i < 4; ++i) {
    float acc = H[i][0] * s.mean[0];
    for (int k = 1; k < 8; ++k) acc += H[i][k] * s.mean[k];
    pm[i] = acc;
  }
  const int dim = only_position ? 2 : 4;
  SMat<4, 4> L4 = S;
  SMat<2, 2> L2;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) L2[a][b] = S[a][b];
  const bool ok = (metric == 0) && (only_position ? cholesky(L2) : cholesky(L4));
  for (int j = 0; j < n; ++j) {
    float d[4];
    for (int k = 0; k < dim; ++k) d[k] = meas[4 * j + k] - pm[k];
    if (ok) { if (only_position) chol_solve(L2, d); else chol_solve(L4, d); }
    float g = d[0] * d[0];
    for (int k = 1; k < dim; ++k) g += d[k] * d[k];
    out[j] = g;
  }
}
// KalmanFilterXYWH::gating_distance, xywh_kf.hpp:140-176: S = H P H^T + diag((wp h)^2), S^-1 by partial-pivot LU,
// d^T S^-1 d as (d^T S^-1) d; only_position uses the leading 2 x 2 block of the 4 x 4 inverse. meas: n rows of 4 (xywh).
inline void gating_xywh(const State8& s, const float* meas, int n, bool only_position, float* out) {
  const float h = s.mean[3];
  const SMat<4, 8> H = obs8();
  SMat<4, 4> Rn = SMat<4, 4>::zero();
  for (int i = 0; i < 4; ++i) { const float sd = KfXYWH::wp * h; Rn[i][i] = sd * sd; }
  const SMat<4, 4> S = add(mul(mul(H, s.cov), transpose(H)), Rn);
  const SMat<4, 4> Si = inverse_lu4(S);
  float pm[4];
  for (int i = 0; i < 4; ++i) {
    float acc = H[i][0] * s.mean[0];
    for (int k = 1; k < 8; ++k) acc += H[i][k] * s.mean[k];
    pm[i] = acc;
  }
  const int dim = only_position ? 2 : 4;
  for (int j = 0; j < n; ++j) {
    float d[4], t[4];
    for (int k = 0; k < 4; ++k) d[k] = meas[4 * j + k] - pm[k];
    for (int c = 0; c < dim; ++c) {
      float a = d[0] * Si[0][c];
      for (int k = 1; k < dim; ++k) a += d[k] * Si[k][c];
      t[c] = a;
    }
    float g = t[0] * d[0];
    for (int k = 1; k < dim; ++k) g += t[k] * d[k];
    out[j] = g;
  }
}

}  // namespace orc
