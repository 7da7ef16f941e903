// Workgroup abstraction shared by the gfx950 kernels and the host thread-emulation harness
// (tests/emu). The algorithms in lap_core.hpp are written against this interface only:
//   tid()/size(), sync(), reduce_min / reduce_max / reduce_top2, exclusive_scan, atomics.
// Device: one 64-lane wavefront reduces with cross-lane shuffles, wavefront partials meet in
// LDS. Host emulation (NOT a product path — test harness only): one OS thread per lane
// group of 1, partials meet in a heap scratch, pthread barrier.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MOT_DEV __device__ __forceinline__
#define MOT_HD __host__ __device__ __forceinline__
#else
#define MOT_DEV inline
#define MOT_HD inline
#endif

namespace mot {

constexpr double kLapLarge = 1000000.0;  // lap_solver.hpp:24 (LARGE)

// Two lexicographically smallest (value, index) pairs; (v1,j1) <= (v2,j2).
struct Top2 {
  double v1, v2;
  int j1, j2;
};
MOT_DEV bool lex_less(double va, int ja, double vb, int jb) { return va < vb || (va == vb && ja < jb); }
MOT_DEV void top2_push(Top2& t, double c, int j) {
  if (lex_less(c, j, t.v2, t.j2)) {
    if (lex_less(c, j, t.v1, t.j1)) { t.v2 = t.v1; t.j2 = t.j1; t.v1 = c; t.j1 = j; }
    else { t.v2 = c; t.j2 = j; }
  }
}
MOT_DEV Top2 top2_merge(Top2 a, const Top2& b) {
  top2_push(a, b.v1, b.j1);
  top2_push(a, b.v2, b.j2);
  return a;
}
constexpr int kNoIdx = 0x7fffffff;
MOT_DEV Top2 top2_empty() { return Top2{1e300, 1e300, kNoIdx, kNoIdx}; }
// a float that is <= v: the nearest one or its lower neighbour (v is not a NaN). Lower bounds may travel as floats — a float minimum is one
// DPP instruction per step where a double's is five.
MOT_DEV float f32_below(double v) {
  float f = static_cast<float>(v);
  if (static_cast<double>(f) > v) {
    const int b = __builtin_bit_cast(int, f);
    if (f > 0.f) f = __builtin_bit_cast(float, b - 1);
    else if (f < 0.f) f = __builtin_bit_cast(float, b + 1);
    else f = -1.17549435e-38f;
  }
  return f;
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------
// Device workgroup: up to 16 wavefronts of 64 lanes. `scratch` is LDS, >= 2*16*32 bytes.
struct DevGroup {
  int tid_, size_;
  char* scratch;  // LDS
  int phase = 0;  // alternates the scratch half so one barrier per reduction suffices
  __device__ DevGroup(char* lds_scratch) : tid_(threadIdx.x), size_(blockDim.x), scratch(lds_scratch) {}
  __device__ __forceinline__ int tid() const { return tid_; }
  __device__ __forceinline__ int size() const { return size_; }
  __device__ __forceinline__ void sync() { __syncthreads(); }
  // Barrier that orders LDS traffic only: global loads and stores of the wavefront stay in flight across it (a __syncthreads()
  // waits for every outstanding memory operation — a global round trip whenever a store or a prefetch is pending). The data
  // handed from lane to lane across such a barrier must live in LDS.
  __device__ __forceinline__ void sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  // waits for the wavefront's own outstanding global loads AND stores (a store counts until the L2 has it): what a later LDS-only barrier
  // needs in front of it when another wavefront is going to overwrite the same global address
  __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  // while set, the barriers INSIDE the reductions / scans below (which exchange their partials through LDS) are of that kind too
  bool lds_only_ = false;
  __device__ __forceinline__ void lds_barriers(bool on) { lds_only_ = on; }
  __device__ __forceinline__ void barrier_() { if (lds_only_) sync_lds(); else __syncthreads(); }
  // Every wavefront of the group can keep a 64-entry table with one entry per lane and read entry `src` (the same for all lanes)
  // with v_readlane: lap_core.hpp's window of upcoming SCAN members. Groups without wavefronts (tests/emu) do not define it.
  static constexpr bool kWaveTable = true;
  __device__ __forceinline__ int wave_lane() const { return tid_ & 63; }
  static __device__ __forceinline__ int wave_get(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
  static __device__ __forceinline__ unsigned long long wave_ballot(bool f) { return __builtin_amdgcn_ballot_w64(f); }
  static __device__ __forceinline__ double wave_get(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
  }

  // ---- wavefront reductions on DPP (row_shr 1,2,4,8 then row_bcast 15/31): VALU-only, no LDS crossbar round trips.
  // After the sequence lane 63 holds the reduction of all 64 lanes; v_readlane broadcasts it.
  template <int CTRL, int ROW_MASK = 0xf>
  static __device__ __forceinline__ int dpp(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);  // lanes without a source keep their own value
  }
  template <int CTRL, int ROW_MASK = 0xf>
  static __device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp<CTRL, ROW_MASK>(__double2loint(v)), hi = dpp<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
  }
  static __device__ __forceinline__ double bcast63_f64(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
  }
#define MOT_DPP_STEPS(OP)            \
  OP(0x111, 0xf) /* row_shr:1 */     \
  OP(0x112, 0xf) /* row_shr:2 */     \
  OP(0x114, 0xf) /* row_shr:4 */     \
  OP(0x118, 0xf) /* row_shr:8 */     \
  OP(0x142, 0xa) /* row_bcast:15 */  \
  OP(0x143, 0xc) /* row_bcast:31 */
  static __device__ __forceinline__ double wave_min_f64(double v) {
#define MOT_STEP(C, M) { const double o = dpp_f64<C, M>(v); v = (o < v) ? o : v; }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    return bcast63_f64(v);
  }
  static __device__ __forceinline__ float wave_min_f32(float v) {  // NaN never wins (same as `o < v ? o : v` folding)
#define MOT_STEP(C, M) { const float o = __int_as_float(dpp<C, M>(__float_as_int(v))); v = (o < v) ? o : v; }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  }
  static __device__ __forceinline__ int wave_max_i32(int v) {
#define MOT_STEP(C, M) { const int o = dpp<C, M>(v); v = (o > v) ? o : v; }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    return __builtin_amdgcn_readlane(v, 63);
  }
  // lexicographic (value, index) minimum
  static __device__ __forceinline__ void wave_lexmin(double& v, int& j) {
#define MOT_STEP(C, M) { const double ov = dpp_f64<C, M>(v); const int oj = dpp<C, M>(j); if (lex_less(ov, oj, v, j)) { v = ov; j = oj; } }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    v = bcast63_f64(v);
    j = __builtin_amdgcn_readlane(j, 63);
  }
  template <class T>
  __device__ __forceinline__ T* slot() {
    T* p = reinterpret_cast<T*>(scratch + (phase & 1) * 16 * 32);
    ++phase;
    return p;
  }
  __device__ __forceinline__ double reduce_min(double v) {
    v = wave_min_f64(v);
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return v;
    double* s = slot<double>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = v;
    barrier_();
    double r = s[0];
    for (int w = 1; w < nw; ++w) { double o = s[w]; r = (o < r) ? o : r; }
    return r;
  }
  __device__ __forceinline__ float reduce_min_f32(float v) {
    v = wave_min_f32(v);
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return v;
    float* s = slot<float>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = v;
    barrier_();
    float r = s[0];
    for (int w = 1; w < nw; ++w) { float o = s[w]; r = (o < r) ? o : r; }
    return r;
  }
  __device__ __forceinline__ int reduce_max(int v) {
    v = wave_max_i32(v);
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return v;
    int* s = slot<int>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = v;
    barrier_();
    int r = s[0];
    for (int w = 1; w < nw; ++w) { int o = s[w]; r = (o > r) ? o : r; }
    return r;
  }
  static __device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); }
  __device__ __forceinline__ int reduce_min_int(int v) { return -reduce_max(-v); }
  // two lexicographically smallest (value,index) pairs of the group: wave-level in two DPP passes (the winner,
  // then the best of everything else), wavefront partials merged through LDS
  __device__ __forceinline__ Top2 reduce_top2(Top2 t) {
    double v1 = t.v1; int j1 = t.j1;
    wave_lexmin(v1, j1);
    double v2 = (t.j1 == j1) ? t.v2 : t.v1;
    int j2 = (t.j1 == j1) ? t.j2 : t.j1;
    wave_lexmin(v2, j2);
    t.v1 = v1; t.j1 = j1; t.v2 = v2; t.j2 = j2;
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return t;
    Top2* s = slot<Top2>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = t;
    barrier_();
    Top2 r = s[0];
    for (int w = 1; w < nw; ++w) r = top2_merge(r, s[w]);
    return r;
  }
  // reduce_top2 of the threads' tuples AND the uniform tuple `cap`, for callers that know `cap` beats almost everything (round 5: the serial
  // rounds of lapjv's row reduction — a real row's two best columns are among a handful of overlapping detections and the two best dummy
  // columns, which are cached: `cap`). An entry that is not lexicographically below cap's second cannot be in the result, so each wavefront
  // collects its few entries that are with a ballot and a readlane per entry (two DPP lexmin passes — ~130 dependent VALU instructions —
  // when there are more than eight of them), the wavefronts' partials meet in LDS, cap is merged last. *lb gets a lower bound (a float) of the
  // minimum of the threads' t.v1: that minimum itself is only ever used as a bound.
  __device__ __forceinline__ Top2 reduce_top2_under(const Top2& t, const Top2& cap, float* lb) {
    const float lbw = wave_min_f32(f32_below(t.v1));
    const bool in1 = lex_less(t.v1, t.j1, cap.v2, cap.j2), in2 = lex_less(t.v2, t.j2, cap.v2, cap.j2);
    unsigned long long m1 = __builtin_amdgcn_ballot_w64(in1), m2 = __builtin_amdgcn_ballot_w64(in2);
    Top2 acc = top2_empty();
    if (__builtin_popcountll(m1) + __builtin_popcountll(m2) <= 8) {
      while (m1) { const int l = __builtin_ctzll(m1); m1 &= m1 - 1; top2_push(acc, wave_get(t.v1, l), wave_get(t.j1, l)); }
      while (m2) { const int l = __builtin_ctzll(m2); m2 &= m2 - 1; top2_push(acc, wave_get(t.v2, l), wave_get(t.j2, l)); }
    } else {
      double v1 = t.v1; int j1 = t.j1;
      wave_lexmin(v1, j1);
      double v2 = (t.j1 == j1) ? t.v2 : t.v1;
      int j2 = (t.j1 == j1) ? t.j2 : t.j1;
      wave_lexmin(v2, j2);
      acc = Top2{v1, v2, j1, j2};
    }
    const int nw = (size_ + 63) >> 6;
    Top2 r = cap;
    if (nw == 1) {
      *lb = lbw;
      if (acc.j1 != kNoIdx) top2_push(r, acc.v1, acc.j1);  // (uniform branches: most rounds have one or two entries to merge, each push is
      if (acc.j2 != kNoIdx) top2_push(r, acc.v2, acc.j2);  //  a dozen double-precision compares and selects)
      return r;
    }
    struct Part { Top2 t; float lb; int pad; };
    static_assert(sizeof(Part) == 32, "one reduction slot per wavefront");
    Part* s = slot<Part>();
    if ((tid_ & 63) == 0) { s[tid_ >> 6].t = acc; s[tid_ >> 6].lb = lbw; }
    barrier_();
    float l = s[0].lb;
    for (int w = 0; w < nw; ++w) {
      const int j1 = s[w].t.j1, j2 = s[w].t.j2;
      if (j1 != kNoIdx) top2_push(r, s[w].t.v1, j1);
      if (j2 != kNoIdx) top2_push(r, s[w].t.v2, j2);
      const float o = s[w].lb;
      l = (o < l) ? o : l;
    }
    *lb = l;
    return r;
  }
  // exclusive prefix minimum over thread ids (threads with no predecessor get +inf); general-path helper
  __device__ __forceinline__ double exclusive_scan_min(double v) {
    const int lane = tid_ & 63;
    // inclusive minimum scan on DPP (round 5; six __shfl_up steps of two LDS-crossbar permutes each before): Kogge-Stone inside the rows of 16,
    // then the row totals forwarded (row_bcast 15 / 31); a lane without a source keeps its own value
    double inc = v;
#define MOT_STEP(C, M) { const double o = dpp_f64<C, M>(inc); inc = (o < inc) ? o : inc; }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    // exclusive within the wavefront: the previous lane's inclusive value (wave_shr:1; lane 0 has no source)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(inc), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(inc), 0x138, 0xf, 0xf, false);
    double ex = (lane == 0) ? 1e300 : __hiloint2double(hi, lo);
    const int nw = (size_ + 63) >> 6;
    if (nw > 1) {
      double* s = slot<double>();
      if (lane == 63) s[tid_ >> 6] = inc;
      barrier_();
      for (int w = 0; w < (tid_ >> 6); ++w) { const double o = s[w]; if (o < ex) ex = o; }
    }
    return ex;
  }
  // exclusive prefix sum of one int per thread; *total gets the group sum
  __device__ __forceinline__ int exclusive_scan(int v, int* total) {
    const int lane = tid_ & 63;
    int inc = v;
#define MOT_STEP(C, M) { inc += __builtin_amdgcn_update_dpp(0, inc, C, M, 0xf, false); }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    const int nw = (size_ + 63) >> 6;
    int* s = slot<int>();
    if (lane == 63 || tid_ == size_ - 1) s[tid_ >> 6] = inc;
    barrier_();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { int c = s[w]; if (w < (tid_ >> 6)) base += c; tot += c; }
    *total = tot;
    return base + inc - v;
  }
  // Prefix sum that ends at the first thread with `stop` set (round 5: the scan steps of lap_core.hpp needed "first stop" and "prefix sums in front
  // of it" as two collectives in a row — two barriers and 2 x 6 LDS-crossbar shuffles per step). cnt = id of the first stopping thread (size()
  // without one); base = exclusive prefix sum of v (meaningful for threads in front of cnt), tot = sum of v over the threads in front of cnt.
  // Inside a wavefront the sums run on DPP (row_shr 1/2/4/8 Kogge-Stone, then row_bcast 15 / 31: an inclusive scan in six VALU steps).
  struct ScanStop { int base, tot, cnt; };
  static __device__ __forceinline__ int wave_inclusive_sum(int v) {
#define MOT_STEP(C, M) { v += __builtin_amdgcn_update_dpp(0, v, C, M, 0xf, false); }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    return v;
  }
  __device__ __forceinline__ ScanStop scan_until_stop(int v, bool stop) {
    const int lane = tid_ & 63, w = tid_ >> 6;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(stop);
    const int first = m ? __builtin_ctzll(m) : 64;
    const int inc = wave_inclusive_sum(v);
    const int upto = (first == 0) ? 0 : __builtin_amdgcn_readlane(inc, first - 1);  // sum of the lanes in front of the wavefront's first stop
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return ScanStop{inc - v, upto, first < size_ ? first : size_};
    int* s = slot<int>();
    if (lane == 0) { s[2 * w] = upto; s[2 * w + 1] = first; }
    barrier_();
    int base = 0, tot = 0, cnt = size_;
    bool open = true;
    for (int k = 0; k < nw; ++k) {
      const int u = s[2 * k], f = s[2 * k + 1];
      if (open) {
        if (k < w) base += u;
        tot += u;
        if (f < 64) { cnt = k * 64 + f; open = false; }
      }
    }
    return ScanStop{base + inc - v, tot, cnt};
  }
  // rank of the calling thread among the threads whose flag is set (ascending thread id) and their number
  __device__ __forceinline__ int flag_rank(bool flag, int* total) {
    if (size_ <= 64) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
      *total = __builtin_popcountll(m);
      return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
    }
    // several wavefronts: rank inside the wavefront from the ballot (mbcnt), the wavefronts' counts meet in LDS — no shuffles
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
    const int in_wave = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
    const int nw = (size_ + 63) >> 6;
    int* s = slot<int>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = __builtin_popcountll(m);
    barrier_();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const int c = s[w]; if (w < (tid_ >> 6)) base += c; tot += c; }
    *total = tot;
    return base + in_wave;
  }
  // sum of one int per thread (DPP inside the wavefront: lanes without a source add 0)
  static __device__ __forceinline__ int wave_sum_i32(int v) {
#define MOT_STEP(C, M) { v += __builtin_amdgcn_update_dpp(0, v, C, M, 0xf, false); }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    return __builtin_amdgcn_readlane(v, 63);
  }
  __device__ __forceinline__ int reduce_sum(int v) {
    v = wave_sum_i32(v);
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return v;
    int* s = slot<int>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = v;
    barrier_();
    int r = s[0];
    for (int w = 1; w < nw; ++w) r += s[w];
    return r;
  }
  // single-wavefront groups only: mask of the lanes whose flag is set, a lane's value for everyone, a value pushed to a lane
  // true when the flag is set in some lane of the calling WAVEFRONT (not of the group): the lanes of a wavefront reach it together
  __device__ __forceinline__ bool wave_any(bool flag) const { return __builtin_amdgcn_ballot_w64(flag) != 0ull; }
  __device__ __forceinline__ unsigned long long ballot(bool flag) { return __builtin_amdgcn_ballot_w64(flag); }
  __device__ __forceinline__ int push_i32(int v, int dst, bool active) { return __builtin_amdgcn_ds_permute((active ? dst : 63) << 2, active ? v : 0); }
  __device__ __forceinline__ int bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
  __device__ __forceinline__ double bcast_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
  }
  // lexicographic (value, index) minimum over the group
  __device__ __forceinline__ void reduce_lexmin(double& v, int& j) {
    if (size_ <= 64) { wave_lexmin(v, j); return; }
    Top2 t{v, 1e300, j, kNoIdx};
    t = reduce_top2(t);
    v = t.v1; j = t.j1;
  }
  static __device__ __forceinline__ void atomic_max(int* p, int v) { atomicMax(p, v); }
  static __device__ __forceinline__ int atomic_add(int* p, int v) { return atomicAdd(p, v); }
  static __device__ __forceinline__ void atomic_min(int* p, int v) { atomicMin(p, v); }
  static __device__ __forceinline__ void atomic_or(int* p, int v) { atomicOr(p, v); }
  static __device__ __forceinline__ void atomic_and(int* p, int v) { atomicAnd(p, v); }
  // minimum of non-negative doubles (no -0.0, no NaN): their bit patterns order like the values, so one 64-bit integer atomic does it
  static __device__ __forceinline__ void atomic_min_f64_nonneg(double* p, double v) {
    atomicMin(reinterpret_cast<long long*>(p), __double_as_longlong(v));
  }
};

// ---- single-wavefront helpers shared by DevGroup (64-thread blocks) and DevWave ----
struct WaveOps {
  // minimum over the wavefront of an unsigned key (DPP, one v_min per step)
  static __device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define MOT_STEP(C, M) { const unsigned o = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), C, M, 0xf, false)); v = (o < v) ? o : v; }
    MOT_DPP_STEPS(MOT_STEP)
#undef MOT_STEP
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
  }
  // lexicographic (value, index) minimum: the double through its order-preserving 2 x 32-bit key (high word, then low word
  // among the lanes that tie on it), the index among the lanes that tie on the value; lanes that do not take part pass j = kNoIdx
  static __device__ __forceinline__ void lexmin(double& v, int& j) {
    const bool in = j != kNoIdx;
    const int hi = __double2hiint(v), lo = __double2loint(v);
    const unsigned khi = in ? (static_cast<unsigned>(hi) ^ ((hi < 0) ? 0xffffffffu : 0x80000000u)) : 0xffffffffu;
    const unsigned klo = static_cast<unsigned>(lo) ^ ((hi < 0) ? 0xffffffffu : 0u);
    const unsigned mhi = wave_min_u32(khi);
    const unsigned mlo = wave_min_u32((in && khi == mhi) ? klo : 0xffffffffu);
    const bool tied = in && khi == mhi && klo == mlo;
    const unsigned mj = wave_min_u32(tied ? static_cast<unsigned>(j) : 0xffffffffu);
    const unsigned long long who = __builtin_amdgcn_ballot_w64(tied && static_cast<unsigned>(j) == mj);
    if (who == 0ull) { j = kNoIdx; return; }
    const int src = __builtin_ctzll(who);
    v = __hiloint2double(__builtin_amdgcn_readlane(hi, src), __builtin_amdgcn_readlane(lo, src));
    j = static_cast<int>(mj);
  }
  // push: every lane with `active` sends v to lane dst (distinct destinations, never lane 63); returns what this lane
  // received (0 if nothing). ds_permute has no "do not send": the other lanes send a 0 to lane 63, which is nobody's target.
  static __device__ __forceinline__ int push_i32(int v, int dst, bool active) {
    return __builtin_amdgcn_ds_permute((active ? dst : 63) << 2, active ? v : 0);
  }
};

// One wavefront acting as a group of its own inside a larger workgroup (the serial path searches of lap_sparse.hpp while
// the other wavefronts of the block wait at the block's next barrier): reductions are DPP-only, sync() orders the wavefront's
// LDS / global accesses without touching the block barrier.
struct DevWave {
  int lane_;
  __device__ DevWave() : lane_(threadIdx.x & 63) {}
  __device__ __forceinline__ int tid() const { return lane_; }
  __device__ __forceinline__ int size() const { return 64; }
  __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ int flag_rank(bool flag, int* total) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
    *total = __builtin_popcountll(m);
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
  }
  __device__ __forceinline__ void reduce_lexmin(double& v, int& j) { WaveOps::lexmin(v, j); }
  __device__ __forceinline__ int push_i32(int v, int dst, bool active) { return WaveOps::push_i32(v, dst, active); }
  __device__ __forceinline__ unsigned long long ballot(bool flag) { return __builtin_amdgcn_ballot_w64(flag); }
  __device__ __forceinline__ int bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
  __device__ __forceinline__ double bcast_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
  }
  __device__ __forceinline__ double reduce_min(double v) { return DevGroup::wave_min_f64(v); }
  __device__ __forceinline__ int reduce_max(int v) { return DevGroup::wave_max_i32(v); }
  __device__ __forceinline__ int reduce_min_int(int v) { return -DevGroup::wave_max_i32(-v); }
};
#endif  // __HIPCC__

}  // namespace mot
