// TEST HARNESS ONLY — host thread emulation of the device workgroup interface (grp.hpp) so the
// kernels' algorithmic cores (lap_core.hpp) can be exercised against the oracle on a machine
// without a GPU. One OS thread per emulated lane, pthread barrier for __syncthreads(), reductions
// through a shared scratch in a fixed lane order. Never linked into the product libraries.
#pragma once
#include <pthread.h>

#include <atomic>
#include <vector>

#include "../../motcpp_amd/csrc/grp.hpp"

namespace mot {

struct EmuShared {
  pthread_barrier_t bar;
  int T;
  std::vector<Top2> s_top2[2];
  std::vector<double> s_f64[2];
  std::vector<int> s_int[2];
  explicit EmuShared(int t) : T(t) {
    pthread_barrier_init(&bar, nullptr, t);
    for (int k = 0; k < 2; ++k) { s_top2[k].resize(t); s_f64[k].resize(t); s_int[k].resize(t); }
  }
  ~EmuShared() { pthread_barrier_destroy(&bar); }
};

struct EmuGroup {
  EmuShared* sh;
  int tid_;
  int phase = 0;
  EmuGroup(EmuShared* s, int t) : sh(s), tid_(t) {}
  int tid() const { return tid_; }
  int size() const { return sh->T; }
  void sync() { pthread_barrier_wait(&sh->bar); }
  void sync_lds() { sync(); }
  void wait_vm() {}
  void lds_barriers(bool) {}
  double reduce_min(double v) {
    auto& s = sh->s_f64[phase++ & 1];
    s[tid_] = v;
    sync();
    double r = s[0];
    for (int t = 1; t < sh->T; ++t) r = (s[t] < r) ? s[t] : r;
    return r;
  }
  float reduce_min_f32(float v) { return static_cast<float>(reduce_min(static_cast<double>(v))); }
  int reduce_max(int v) {
    auto& s = sh->s_int[phase++ & 1];
    s[tid_] = v;
    sync();
    int r = s[0];
    for (int t = 1; t < sh->T; ++t) r = (s[t] > r) ? s[t] : r;
    return r;
  }
  int reduce_min_int(int v) { return -reduce_max(-v); }
  Top2 reduce_top2(Top2 v) {
    auto& s = sh->s_top2[phase++ & 1];
    s[tid_] = v;
    sync();
    Top2 r = s[0];
    for (int t = 1; t < sh->T; ++t) r = top2_merge(r, s[t]);
    return r;
  }
  Top2 reduce_top2_under(const Top2& v, const Top2& cap, float* lb) {  // (grp.hpp: same result, same float bound; nothing is skipped here)
    const Top2 r = top2_merge(reduce_top2(v), cap);
    *lb = reduce_min_f32(f32_below(v.v1));
    return r;
  }
  double exclusive_scan_min(double v) {
    auto& s = sh->s_f64[phase++ & 1];
    s[tid_] = v;
    sync();
    double r = 1e300;
    for (int t = 0; t < tid_; ++t) r = (s[t] < r) ? s[t] : r;
    return r;
  }
  int exclusive_scan(int v, int* total) {
    auto& s = sh->s_int[phase++ & 1];
    s[tid_] = v;
    sync();
    int base = 0, tot = 0;
    for (int t = 0; t < sh->T; ++t) { if (t < tid_) base += s[t]; tot += s[t]; }
    *total = tot;
    return base;
  }
  int flag_rank(bool flag, int* total) { return exclusive_scan(flag ? 1 : 0, total); }
  struct ScanStop { int base, tot, cnt; };
  ScanStop scan_until_stop(int v, bool stop) {  // see DevGroup::scan_until_stop
    auto& a = sh->s_int[phase++ & 1];
    a[tid_] = v;
    sync();
    std::vector<int> vals(a.begin(), a.begin() + sh->T);
    auto& b = sh->s_int[phase++ & 1];
    b[tid_] = stop ? 1 : 0;
    sync();
    ScanStop r{0, 0, sh->T};
    for (int t = 0; t < sh->T; ++t) {
      if (b[t]) { r.cnt = t; break; }
      if (t < tid_) r.base += vals[t];
      r.tot += vals[t];
    }
    return r;
  }
  int reduce_sum(int v) { int tot; exclusive_scan(v, &tot); return tot; }
  bool wave_any(bool flag) { return reduce_max(flag ? 1 : 0) != 0; }  // (the emulated group is one "wavefront" whatever its size)
  unsigned long long ballot(bool flag) {
    auto& s = sh->s_int[phase++ & 1];
    s[tid_] = flag ? 1 : 0;
    sync();
    unsigned long long m = 0;
    for (int t = 0; t < sh->T && t < 64; ++t) if (s[t]) m |= 1ull << t;
    return m;
  }
  int push_i32(int v, int dst, bool active) {  // every active lane sends v to lane dst; a lane that gets nothing reads 0
    auto& s = sh->s_int[phase++ & 1];
    s[tid_] = 0;
    sync();
    if (active && dst >= 0 && dst < sh->T && dst != 63) s[dst] = v;  // (lane 63 is the device's "send nothing" target)
    sync();
    const int r = s[tid_];
    sync();
    return r;
  }
  int bcast_i32(int v, int src) {
    auto& s = sh->s_int[phase++ & 1];
    s[tid_] = v;
    sync();
    return s[src];
  }
  double bcast_f64(double v, int src) {
    auto& s = sh->s_f64[phase++ & 1];
    s[tid_] = v;
    sync();
    return s[src];
  }
  void reduce_lexmin(double& v, int& j) {
    Top2 t{v, 1e300, j, kNoIdx};
    t = reduce_top2(t);
    v = t.v1; j = t.j1;
  }
  static void atomic_max(int* p, int v) {
    int cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  }
  static int atomic_add(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
  static void atomic_min(int* p, int v) {
    int cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur > v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  }
  static void atomic_min_f64_nonneg(double* p, double v) {
    long long* q = reinterpret_cast<long long*>(p);
    const long long nv = __builtin_bit_cast(long long, v);
    long long cur = __atomic_load_n(q, __ATOMIC_RELAXED);
    while (cur > nv && !__atomic_compare_exchange_n(q, &cur, nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  }
  static void atomic_or(int* p, int v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
  static void atomic_and(int* p, int v) { __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
};

}  // namespace mot
