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

  static __device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
  }
  template <class T>
  __device__ __forceinline__ T* slot() {
    T* p = reinterpret_cast<T*>(scratch + (phase & 1) * 16 * 32);
    ++phase;
    return p;
  }
  __device__ __forceinline__ double reduce_min(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { double o = shfl_xor_f64(v, m); v = (o < v) ? o : v; }
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return v;
    double* s = slot<double>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = v;
    __syncthreads();
    double r = s[0];
    for (int w = 1; w < nw; ++w) { double o = s[w]; r = (o < r) ? o : r; }
    return r;
  }
  __device__ __forceinline__ int reduce_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m, 64); v = (o > v) ? o : v; }
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return v;
    int* s = slot<int>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = v;
    __syncthreads();
    int r = s[0];
    for (int w = 1; w < nw; ++w) { int o = s[w]; r = (o > r) ? o : r; }
    return r;
  }
  __device__ __forceinline__ int reduce_min_int(int v) { return -reduce_max(-v); }
  __device__ __forceinline__ Top2 reduce_top2(Top2 t) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      Top2 o;
      o.v1 = shfl_xor_f64(t.v1, m); o.v2 = shfl_xor_f64(t.v2, m);
      o.j1 = __shfl_xor(t.j1, m, 64); o.j2 = __shfl_xor(t.j2, m, 64);
      t = top2_merge(t, o);
    }
    const int nw = (size_ + 63) >> 6;
    if (nw == 1) return t;
    Top2* s = slot<Top2>();
    if ((tid_ & 63) == 0) s[tid_ >> 6] = t;
    __syncthreads();
    Top2 r = s[0];
    for (int w = 1; w < nw; ++w) r = top2_merge(r, s[w]);
    return r;
  }
  // exclusive prefix sum of one int per thread; *total gets the group sum
  __device__ __forceinline__ int exclusive_scan(int v, int* total) {
    const int lane = tid_ & 63;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    const int nw = (size_ + 63) >> 6;
    int* s = slot<int>();
    if (lane == 63 || tid_ == size_ - 1) s[tid_ >> 6] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { int c = s[w]; if (w < (tid_ >> 6)) base += c; tot += c; }
    *total = tot;
    return base + inc - v;
  }
  static __device__ __forceinline__ void atomic_max(int* p, int v) { atomicMax(p, v); }
  static __device__ __forceinline__ int atomic_add(int* p, int v) { return atomicAdd(p, v); }
};
#endif  // __HIPCC__

}  // namespace mot
