// Pointers with a compile-time address space. On gfx950 a pointer whose address space the compiler cannot prove is
// accessed with flat_* instructions: they go through the address-space aperture check, count against BOTH the vector-
// memory and the LDS wait counters, and so serialise against every LDS access that follows. The assignment kernel keeps
// part of its state in LDS and part in global scratch and chooses per launch which is which — these wrappers carry that
// choice in the type, so each access compiles to ds_* or global_* and a global prefetch can stay in flight while the LDS
// work goes on. In the host emulation build (tests/emu) the spaces collapse to ordinary pointers.
#pragma once
#include "grp.hpp"

namespace mot {

enum : int { kMemAny = 0, kMemGlobal = 1, kMemLds = 3 };

#if defined(__HIP_DEVICE_COMPILE__)
#define MOT_AS(n) __attribute__((address_space(n)))
template <int AS, class T>
MOT_DEV T mem_load(const T* p) {
  if constexpr (AS == kMemGlobal) return *(const MOT_AS(1) T*)p;
  else if constexpr (AS == kMemLds) return *(const MOT_AS(3) T*)p;
  else return *p;
}
template <int AS, class T>
MOT_DEV void mem_store(T* p, T v) {
  if constexpr (AS == kMemGlobal) *(MOT_AS(1) T*)p = v;
  else if constexpr (AS == kMemLds) *(MOT_AS(3) T*)p = v;
  else *p = v;
}
#else
template <int AS, class T>
MOT_DEV T mem_load(const T* p) { return *p; }
template <int AS, class T>
MOT_DEV void mem_store(T* p, T v) { *p = v; }
#endif

// global-memory read through a plain pointer (task inputs: cost/IoU matrices, embeddings distances)
template <class T>
MOT_DEV T gld(const T* p, size_t i) { return mem_load<kMemGlobal>(p + i); }

template <int AS>
struct mem_atomic {
#if defined(__HIP_DEVICE_COMPILE__)
  template <class T> static MOT_DEV T add(T* q, T v) {
    if constexpr (AS == kMemLds) return __hip_atomic_fetch_add((MOT_AS(3) T*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (AS == kMemGlobal) return __hip_atomic_fetch_add((MOT_AS(1) T*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  template <class T> static MOT_DEV T min(T* q, T v) {
    if constexpr (AS == kMemLds) return __hip_atomic_fetch_min((MOT_AS(3) T*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (AS == kMemGlobal) return __hip_atomic_fetch_min((MOT_AS(1) T*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_fetch_min(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  template <class T> static MOT_DEV T band(T* q, T v) {
    if constexpr (AS == kMemLds) return __hip_atomic_fetch_and((MOT_AS(3) T*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (AS == kMemGlobal) return __hip_atomic_fetch_and((MOT_AS(1) T*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_fetch_and(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  template <class T> static MOT_DEV T cas(T* q, T expect, T v) {
    if constexpr (AS == kMemLds) __hip_atomic_compare_exchange_strong((MOT_AS(3) T*)q, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (AS == kMemGlobal) __hip_atomic_compare_exchange_strong((MOT_AS(1) T*)q, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_compare_exchange_strong(q, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;
  }
#else
  template <class T> static MOT_DEV T add(T* q, T v) { return __atomic_fetch_add(q, v, __ATOMIC_RELAXED); }
  template <class T> static MOT_DEV T min(T* q, T v) {
    T cur = __atomic_load_n(q, __ATOMIC_RELAXED);
    while (cur > v && !__atomic_compare_exchange_n(q, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return cur;
  }
  template <class T> static MOT_DEV T band(T* q, T v) { return __atomic_fetch_and(q, v, __ATOMIC_RELAXED); }
  template <class T> static MOT_DEV T cas(T* q, T expect, T v) { __atomic_compare_exchange_n(q, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return expect; }
#endif
};
// minimum of NON-NEGATIVE doubles (no -0.0, no NaN): their bit patterns order like the values, one 64-bit integer atomic does it
template <int AS>
MOT_DEV void mem_atomic_min_f64_nonneg(double* q, double v) {
  mem_atomic<AS>::min(reinterpret_cast<long long*>(q), __builtin_bit_cast(long long, v));
}

template <class T, int AS>
struct MemPtr {
  static constexpr int kSpace = AS;
  T* p = nullptr;
  struct Ref {
    T* q;
    MOT_DEV operator T() const { return mem_load<AS>(q); }
    MOT_DEV const Ref& operator=(T v) const { mem_store<AS>(q, v); return *this; }
    MOT_DEV const Ref& operator=(const Ref& o) const { mem_store<AS>(q, mem_load<AS>(o.q)); return *this; }
    MOT_DEV const Ref& operator+=(T v) const { mem_store<AS>(q, static_cast<T>(mem_load<AS>(q) + v)); return *this; }
    MOT_DEV const Ref& operator-=(T v) const { mem_store<AS>(q, static_cast<T>(mem_load<AS>(q) - v)); return *this; }
  };
  MOT_DEV Ref operator[](long i) const { return Ref{p + i}; }
  MOT_DEV T* raw(long i = 0) const { return p + i; }  // for atomics
  // atomics in the pointer's own address space (ds_* / global_* instead of flat_*); relaxed, workgroup scope: the callers
  // order them with the group's barriers
  MOT_DEV T atomic_add(long i, T v) const { return mem_atomic<AS>::add(p + i, v); }
  MOT_DEV T atomic_min(long i, T v) const { return mem_atomic<AS>::min(p + i, v); }
  MOT_DEV T atomic_and(long i, T v) const { return mem_atomic<AS>::band(p + i, v); }
  MOT_DEV T atomic_cas(long i, T expect, T v) const { return mem_atomic<AS>::cas(p + i, expect, v); }  // returns the old value
};

}  // namespace mot
