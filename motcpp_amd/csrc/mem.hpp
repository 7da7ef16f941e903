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

template <class T, int AS>
struct MemPtr {
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
};

}  // namespace mot
