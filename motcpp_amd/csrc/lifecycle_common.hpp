// Shared by the device-lifecycle translation units (bt_device.hip, sort_device.hip): one wavefront per stream, the
// order-preserving compaction every "for ... push_back" of the reference turns into, and the launch helpers of the
// numeric kernels those files chain between their bookkeeping kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/motcpp_amd.h"
#include "ctx.hpp"

namespace mot {
hipError_t launch_kf_op(int op, int kind, const mot_kf_task*, int, int, hipStream_t);
hipError_t launch_det(int kind, const mot_det_task*, int, int, hipStream_t);
hipError_t launch_iou(const mot_iou_task*, int, int, int, bool, hipStream_t);
// hint_n / hint_m (0: none): sizes most problems of the launch stay within, tighter than the hard bounds max_n / max_m — the sparse
// solver sizes its LDS with them (more problems per CU) and leaves a problem that exceeds them to the exact solver
hipError_t launch_lap(const mot_lap_task*, int, int, int, bool, bool, bool, hipStream_t, int hint_n = 0, int hint_m = 0, bool try_fast = true,
                      int** declined_out = nullptr, hipEvent_t mid_event = nullptr);
size_t lap_scratch_bytes(int n, int m);
size_t lap_rowlist_scratch_bytes(int n);

namespace lifecycle {

constexpr int kW = 64;  // one wavefront per stream

// Order-preserving append: the lanes whose pred holds get consecutive positions from `base` (uniform), which advances.
__device__ __forceinline__ int compact(bool pred, int& base) {
  const unsigned long long m = __ballot(pred);
  const int pos = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
  base += __popcll(m);
  return pos;
}

// ---- packed output tables -------------------------------------------------------------------------------------------
// The per-stream tables are written into a staging area of `cap_stage` rows per stream (a stream can emit as many rows as it
// has tracks: the reference's ByteTrack keeps tracks in its tracked list without a detection in some branches), then packed
// back to back so that the copy to the host moves the emitted rows only.
// offsets[s] = first row of stream s, offsets[S] = total; counts < 0 (a staging overflow) count as 0 rows.
[[maybe_unused]] static __global__ void __launch_bounds__(1024) pack_offsets(const int* counts, int S, int* offsets) {
  __shared__ int part[1024];
  const int t = static_cast<int>(threadIdx.x);
  const int L = (S + 1023) / 1024;
  const int b0 = t * L, b1 = (b0 + L < S) ? b0 + L : S;
  int s = 0;
  for (int i = b0; i < b1; ++i) { const int c = counts[i]; s += (c > 0) ? c : 0; }
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan (Hillis-Steele; 10 rounds, once per frame)
    const int v = (t >= d) ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int base = part[t] - s;
  for (int i = b0; i < b1; ++i) { offsets[i] = base; const int c = counts[i]; base += (c > 0) ? c : 0; }
  if (t == 1023) offsets[S] = part[1023];
}
[[maybe_unused]] static __global__ void __launch_bounds__(256) pack_rows(const float* stage, int cap_stage, const int* counts, const int* offsets, float* packed,
                                                 int packed_cap) {
  const int s = blockIdx.x;
  const int c = counts[s];
  if (c <= 0) return;
  const int o = offsets[s];
  if (o + c > packed_cap) return;  // (the host sees total > packed_cap and reports it)
  const float4* src = reinterpret_cast<const float4*>(stage + static_cast<size_t>(s) * cap_stage * 8);
  float4* dst = reinterpret_cast<float4*>(packed + static_cast<size_t>(o) * 8);
  for (int i = threadIdx.x; i < 2 * c; i += 256) dst[i] = src[i];
}

// Device allocations of a batch, freed together.
struct Allocs {
  std::vector<void*> ptrs;
  template <class T>
  T* get(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return static_cast<T*>(p);
  }
  void release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
  }
};

}  // namespace lifecycle
}  // namespace mot

// HIP call inside a function returning a mot_status, with the batch's context recording the message
#define MOT_LC_HIP(b, call)                                                                                                \
  do {                                                                                                                     \
    hipError_t e_ = (call);                                                                                                \
    if (e_ != hipSuccess) { (b)->ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return MOT_ERR_HIP; }       \
  } while (0)
