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
hipError_t launch_lap(const mot_lap_task*, int, int, int, bool, bool, bool, hipStream_t);
size_t lap_scratch_bytes(int n, int m);

namespace lifecycle {

constexpr int kW = 64;  // one wavefront per stream

// Order-preserving append: the lanes whose pred holds get consecutive positions from `base` (uniform), which advances.
__device__ __forceinline__ int compact(bool pred, int& base) {
  const unsigned long long m = __ballot(pred);
  const int pos = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
  base += __popcll(m);
  return pos;
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
