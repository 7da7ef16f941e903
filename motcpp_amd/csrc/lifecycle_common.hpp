// Shared by the device-lifecycle translation units (bt_device.hip, sort_device.hip): one wavefront per stream, the
// order-preserving compaction every "for ... push_back" of the reference turns into, and the launch helpers of the
// numeric kernels those files chain between their bookkeeping kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
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
// What the host needs back from a frame, gathered into ONE device buffer (dev == nullptr: nothing) so that one copy brings it home
// instead of five small ones: [0] total rows, [1] error flag, [dec_at + k] problems the sparse solver declined in association k (-1: not
// counted), [maxt_at ..) the frame's per-stream maxima, [counts_at ..) the rows per stream; counts_copy = a device copy of the counts
// that belongs to the frame (the lifecycle's own array is overwritten by the next frame).
struct PackMeta {
  int* dev = nullptr;
  const int* err = nullptr;
  const int* dec[3] = {nullptr, nullptr, nullptr};
  int dec_at = -1;
  const int* maxt = nullptr;
  int n_maxt = 0, maxt_at = 0, counts_at = 0;
  int* counts_copy = nullptr;
};
[[maybe_unused]] static __global__ void __launch_bounds__(1024) pack_offsets(const int* counts, int S, int* offsets, PackMeta pm) {
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
  if (pm.dev) {
    if (t == 0) {
      pm.dev[0] = part[1023];
      pm.dev[1] = *pm.err;
      if (pm.dec_at >= 0)
        for (int k = 0; k < 3; ++k) pm.dev[pm.dec_at + k] = pm.dec[k] ? *pm.dec[k] : -1;
    }
    for (int i = t; i < pm.n_maxt; i += 1024) pm.dev[pm.maxt_at + i] = pm.maxt[i];
    for (int i = t; i < S; i += 1024) { const int c = counts[i]; pm.dev[pm.counts_at + i] = c; if (pm.counts_copy) pm.counts_copy[i] = c; }
  }
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

// The frame's meta words, device -> page-locked host. hipMemcpyAsync takes a much slower route for a device-to-host copy above 16 KB
// (measured: one 16.6 KB copy per frame took SORT from 10.6 M to 6.4 M frames/s and ByteTrack 256 x 128 from 9.1 M to 5.5 M, with the kernels
// unchanged), so the buffer goes in pieces of at most 16 KB.
inline hipError_t copy_meta_d2h(int* h_dst, const int* d_src, size_t n_ints, hipStream_t st) {
  static const size_t piece = std::getenv("MOT_META_PIECE") ? static_cast<size_t>(std::atol(std::getenv("MOT_META_PIECE"))) : 4096;  // ints
  for (size_t o = 0; o < n_ints; o += piece) {
    const size_t n = (n_ints - o < piece) ? n_ints - o : piece;
    const hipError_t e = hipMemcpyAsync(h_dst + o, d_src + o, sizeof(int) * n, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
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

// ---- frames in flight (round 3: shared by the SORT and OC-SORT lifecycles; ByteTrack / BoT-SORT carry their own copy of the same scheme) ----
// step_packed split in two so that ONE host thread overlaps the result copy of frame f with the kernels of frame f + 1: enqueue(f + 1)
// returns once its launches are queued, collect(f) waits for frame f's event only and copies its rows on a second stream. Two sets of
// packed tables; what the host needs back from a frame travels in page-locked memory:
//   h_meta: [0] total rows, [1] error flag, [2..5) problems the sparse solver declined in up to three associations (-1: not counted),
//           [5..69) the frame's per-stream maxima (live tracks), then counts out [S], counts in [S] (host only)
constexpr int kMetaDec = 2, kMetaMaxt = 5, kMetaHead = 69;
struct Flight {
  float* d_packed = nullptr; int* d_offsets = nullptr; int* d_counts = nullptr; int packed_cap = 0;
  int* h_meta = nullptr;
  int* d_meta = nullptr;  // device image of h_meta's first kMetaHead + S words
  hipEvent_t done = nullptr;
  bool pending = false;
  int rows_cap = 0;  // the row limit pack_rows ran with (the enqueue call's rows_cap)
  int bd = 0;        // largest detection count of the frame (bounds the tracks it may add)
};
struct Flights {
  Flight fl[2];
  int head = 0, count = 0;  // oldest pending frame, frames pending
  hipStream_t copy_st = nullptr;
  int slot_for_enqueue() const { return (head + count) & 1; }
  // buffers of the slot (allocated on first use / when rows_cap grows); counts_in = page-locked copy of the caller's counts
  hipError_t prepare(Allocs& mem, int slot, int S, int rows_cap, const int* h_counts, int** counts_in, int* bd_out) {
    Flight& F = fl[slot];
    hipError_t e = hipSuccess;
    if (!copy_st && (e = hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking)) != hipSuccess) return e;
    if (!F.done && (e = hipEventCreateWithFlags(&F.done, hipEventDisableTiming)) != hipSuccess) return e;
    if (!F.h_meta && (e = hipHostMalloc(reinterpret_cast<void**>(&F.h_meta), sizeof(int) * (kMetaHead + 2 * static_cast<size_t>(S)), hipHostMallocDefault)) != hipSuccess) return e;
    if (!F.d_offsets) { F.d_offsets = mem.get<int>(static_cast<size_t>(S) + 1); F.d_counts = mem.get<int>(S); F.d_meta = mem.get<int>(kMetaHead + static_cast<size_t>(S)); }
    if (rows_cap > F.packed_cap) { F.d_packed = mem.get<float>(static_cast<size_t>(rows_cap) * 8); F.packed_cap = F.d_packed ? rows_cap : 0; }
    if (!F.d_offsets || !F.d_counts || !F.d_packed || !F.d_meta) return hipErrorOutOfMemory;
    int* ci = F.h_meta + kMetaHead + S;
    int bd = 1;
    for (int s = 0; s < S; ++s) { ci[s] = h_counts[s]; bd = (h_counts[s] > bd) ? h_counts[s] : bd; }
    *counts_in = ci; *bd_out = bd;
    return hipSuccess;
  }
  // behind the frame's launches: pack the staged tables, bring the small results back, record the frame's event
  hipError_t finish(int slot, hipStream_t st, const float* d_stage, int cap_stage, const int* d_out_counts, int S, const int* d_err, const int* d_maxt,
                    const int* d_declined, int rows_cap, int bd, const int* d_declined_b = nullptr, const int* d_declined_c = nullptr) {
    Flight& F = fl[slot];
    PackMeta pm;
    pm.dev = F.d_meta; pm.err = d_err; pm.dec[0] = d_declined; pm.dec[1] = d_declined_b; pm.dec[2] = d_declined_c; pm.dec_at = kMetaDec;
    pm.maxt = d_maxt; pm.n_maxt = 64; pm.maxt_at = kMetaMaxt; pm.counts_at = kMetaHead; pm.counts_copy = F.d_counts;
    hipLaunchKernelGGL(pack_offsets, dim3(1), dim3(1024), 0, st, d_out_counts, S, F.d_offsets, pm);
    hipLaunchKernelGGL(pack_rows, dim3(S), dim3(256), 0, st, d_stage, cap_stage, d_out_counts, F.d_offsets, F.d_packed, rows_cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if ((e = copy_meta_d2h(F.h_meta, F.d_meta, kMetaHead + static_cast<size_t>(S), st)) != hipSuccess) return e;
    if ((e = hipEventRecord(F.done, st)) != hipSuccess) return e;
    F.pending = true; F.bd = bd; F.rows_cap = rows_cap;
    count += 1;
    return hipSuccess;
  }
  // the oldest pending frame: waits for it, hands out its meta words; the caller copies the rows with copy_rows
  hipError_t pop(Flight** out) {
    Flight& F = fl[head];
    const hipError_t e = hipEventSynchronize(F.done);
    if (e != hipSuccess) return e;
    F.pending = false;
    head ^= 1; count -= 1;
    *out = &F;
    return hipSuccess;
  }
  hipError_t copy_rows(const Flight& F, float* rows, int total) {
    if (total <= 0) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(rows, F.d_packed, sizeof(float) * static_cast<size_t>(total) * 8, hipMemcpyDeviceToHost, copy_st);
    return (e != hipSuccess) ? e : hipStreamSynchronize(copy_st);
  }
  int pending_bd() const { int v = 0; for (const Flight& F : fl) if (F.pending && F.bd > v) v = F.bd; return v; }
  void drop_all() { for (Flight& F : fl) F.pending = false; head = 0; count = 0; }
  void release() {
    for (Flight& F : fl) { if (F.done) (void)hipEventDestroy(F.done); if (F.h_meta) (void)hipHostFree(F.h_meta); F.done = nullptr; F.h_meta = nullptr; }
    if (copy_st) (void)hipStreamDestroy(copy_st);
    copy_st = nullptr;
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
