// Shared by the device-lifecycle translation units (bt_device.hip, sort_device.hip): one wavefront per stream, the
// order-preserving compaction every "for ... push_back" of the reference turns into, and the launch helpers of the
// numeric kernels those files chain between their bookkeeping kernels.
#pragma once
#include <time.h>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/motcpp_amd.h"
#include "ctx.hpp"

namespace mot {
hipError_t launch_kf_op(int op, int kind, const mot_kf_task*, int, int, hipStream_t);
hipError_t launch_kf_update_blocks(const mot_kf_task* tasks, mot_kf_task* fallback, int ntasks, int max_n, hipStream_t st);
hipError_t launch_det(int kind, const mot_det_task*, int, int, hipStream_t);
hipError_t launch_iou(const mot_iou_task*, int, int, int, bool, hipStream_t);
// hint_n / hint_m (0: none): sizes most problems of the launch stay within, tighter than the hard bounds max_n / max_m — the sparse
// solver sizes its LDS with them (more problems per CU) and leaves a problem that exceeds them to the exact solver
hipError_t launch_lap(const mot_lap_task*, int, int, int, bool, bool, bool, hipStream_t, int hint_n = 0, int hint_m = 0, bool try_fast = true,
                      int** declined_out = nullptr, hipEvent_t mid_event = nullptr, int* prezeroed = nullptr,
                      int active_tasks = 0);
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
  int* maxt_zero = nullptr;  // == maxt when the maxima are to be cleared once they are in dev (the next frame then needs no memset launch)
  int n_maxt = 0, maxt_at = 0, counts_at = 0;
  int* counts_copy = nullptr;
  const int* alive = nullptr;  // optional [S]: live tracks per stream after the frame (pooled trackers size their next frame with it)
  int alive_at = 0;
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
    for (int i = t; i < pm.n_maxt; i += 1024) { pm.dev[pm.maxt_at + i] = pm.maxt[i]; if (pm.maxt_zero) pm.maxt_zero[i] = 0; }
    for (int i = t; i < S; i += 1024) { const int c = counts[i]; pm.dev[pm.counts_at + i] = c; if (pm.counts_copy) pm.counts_copy[i] = c; }
    if (pm.alive) for (int i = t; i < S; i += 1024) pm.dev[pm.alive_at + i] = pm.alive[i];
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

// pack_offsets + pack_rows in one launch for a batch of up to a thousand streams (a frame of one camera is a chain of dependent launches:
// one less is one launch latency less): every workgroup sums the counts in front of its stream itself, workgroup 0 also writes the
// offsets and the meta words. Same results as the two kernels.
[[maybe_unused]] static __global__ void __launch_bounds__(256) pack_small(const float* stage, int cap_stage, const int* counts, int S, int* offsets, float* packed,
                                                                         int packed_cap, PackMeta pm) {
  __shared__ int part[4];
  __shared__ int s_base;
  const int s = blockIdx.x, t = static_cast<int>(threadIdx.x);
  int a = 0;
  for (int i = t; i < s; i += 256) { const int c = counts[i]; a += (c > 0) ? c : 0; }
  for (int d = 32; d > 0; d >>= 1) a += __shfl_down(a, d, 64);
  if ((t & 63) == 0) part[t >> 6] = a;
  __syncthreads();
  if (t == 0) s_base = part[0] + part[1] + part[2] + part[3];
  __syncthreads();
  const int o = s_base;
  const int c = counts[s];
  if (t == 0) offsets[s] = o;
  if (s == S - 1 && t == 0) offsets[S] = o + ((c > 0) ? c : 0);
  if (s == 0 && pm.dev) {  // the meta words (pack_offsets' second half); the total is the last workgroup's to know - summed here once more
    int tot = 0;
    for (int i = t; i < S; i += 256) { const int ci = counts[i]; tot += (ci > 0) ? ci : 0; }
    for (int d = 32; d > 0; d >>= 1) tot += __shfl_down(tot, d, 64);
    __syncthreads();
    if ((t & 63) == 0) part[t >> 6] = tot;
    __syncthreads();
    if (t == 0) {
      pm.dev[0] = part[0] + part[1] + part[2] + part[3];
      pm.dev[1] = *pm.err;
      if (pm.dec_at >= 0)
        for (int k = 0; k < 3; ++k) pm.dev[pm.dec_at + k] = pm.dec[k] ? *pm.dec[k] : -1;
    }
    for (int i = t; i < pm.n_maxt; i += 256) { pm.dev[pm.maxt_at + i] = pm.maxt[i]; if (pm.maxt_zero) pm.maxt_zero[i] = 0; }
    for (int i = t; i < S; i += 256) { const int ci = counts[i]; pm.dev[pm.counts_at + i] = ci; if (pm.counts_copy) pm.counts_copy[i] = ci; }
    if (pm.alive) for (int i = t; i < S; i += 256) pm.dev[pm.alive_at + i] = pm.alive[i];
  }
  if (c <= 0) return;
  if (o + c > packed_cap) return;
  const float4* src = reinterpret_cast<const float4*>(stage + static_cast<size_t>(s) * cap_stage * 8);
  float4* dst = reinterpret_cast<float4*>(packed + static_cast<size_t>(o) * 8);
  for (int i = t; i < 2 * c; i += 256) dst[i] = src[i];
}

// The frame's meta words, device -> page-locked host. hipMemcpyAsync takes a much slower route for a device-to-host copy above 16 KB
// (measured: one 16.6 KB copy per frame took SORT from 10.6 M to 6.4 M frames/s and ByteTrack 256 x 128 from 9.1 M to 5.5 M, with the kernels
// unchanged), so the buffer goes in pieces of at most 16 KB.
inline hipError_t copy_meta_d2h(int* h_dst, const int* d_src, size_t n_ints, hipStream_t st) {
  static const size_t piece = [] {
    const char* e = std::getenv("MOT_META_PIECE");
    const long v = e ? std::atol(e) : 4096;
    return static_cast<size_t>(v > 0 ? v : 4096);  // (a value that is not a positive number would never advance the loop below)
  }();  // ints
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

// ---- a frame's inputs as the bookkeeping kernels see them ---------------------------------------------------------------
// Classic form (mot_*_step*, mot_*_enqueue_packed): every stream takes part, stream s's detections are the SoA planes
// [6][max_dets] at d_dets + s * 6 * max_dets — only `counts` is set. Pooled form (mot_*_enqueue_frame, round 4): the streams of a
// batch belong to independent tracker objects, only some of them have a frame to process, and their detections arrive packed back
// to back: counts[s] < 0 = stream s sits this frame out (its state is not touched, it reports 0 rows), det_off[s] / ld[s] = where
// its planes [6][ld] start and how long they are. One device block per batch, refilled by one copy per frame.
struct FrameDev {
  const int* counts = nullptr;
  const int* ld = nullptr;
  const long long* det_off = nullptr;
  const long long* emb_off = nullptr;  // BoT-SORT: float offset of the stream's [n][E] feature rows, < 0: none this frame
};
__device__ __forceinline__ const float* frame_dets(const FrameDev& F, const float* base, int s, int D, int& ld) {
  if (F.det_off) { ld = F.ld[s]; return base + F.det_off[s]; }
  ld = D;
  return base + static_cast<size_t>(s) * 6 * D;
}
// host image of the block: [counts S][ld S] ints, then [det_off S][emb_off S] long longs (8-byte aligned: 2 S ints precede)
inline size_t frame_block_ints(int S) { return static_cast<size_t>(S) * 2 + static_cast<size_t>(S) * 4; }

// ---- frames in flight (round 3: SORT / OC-SORT; round 4: all four lifecycles share this one copy) ------------------------
// step_packed split in two so that ONE host thread overlaps the result copy of frame f with the kernels of frame f + 1: enqueue(f + 1)
// returns once its launches are queued, collect(f) waits for frame f's event only and copies its rows on a second stream. Two sets of
// packed tables; what the host needs back from a frame travels in page-locked memory:
//   h_meta: [0] total rows, [1] error flag, [2..5) problems the sparse solver declined in up to three associations (-1: not counted),
//           [5..5 + n_maxt) the frame's per-stream maxima (64 words: live tracks; ByteTrack / BoT-SORT keep three more sets of 64: the
//           largest problems of their associations), then counts out [S], live tracks per stream [S] (alive != nullptr), and — host
//           only — counts in [S], extra_host ints per stream (BoT-SORT: has_warp + 6 warp floats), the pooled input block
// Zero-copy rows (view mode, what the pooled trackers use): pack_rows writes the packed table straight into page-locked host memory
// (h_rows), so a frame needs no device-to-host copy at all and collect is one event wait.
constexpr int kMetaDec = 2, kMetaMaxt = 5;
constexpr size_t kZeroCopyBytes = 32 * 1024;  // inputs up to this size are read in place from page-locked memory
struct Flight {
  float* d_packed = nullptr; int* d_offsets = nullptr; int* d_counts = nullptr; int packed_cap = 0;
  float* h_rows = nullptr; int h_rows_cap = 0;  // view mode: the packed rows in page-locked memory, written by the kernel
  int* h_meta = nullptr;
  int* d_meta = nullptr;  // device image of h_meta's device-filled words
  int* h_in = nullptr;    // pooled form: page-locked image of the frame's input block
  hipEvent_t done = nullptr;
  hipEvent_t ev[12] = {};
  bool pending = false, prof = false, view = false;
  bool polite = false;  // the host waits for this frame with sleeps between polls instead of spinning (see Flights::pop)
  int rows_cap = 0;  // the row limit pack_rows ran with (the enqueue call's rows_cap)
  int bd = 0;        // largest detection count of the frame (bounds the tracks it may add)
};
struct Flights {
  Flight fl[2];
  int head = 0, count = 0;  // oldest pending frame, frames pending
  int n_maxt = 64;          // words of per-frame maxima the lifecycle keeps
  int extra_host = 0;       // page-locked ints per stream behind counts_in
  bool with_alive = false;  // the lifecycle reports the live tracks per stream
  hipStream_t copy_st = nullptr;
  int* d_in = nullptr;      // the pooled input block on the device
  // the per-frame maxima are all zero (pack_offsets of the last frame cleared them): the lifecycle's frame needs no memset in front.
  // Cleared by a lifecycle whenever it starts a frame, set when a flight's pack_offsets is queued.
  bool maxt_clean = false;
  int meta_head() const { return kMetaMaxt + n_maxt; }
  int meta_dev_words(int S) const { return meta_head() + (with_alive ? 2 : 1) * S; }
  int slot_for_enqueue() const { return (head + count) & 1; }
  const int* maxt_of(const Flight& F) const { return F.h_meta + kMetaMaxt; }
  const int* counts_of(const Flight& F) const { return F.h_meta + meta_head(); }
  const int* alive_of(const Flight& F, int S) const { return with_alive ? F.h_meta + meta_head() + S : nullptr; }
  int* counts_in_of(Flight& F, int S) const { return F.h_meta + meta_dev_words(S); }
  int* extra_of(Flight& F, int S) const { return F.h_meta + meta_dev_words(S) + S; }
  // buffers of the slot (allocated on first use / when rows_cap grows); counts_in = page-locked copy of the caller's counts
  hipError_t prepare(Allocs& mem, int slot, int S, int rows_cap, const int* h_counts, int** counts_in, int* bd_out, bool view = false, bool prof = false) {
    Flight& F = fl[slot];
    hipError_t e = hipSuccess;
    if (!copy_st && (e = hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking)) != hipSuccess) return e;
    if (!F.done && (e = hipEventCreateWithFlags(&F.done, hipEventDisableTiming)) != hipSuccess) return e;
    if (!F.h_meta && (e = hipHostMalloc(reinterpret_cast<void**>(&F.h_meta), sizeof(int) * (meta_dev_words(S) + static_cast<size_t>(1 + extra_host) * S), hipHostMallocDefault)) != hipSuccess) return e;
    if (!F.d_offsets) { F.d_offsets = mem.get<int>(static_cast<size_t>(S) + 1); F.d_counts = mem.get<int>(S); F.d_meta = mem.get<int>(meta_dev_words(S)); }
    if (!F.d_offsets || !F.d_counts || !F.d_meta) return hipErrorOutOfMemory;
    if (view) {
      if (rows_cap > F.h_rows_cap) {
        if (F.h_rows) (void)hipHostFree(F.h_rows);
        F.h_rows = nullptr; F.h_rows_cap = 0;
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&F.h_rows), sizeof(float) * 8 * static_cast<size_t>(rows_cap), hipHostMallocDefault)) != hipSuccess) return e;
        F.h_rows_cap = rows_cap;
      }
    } else if (rows_cap > F.packed_cap) {
      F.d_packed = mem.get<float>(static_cast<size_t>(rows_cap) * 8); F.packed_cap = F.d_packed ? rows_cap : 0;
      if (!F.d_packed) return hipErrorOutOfMemory;
    }
    F.view = view;
    if (prof && !F.ev[0]) for (auto& ev : F.ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return e;
    F.prof = prof;
    int* ci = counts_in_of(F, S);
    int bd = 1, active = 0;
    for (int s = 0; s < S; ++s) { ci[s] = h_counts[s]; bd = (h_counts[s] > bd) ? h_counts[s] : bd; active += h_counts[s] >= 0 ? 1 : 0; }
    *counts_in = ci; *bd_out = bd;
    // Round 6. A pooled round of many tracker objects means as many host threads that want a CPU the moment the round ends, on boxes whose
    // CPU quota (16 CPUs for 256 hardware threads on the MI355X boxes) is the scarce resource: a leader that SPINS on the frame's event for the
    // whole GPU time of the round — hipEventSynchronize's default — burns one CPU per segment, a quarter of the quota with four segments, and
    // the cgroup then throttles everybody for the rest of its 100 ms period (measured: 1024 objects, p99 of update() 78 ms against a median of
    // 1.0). From 48 streams on, the wait polls the event with 30 us sleeps in between. A handful of cameras keep the spin: their frame is
    // 0.2 ms and a sleep's wake-up latency would show. MOT_POLITE_WAIT=0 / 1 forces either.
    static const int polite_env = [] { const char* e = std::getenv("MOT_POLITE_WAIT"); return (e && *e) ? std::atoi(e) : -1; }();
    F.polite = view && (polite_env >= 0 ? polite_env != 0 : active >= 48);
    return hipSuccess;
  }
  // pooled form: fills the slot's page-locked input block from the caller's arrays, queues its copy to the device and returns the
  // device view. emb_off may be nullptr (every stream: none).
  hipError_t upload_block(Allocs& mem, int slot, int S, const int* counts, const int* ld, const long long* det_off, const long long* emb_off,
                          hipStream_t st, FrameDev* out) {
    Flight& F = fl[slot];
    hipError_t e = hipSuccess;
    const size_t words = frame_block_ints(S);
    if (!F.h_in && (e = hipHostMalloc(reinterpret_cast<void**>(&F.h_in), sizeof(int) * words, hipHostMallocDefault)) != hipSuccess) return e;
    if (!d_in) { d_in = mem.get<int>(words); if (!d_in) return hipErrorOutOfMemory; }
    int* hc = F.h_in; int* hl = hc + S;
    long long* hd = reinterpret_cast<long long*>(hl + S); long long* he = hd + S;
    for (int s = 0; s < S; ++s) { hc[s] = counts[s]; hl[s] = ld[s]; hd[s] = det_off[s]; he[s] = emb_off ? emb_off[s] : -1; }
    // a small block is read by the kernels where it lies (page-locked memory is mapped into the device's address space): one copy launch
    // less in front of a single camera's frame; a large one (thousands of streams, one 64-byte PCIe read per workgroup) is copied
    const int* blk = F.h_in;
    if (words * sizeof(int) > kZeroCopyBytes) {
      if ((e = hipMemcpyAsync(d_in, F.h_in, sizeof(int) * words, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
      blk = d_in;
    }
    out->counts = blk; out->ld = blk + S;
    out->det_off = reinterpret_cast<const long long*>(blk + 2 * static_cast<size_t>(S)); out->emb_off = out->det_off + S;
    return hipSuccess;
  }
  float* rows_target(int slot) { Flight& F = fl[slot]; return F.view ? F.h_rows : F.d_packed; }
  PackMeta pack_meta(int slot, const int* d_err, const int* d_maxt, const int* d_declined, const int* d_declined_b = nullptr, const int* d_declined_c = nullptr) {
    Flight& F = fl[slot];
    PackMeta pm;
    // view mode: the kernel writes the words straight into the page-locked buffer (like the rows) - no copy behind the frame
    pm.dev = F.view ? F.h_meta : F.d_meta; pm.err = d_err; pm.dec[0] = d_declined; pm.dec[1] = d_declined_b; pm.dec[2] = d_declined_c; pm.dec_at = kMetaDec;
    pm.maxt = d_maxt; pm.maxt_zero = const_cast<int*>(d_maxt); pm.n_maxt = n_maxt; pm.maxt_at = kMetaMaxt; pm.counts_at = meta_head(); pm.counts_copy = F.d_counts;
    return pm;
  }
  // behind the frame's launches: pack the staged tables, bring the small results back, record the frame's event
  hipError_t finish(int slot, hipStream_t st, const float* d_stage, int cap_stage, const int* d_out_counts, int S, const int* d_err, const int* d_maxt,
                    const int* d_declined, int rows_cap, int bd, const int* d_declined_b = nullptr, const int* d_declined_c = nullptr,
                    const int* d_alive = nullptr) {
    Flight& F = fl[slot];
    PackMeta pm = pack_meta(slot, d_err, d_maxt, d_declined, d_declined_b, d_declined_c);
    pm.alive = with_alive ? d_alive : nullptr; pm.alive_at = meta_head() + S;
    if (S <= 1024) hipLaunchKernelGGL(pack_small, dim3(S), dim3(256), 0, st, d_stage, cap_stage, d_out_counts, S, F.d_offsets, rows_target(slot), rows_cap, pm);
    else {
      hipLaunchKernelGGL(pack_offsets, dim3(1), dim3(1024), 0, st, d_out_counts, S, F.d_offsets, pm);
      hipLaunchKernelGGL(pack_rows, dim3(S), dim3(256), 0, st, d_stage, cap_stage, d_out_counts, F.d_offsets, rows_target(slot), rows_cap);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return finish_copies(slot, st, S, rows_cap, bd);
  }
  // (the lifecycles that pack inside their own launch sequence call this part only)
  hipError_t finish_copies(int slot, hipStream_t st, int S, int rows_cap, int bd) {
    Flight& F = fl[slot];
    hipError_t e = hipSuccess;
    if (!F.view && (e = copy_meta_d2h(F.h_meta, F.d_meta, static_cast<size_t>(meta_dev_words(S)), st)) != hipSuccess) return e;
    if ((e = hipEventRecord(F.done, st)) != hipSuccess) return e;
    maxt_clean = true;
    F.pending = true; F.bd = bd; F.rows_cap = rows_cap;
    count += 1;
    return hipSuccess;
  }
  // the oldest pending frame: waits for it, hands out its meta words; the caller copies the rows with copy_rows
  hipError_t pop(Flight** out) {
    Flight& F = fl[head];
    hipError_t e = hipSuccess;
    if (F.polite) {
      for (;;) {
        e = hipEventQuery(F.done);
        if (e != hipErrorNotReady) break;
        struct timespec ts = {0, 30000};
        nanosleep(&ts, nullptr);
      }
    } else e = hipEventSynchronize(F.done);
    if (e != hipSuccess) return e;
    F.pending = false;
    head ^= 1; count -= 1;
    *out = &F;
    return hipSuccess;
  }
  hipError_t copy_rows(const Flight& F, float* rows, int total) {
    if (total <= 0) return hipSuccess;
    if (F.view) { std::memcpy(rows, F.h_rows, sizeof(float) * static_cast<size_t>(total) * 8); return hipSuccess; }
    const hipError_t e = hipMemcpyAsync(rows, F.d_packed, sizeof(float) * static_cast<size_t>(total) * 8, hipMemcpyDeviceToHost, copy_st);
    return (e != hipSuccess) ? e : hipStreamSynchronize(copy_st);
  }
  int pending_bd() const { int v = 0; for (const Flight& F : fl) if (F.pending && F.bd > v) v = F.bd; return v; }
  void drop_all() { for (Flight& F : fl) F.pending = false; head = 0; count = 0; }
  void release() {
    for (Flight& F : fl) {
      if (F.done) (void)hipEventDestroy(F.done);
      if (F.h_meta) (void)hipHostFree(F.h_meta);
      if (F.h_in) (void)hipHostFree(F.h_in);
      if (F.h_rows) (void)hipHostFree(F.h_rows);
      for (auto& e : F.ev) if (e) (void)hipEventDestroy(e);
      F = Flight{};
    }
    if (copy_st) (void)hipStreamDestroy(copy_st);
    copy_st = nullptr;
  }
};

// one stream's persistent record back to its state at creation (a pooled tracker object's reset() / a slot handed to a new object);
// keep_ids: the id counter keeps running (Sort::reset, sort.cpp:97-100)
template <class StreamT>
static __global__ void reset_stream_kernel(StreamT* streams, int s, StreamT fresh, int keep_ids, int* batch_err) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    // the batch's error word starts over with the stream: whichever other stream is still in error raises it again in its next frame (every
    // stream reports its own sticky word per frame), so a tracker object that was reset after a capacity error does not keep failing
    // everybody's rounds (round 6; the pooled trackers read the per-stream words, mot_frame_view.alive)
    if (batch_err) *batch_err = 0;
    // The record is written ONCE, with the counter already in it. (Round 5: `streams[s] = fresh; streams[s].next_id = saved;` lost the counter for the
    // first objects of a process — a 16-byte store and a later 4-byte store of the same wavefront to the same word, completed out of order while the
    // page was cold. Measured: plain read + single store 0 of 12 fresh interpreters, the two stores 8 of 12. Two stores of one thread to
    // overlapping addresses need a wait between them; nothing else in the library does that.)
    StreamT f = fresh;
    if (keep_ids) f.next_id = __atomic_load_n(&streams[s].next_id, __ATOMIC_RELAXED);
    streams[s] = f;
  }
}
// arrays of a stream that outlive a frame, copied element-wise when a stream moves to a batch with larger capacities
template <class T>
__device__ __forceinline__ void move_array(T* dst, const T* src, size_t n) {
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

}  // namespace lifecycle
}  // namespace mot

// HIP call inside a function returning a mot_status, with the batch's context recording the message
#define MOT_LC_HIP(b, call)                                                                                                \
  do {                                                                                                                     \
    hipError_t e_ = (call);                                                                                                \
    if (e_ != hipSuccess) { (b)->ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return MOT_ERR_HIP; }       \
  } while (0)
