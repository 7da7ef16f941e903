// C ABI implementation (include/motcpp_amd.h): context/stream/memory plumbing, launch wrappers
// for the gfx950 kernels and the synchronous host-pointer conveniences. No CPU compute path
// exists in this library: every entry point either launches a HIP kernel or fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/motcpp_amd.h"
#include "lap_core.hpp"

namespace mot {
hipError_t launch_kf_op(int op, int kind, const mot_kf_task*, int, int, hipStream_t);
hipError_t launch_kf_update_blocks(const mot_kf_task* tasks, mot_kf_task* fallback, int ntasks, int max_n, hipStream_t st);
hipError_t launch_det(int kind, const mot_det_task*, int, int, hipStream_t);
hipError_t launch_gate(int kind, const mot_gate_task*, int, int, int, hipStream_t);
hipError_t launch_iou(const mot_iou_task*, int, int, int, bool, hipStream_t);
hipError_t launch_ocsort(const mot_ocsort_task*, int, int, int, bool, hipStream_t);
hipError_t launch_feat(const mot_feat_task*, int, int, hipStream_t);
hipError_t launch_ss_nn(const mot_ss_nn_task*, int, int, int, hipStream_t);
hipError_t launch_ss_iou(const mot_ss_iou_task*, int, int, int, hipStream_t);
hipError_t launch_ucmc(int, const mot_ucmc_task*, int, int, int, hipStream_t);
hipError_t launch_boost(int, const mot_boost_task*, int, int, int, hipStream_t);
hipError_t launch_hyb(int, const mot_hyb_task*, int, int, int, hipStream_t);
hipError_t launch_deep(const mot_deep_task*, int, int, int, hipStream_t);
hipError_t launch_embed(int metric, const mot_cos_task*, int, int, int, hipStream_t);
hipError_t launch_cosine(const mot_cos_task*, int, int, int, hipStream_t);
// hint_n / hint_m (0: none): sizes most problems of the launch stay within, tighter than the hard bounds max_n / max_m — the sparse
// solver sizes its LDS with them (more problems per CU) and leaves a problem that exceeds them to the exact solver
hipError_t launch_lap(const mot_lap_task*, int, int, int, bool, bool, bool, hipStream_t, int hint_n = 0, int hint_m = 0, bool try_fast = true,
                      int** declined_out = nullptr, hipEvent_t mid_event = nullptr, int* prezeroed = nullptr,
                      int active_tasks = 0);
size_t lap_scratch_bytes(int n, int m);
size_t lap_rowlist_scratch_bytes(int n);
hipError_t lap_fast_stats(unsigned long long* out16, bool reset, hipStream_t st);
hipError_t lap_behind_stats(long long* out80, bool reset, hipStream_t st);
hipError_t launch_embed_gated(const mot_cos_task*, const mot_lap_task*, int, int, int, int, hipStream_t);
}  // namespace mot

#include "ctx.hpp"

namespace {
int fail(mot_ctx* c, hipError_t e, const char* what) {
  if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
  return MOT_ERR_HIP;
}
#define MOT_HIP(c, call)                                    \
  do {                                                      \
    hipError_t e__ = (call);                                \
    if (e__ != hipSuccess) return fail((c), e__, #call);    \
  } while (0)

// RAII device buffer for the _host conveniences
struct DBuf {
  void* p = nullptr;
  ~DBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

namespace mot { hipError_t lap_sparse_timeline(unsigned long long*, int, hipStream_t); }
extern "C" {

const char* mot_version(void) { return "motcpp_amd 0.1 (gfx950)"; }

int mot_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mot_ctx_create(int device, void* hip_stream, mot_ctx** out) {
  if (!out) return MOT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MOT_ERR_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MOT_ERR_NODEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MOT_ERR_NODEVICE;  // kernels are built for gfx950 only
  if (hipSetDevice(device) != hipSuccess) return MOT_ERR_HIP;
  mot_ctx* c = new mot_ctx();
  c->device = device;
  if (hip_stream) c->stream = static_cast<hipStream_t>(hip_stream);
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MOT_ERR_HIP; }
    c->own_stream = true;
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return MOT_ERR_HIP; }
  *out = c;
  return MOT_OK;
}
int mot_ctx_destroy(mot_ctx* c) {
  if (!c) return MOT_OK;
  (void)hipStreamSynchronize(c->stream);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MOT_OK;
}
int mot_ctx_bind(mot_ctx* c) { if (!c) return MOT_ERR_INVALID; MOT_HIP(c, hipSetDevice(c->device)); return MOT_OK; }
int mot_ctx_sync(mot_ctx* c) { MOT_HIP(c, hipStreamSynchronize(c->stream)); return MOT_OK; }
void* mot_ctx_stream(mot_ctx* c) { return c ? c->stream : nullptr; }
const char* mot_ctx_last_error(mot_ctx* c) { return c ? c->err.c_str() : "null context"; }

int mot_malloc(mot_ctx* c, size_t bytes, void** d) { MOT_HIP(c, hipSetDevice(c->device)); MOT_HIP(c, hipMalloc(d, bytes ? bytes : 16)); return MOT_OK; }
int mot_free(mot_ctx* c, void* d) { if (d) MOT_HIP(c, hipFree(d)); return MOT_OK; }
int mot_host_alloc(mot_ctx* c, size_t bytes, void** h) { MOT_HIP(c, hipHostMalloc(h, bytes ? bytes : 16, hipHostMallocDefault)); return MOT_OK; }
int mot_host_free(mot_ctx* c, void* h) { if (h) MOT_HIP(c, hipHostFree(h)); return MOT_OK; }
int mot_memcpy_h2d(mot_ctx* c, void* d, const void* h, size_t b) { if (b) MOT_HIP(c, hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, c->stream)); return MOT_OK; }
int mot_memcpy_d2h(mot_ctx* c, void* h, const void* d, size_t b) { if (b) MOT_HIP(c, hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, c->stream)); return MOT_OK; }
int mot_memcpy_d2d(mot_ctx* c, void* dst, const void* src, size_t b) { if (b) MOT_HIP(c, hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToDevice, c->stream)); return MOT_OK; }
int mot_memset(mot_ctx* c, void* d, int v, size_t b) { if (b) MOT_HIP(c, hipMemsetAsync(d, v, b, c->stream)); return MOT_OK; }
int mot_timer_start(mot_ctx* c) { MOT_HIP(c, hipEventRecord(c->ev0, c->stream)); return MOT_OK; }
int mot_timer_stop(mot_ctx* c, float* ms) {
  MOT_HIP(c, hipEventRecord(c->ev1, c->stream));
  MOT_HIP(c, hipEventSynchronize(c->ev1));
  MOT_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return MOT_OK;
}

int mot_event_create(mot_ctx* c, void** ev) { hipEvent_t e; MOT_HIP(c, hipEventCreate(&e)); *ev = e; return MOT_OK; }
int mot_event_destroy(mot_ctx* c, void* ev) { if (ev) MOT_HIP(c, hipEventDestroy(static_cast<hipEvent_t>(ev))); return MOT_OK; }
int mot_event_record(mot_ctx* c, void* ev) { MOT_HIP(c, hipEventRecord(static_cast<hipEvent_t>(ev), c->stream)); return MOT_OK; }
int mot_event_elapsed(mot_ctx* c, void* a, void* b, float* ms) {
  MOT_HIP(c, hipEventElapsedTime(ms, static_cast<hipEvent_t>(a), static_cast<hipEvent_t>(b)));
  return MOT_OK;
}

int mot_kf_dim(int kind) { return kind == MOT_KF_XYSR ? 7 : 8; }

int mot_det_prepare(mot_ctx* c, int kind, const mot_det_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_det(kind, t, nt, max_n, c->stream)); return MOT_OK; }
int mot_kf_initiate(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_kf_op(0, kind, t, nt, max_n, c->stream)); return MOT_OK; }
int mot_kf_predict(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_kf_op(1, kind, t, nt, max_n, c->stream)); return MOT_OK; }
int mot_kf_predict_boxes(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_kf_op(6, kind, t, nt, max_n, c->stream)); return MOT_OK; }
int mot_kf_update(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_kf_op(2, kind, t, nt, max_n, c->stream)); return MOT_OK; }
int mot_kf_boxes(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_kf_op(3, kind, t, nt, max_n, c->stream)); return MOT_OK; }
int mot_kf_warp(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) {
  if (kind == MOT_KF_XYAH) { c->err = "mot_kf_warp: the XYAH filter has no camera-motion step"; return MOT_ERR_INVALID; }
  MOT_HIP(c, mot::launch_kf_op(4, kind, t, nt, max_n, c->stream));
  return MOT_OK;
}
int mot_kf_predict_warp(mot_ctx* c, int kind, const mot_kf_task* t, int nt, int max_n) {
  if (kind == MOT_KF_XYAH) { c->err = "mot_kf_predict_warp: the XYAH filter has no camera-motion step"; return MOT_ERR_INVALID; }
  MOT_HIP(c, mot::launch_kf_op(5, kind, t, nt, max_n, c->stream));
  return MOT_OK;
}
int mot_iou_cost(mot_ctx* c, const mot_iou_task* t, int nt, int max_n, int max_m) { return mot_iou_cost_ex(c, t, nt, max_n, max_m, 0); }
int mot_iou_cost_ex(mot_ctx* c, const mot_iou_task* t, int nt, int max_n, int max_m, int flags) {
  MOT_HIP(c, mot::launch_iou(t, nt, max_n, max_m, (flags & MOT_COST_F_IOU_ONLY) != 0, c->stream));
  return MOT_OK;
}
int mot_ocsort_cost(mot_ctx* c, const mot_ocsort_task* t, int nt, int max_nd, int max_nt) { return mot_ocsort_cost_ex(c, t, nt, max_nd, max_nt, 0); }
int mot_ocsort_cost_ex(mot_ctx* c, const mot_ocsort_task* t, int nt, int max_nd, int max_nt, int flags) {
  MOT_HIP(c, mot::launch_ocsort(t, nt, max_nd, max_nt, (flags & MOT_COST_F_IOU_ONLY) != 0, c->stream));
  return MOT_OK;
}
int mot_cosine_cost(mot_ctx* c, const mot_cos_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_cosine(t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_deepoc_cost(mot_ctx* c, const mot_deep_task* t, int nt, int max_nd, int max_nt) { MOT_HIP(c, mot::launch_deep(t, nt, max_nd, max_nt, c->stream)); return MOT_OK; }
int mot_ss_nn_cost(mot_ctx* c, const mot_ss_nn_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_ss_nn(t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_ss_iou_cost(mot_ctx* c, const mot_ss_iou_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_ss_iou(t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_hyb_run(mot_ctx* c, int op, const mot_hyb_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_hyb(op, t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_boost_run(mot_ctx* c, int op, const mot_boost_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_boost(op, t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_ucmc_run(mot_ctx* c, int op, const mot_ucmc_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_ucmc(op, t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_feat_update(mot_ctx* c, const mot_feat_task* t, int nt, int max_n) { MOT_HIP(c, mot::launch_feat(t, nt, max_n, c->stream)); return MOT_OK; }
int mot_lap_behind_stats(mot_ctx* c, long long* out80, int reset) { MOT_HIP(c, mot::lap_behind_stats(out80, reset != 0, c->stream)); return MOT_OK; }
int mot_debug_sparse_timeline(mot_ctx* c, unsigned long long* out, int n) { MOT_HIP(c, mot::lap_sparse_timeline(out, n, c->stream)); return MOT_OK; }
int mot_lap_fast_stats(mot_ctx* c, unsigned long long* out16, int reset) { MOT_HIP(c, mot::lap_fast_stats(out16, reset != 0, c->stream)); return MOT_OK; }
size_t mot_lap_work_bytes(int n, int m) { return (mot::lap_scratch_bytes(n, m) + 255) & ~size_t(255); }
size_t mot_lap_rowlist_bytes(int n) { return (mot::lap_rowlist_scratch_bytes(n) + 255) & ~size_t(255); }
int mot_lap_solve(mot_ctx* c, const mot_lap_task* t, int nt, int max_n, int max_m, int flags) { MOT_HIP(c, mot::launch_lap(t, nt, max_n, max_m, (flags & MOT_LAP_F_GEOM) != 0, (flags & MOT_LAP_F_ASSOC) != 0, (flags & MOT_LAP_F_PLAIN) != 0, c->stream)); return MOT_OK; }

// ---- host-pointer conveniences ------------------------------------------------------------------
static void to_soa4(const float* aos, int n, int cols, int stride, std::vector<float>& soa) {
  soa.assign(static_cast<size_t>(cols) * (n ? n : 1), 0.f);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < cols; ++k) soa[static_cast<size_t>(k) * n + i] = aos[static_cast<size_t>(i) * stride + k];
}

int mot_iou_cost_host(mot_ctx* c, const float* a, int n, const float* b, int m, const float* bconf, int mode, float* cost) {
  return mot_assoc_cost_host(c, a, n, b, m, bconf, mode, MOT_ASSOC_IOU, 1, 1, cost);
}
int mot_assoc_cost_host(mot_ctx* c, const float* a, int n, const float* b, int m, const float* bconf, int mode, int assoc,
                        int frame_w, int frame_h, float* cost) {
  if (n <= 0 || m <= 0) return MOT_OK;
  std::vector<float> sa, sb;
  to_soa4(a, n, 4, 4, sa);
  to_soa4(b, m, 4, 4, sb);
  DBuf da, db, dc, dcost, dt;
  MOT_HIP(c, da.alloc(sa.size() * 4)); MOT_HIP(c, db.alloc(sb.size() * 4)); MOT_HIP(c, dc.alloc(m * 4));
  MOT_HIP(c, dcost.alloc(static_cast<size_t>(n) * m * 4)); MOT_HIP(c, dt.alloc(sizeof(mot_iou_task)));
  MOT_HIP(c, hipMemcpyAsync(da.p, sa.data(), sa.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(db.p, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (bconf) MOT_HIP(c, hipMemcpyAsync(dc.p, bconf, m * 4, hipMemcpyHostToDevice, c->stream));
  mot_iou_task t{};
  t.n = n; t.m = m; t.a = da.as<float>(); t.lda = n; t.b = db.as<float>(); t.ldb = m;
  t.bconf = bconf ? dc.as<float>() : nullptr; t.cost = dcost.as<float>(); t.ldc = m; t.mode = mode;
  t.assoc = assoc;
  t.frame_diag = static_cast<float>(sqrt(static_cast<double>(frame_w * frame_w + frame_h * frame_h)));  // iou.hpp:329
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_iou(dt.as<mot_iou_task>(), 1, n, m, assoc == MOT_ASSOC_IOU, c->stream));
  MOT_HIP(c, hipMemcpyAsync(cost, dcost.p, static_cast<size_t>(n) * m * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

// fuse_iou (matching.cpp:109-128): the ReID cost matrix blended with the IoU of the same pairs
int mot_fuse_iou_host(mot_ctx* c, const float* reid_cost, const float* a, int n, const float* b, int m, float* cost) {
  if (n <= 0 || m <= 0) return MOT_OK;
  std::vector<float> sa, sb;
  to_soa4(a, n, 4, 4, sa);
  to_soa4(b, m, 4, 4, sb);
  DBuf da, db, de, dcost, dt;
  MOT_HIP(c, da.alloc(sa.size() * 4)); MOT_HIP(c, db.alloc(sb.size() * 4)); MOT_HIP(c, de.alloc(static_cast<size_t>(n) * m * 4));
  MOT_HIP(c, dcost.alloc(static_cast<size_t>(n) * m * 4)); MOT_HIP(c, dt.alloc(sizeof(mot_iou_task)));
  MOT_HIP(c, hipMemcpyAsync(da.p, sa.data(), sa.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(db.p, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(de.p, reid_cost, static_cast<size_t>(n) * m * 4, hipMemcpyHostToDevice, c->stream));
  mot_iou_task t{};
  t.n = n; t.m = m; t.a = da.as<float>(); t.lda = n; t.b = db.as<float>(); t.ldb = m;
  t.cost = dcost.as<float>(); t.ldc = m; t.mode = MOT_COST_FUSE_IOU; t.emb = de.as<float>(); t.lde = m;
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_iou(dt.as<mot_iou_task>(), 1, n, m, true, c->stream));
  MOT_HIP(c, hipMemcpyAsync(cost, dcost.p, static_cast<size_t>(n) * m * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

int mot_gate_cost(mot_ctx* c, int kind, const mot_gate_task* t, int nt, int max_n, int max_m) {
  MOT_HIP(c, mot::launch_gate(kind, t, nt, max_n, max_m, c->stream));
  return MOT_OK;
}
int mot_gate_cost_host(mot_ctx* c, int kind, int mode, int n, int m, const float* mean8, const float* cov64, const float* meas4,
                       const float* cost, int only_position, int metric, float lambda, float gated_cost, float* out) {
  if (n <= 0 || m <= 0) return MOT_OK;
  if ((kind != MOT_KF_XYAH && kind != MOT_KF_XYWH) || mode < 0 || mode > 2 || (mode != 0 && !cost)) return MOT_ERR_INVALID;
  std::vector<float> rec(static_cast<size_t>(n) * 72), sm;
  for (int i = 0; i < n; ++i) {
    std::memcpy(&rec[static_cast<size_t>(i) * 72], mean8 + static_cast<size_t>(i) * 8, 8 * sizeof(float));
    std::memcpy(&rec[static_cast<size_t>(i) * 72 + 8], cov64 + static_cast<size_t>(i) * 64, 64 * sizeof(float));
  }
  to_soa4(meas4, m, 4, 4, sm);
  DBuf dr, dm, dc, dout, dt;
  MOT_HIP(c, dr.alloc(rec.size() * 4)); MOT_HIP(c, dm.alloc(sm.size() * 4)); MOT_HIP(c, dc.alloc(static_cast<size_t>(n) * m * 4));
  MOT_HIP(c, dout.alloc(static_cast<size_t>(n) * m * 4)); MOT_HIP(c, dt.alloc(sizeof(mot_gate_task)));
  MOT_HIP(c, hipMemcpyAsync(dr.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dm.p, sm.data(), sm.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (cost) MOT_HIP(c, hipMemcpyAsync(dc.p, cost, static_cast<size_t>(n) * m * 4, hipMemcpyHostToDevice, c->stream));
  mot_gate_task t{};
  t.n = n; t.m = m; t.mean = dr.as<float>(); t.src = nullptr; t.meas = dm.as<float>(); t.ldm = m;
  t.cost = dc.as<float>(); t.ldc = m; t.out = dout.as<float>(); t.ldo = m;
  t.mode = mode; t.only_position = only_position; t.metric = metric; t.lambda = lambda; t.gated_cost = gated_cost;
  for (int r0 = 0; r0 < n; r0 += 32768) {  // grid.y limit: at most 32768 track rows per launch
    mot_gate_task tt = t;
    tt.n = (n - r0 < 32768) ? n - r0 : 32768;
    tt.mean = t.mean + static_cast<size_t>(r0) * 72; tt.cost = t.cost + static_cast<size_t>(r0) * m; tt.out = t.out + static_cast<size_t>(r0) * m;
    MOT_HIP(c, hipMemcpyAsync(dt.p, &tt, sizeof(tt), hipMemcpyHostToDevice, c->stream));
    MOT_HIP(c, hipStreamSynchronize(c->stream));  // tt is a local
    MOT_HIP(c, mot::launch_gate(kind, dt.as<mot_gate_task>(), 1, tt.n, m, c->stream));
    MOT_HIP(c, hipStreamSynchronize(c->stream));  // the descriptor is rewritten by the next slice
  }
  MOT_HIP(c, hipMemcpyAsync(out, dout.p, static_cast<size_t>(n) * m * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

int mot_embedding_cost(mot_ctx* c, int metric, const mot_cos_task* t, int nt, int max_n, int max_m) { MOT_HIP(c, mot::launch_embed(metric, t, nt, max_n, max_m, c->stream)); return MOT_OK; }
int mot_cosine_cost_host(mot_ctx* c, const float* a, int n, const float* b, int m, int d, float* out) { return mot_embedding_cost_host(c, MOT_EMB_COSINE, a, n, b, m, d, out); }
int mot_embedding_cost_host(mot_ctx* c, int metric, const float* a, int n, const float* b, int m, int d, float* out) {
  if (n <= 0 || m <= 0) return MOT_OK;
  if (metric < 0 || metric > 2) return MOT_ERR_INVALID;
  DBuf da, db, dout, dna, dnb, dt;
  MOT_HIP(c, da.alloc(static_cast<size_t>(n) * d * 4)); MOT_HIP(c, db.alloc(static_cast<size_t>(m) * d * 4));
  MOT_HIP(c, dout.alloc(static_cast<size_t>(n) * m * 4)); MOT_HIP(c, dna.alloc(n * 4)); MOT_HIP(c, dnb.alloc(m * 4));
  MOT_HIP(c, dt.alloc(sizeof(mot_cos_task)));
  MOT_HIP(c, hipMemcpyAsync(da.p, a, static_cast<size_t>(n) * d * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(db.p, b, static_cast<size_t>(m) * d * 4, hipMemcpyHostToDevice, c->stream));
  mot_cos_task t{};
  t.n = n; t.m = m; t.d = d; t.a = da.as<float>(); t.lda = d; t.b = db.as<float>(); t.ldb = d;
  t.out = dout.as<float>(); t.ldo = m; t.norm_a = dna.as<float>(); t.norm_b = dnb.as<float>();
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_embed(metric, dt.as<mot_cos_task>(), 1, n, m, c->stream));
  MOT_HIP(c, hipMemcpyAsync(out, dout.p, static_cast<size_t>(n) * m * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

int mot_cosine_cost_gated(mot_ctx* c, const mot_cos_task* t, const mot_lap_task* lap, int lap_stride, int nt, int max_n, int max_m) {
  MOT_HIP(c, mot::launch_embed_gated(t, lap, lap_stride, nt, max_n, max_m, c->stream));
  return MOT_OK;
}
int mot_cosine_cost_gated_host(mot_ctx* c, const float* a, int n, const float* b, int m, int d, const float* a_xyxy, const float* b_xyxy,
                               int cost_mode, float prox_thresh, float* out) {
  if (n <= 0 || m <= 0) return MOT_OK;
  std::vector<float> sa, sb;
  to_soa4(a_xyxy, n, 4, 4, sa);
  to_soa4(b_xyxy, m, 4, 4, sb);
  DBuf da, db, dout, dba, dbb, dt, dl;
  MOT_HIP(c, da.alloc(static_cast<size_t>(n) * d * 4)); MOT_HIP(c, db.alloc(static_cast<size_t>(m) * d * 4));
  MOT_HIP(c, dout.alloc(static_cast<size_t>(n) * m * 4)); MOT_HIP(c, dba.alloc(sa.size() * 4)); MOT_HIP(c, dbb.alloc(sb.size() * 4));
  MOT_HIP(c, dt.alloc(sizeof(mot_cos_task))); MOT_HIP(c, dl.alloc(sizeof(mot_lap_task)));
  MOT_HIP(c, hipMemcpyAsync(da.p, a, static_cast<size_t>(n) * d * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(db.p, b, static_cast<size_t>(m) * d * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dba.p, sa.data(), sa.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dbb.p, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dout.p, out, static_cast<size_t>(n) * m * 4, hipMemcpyHostToDevice, c->stream));  // (entries of pairs that fail the test keep the caller's values)
  mot_cos_task t{};
  t.n = n; t.m = m; t.d = d; t.a = da.as<float>(); t.lda = d; t.b = db.as<float>(); t.ldb = d; t.out = dout.as<float>(); t.ldo = m;
  mot_lap_task L{};
  L.n = n; L.m = m; L.geom.n = n; L.geom.m = m; L.geom.a = dba.as<float>(); L.geom.lda = n; L.geom.b = dbb.as<float>(); L.geom.ldb = m;
  L.geom.mode = cost_mode; L.geom.prox_thresh = prox_thresh;
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dl.p, &L, sizeof(L), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_embed_gated(dt.as<mot_cos_task>(), dl.as<mot_lap_task>(), 1, 1, n, m, c->stream));
  MOT_HIP(c, hipMemcpyAsync(out, dout.p, static_cast<size_t>(n) * m * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

int mot_ocsort_cost_host(mot_ctx* c, const float* dets5, int nd, const float* trks4, int nt, const float* vel2,
                         const float* prev5, float vdc, float* cost, float* iou) {
  if (nd <= 0 || nt <= 0) return MOT_OK;
  std::vector<float> sd, st, sv, sp;
  to_soa4(dets5, nd, 5, 5, sd);
  to_soa4(trks4, nt, 4, 4, st);
  to_soa4(vel2, nt, 2, 2, sv);
  to_soa4(prev5, nt, 5, 5, sp);
  DBuf dd, dtb, dv, dp, dc, di, dtask;
  MOT_HIP(c, dd.alloc(sd.size() * 4)); MOT_HIP(c, dtb.alloc(st.size() * 4)); MOT_HIP(c, dv.alloc(sv.size() * 4));
  MOT_HIP(c, dp.alloc(sp.size() * 4)); MOT_HIP(c, dc.alloc(static_cast<size_t>(nd) * nt * 4));
  MOT_HIP(c, di.alloc(static_cast<size_t>(nd) * nt * 4)); MOT_HIP(c, dtask.alloc(sizeof(mot_ocsort_task)));
  MOT_HIP(c, hipMemcpyAsync(dd.p, sd.data(), sd.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dtb.p, st.data(), st.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dv.p, sv.data(), sv.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dp.p, sp.data(), sp.size() * 4, hipMemcpyHostToDevice, c->stream));
  mot_ocsort_task t{};
  t.nd = nd; t.nt = nt; t.dbox = dd.as<float>(); t.ldd = nd; t.dconf = dd.as<float>() + static_cast<size_t>(4) * nd;
  t.tbox = dtb.as<float>(); t.ldt = nt; t.vel = dv.as<float>(); t.ldv = nt; t.prev = dp.as<float>(); t.ldp = nt;
  t.vdc_weight = vdc; t.cost = dc.as<float>(); t.iou = di.as<float>(); t.ldc = nt;
  MOT_HIP(c, hipMemcpyAsync(dtask.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_ocsort(dtask.as<mot_ocsort_task>(), 1, nd, nt, true, c->stream));
  MOT_HIP(c, hipMemcpyAsync(cost, dc.p, static_cast<size_t>(nd) * nt * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(iou, di.p, static_cast<size_t>(nd) * nt * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

int mot_lap_solve_host(mot_ctx* c, const float* cost, int n, int m, float thresh, int mode, const float* iou, float gate,
                       int* x, int* y, int* info) {
  return mot_lap_solve_prof_host(c, cost, n, m, thresh, mode, iou, gate, x, y, info, nullptr);
}
int mot_lap_solve_prof_host(mot_ctx* c, const float* cost, int n, int m, float thresh, int mode, const float* iou, float gate,
                            int* x, int* y, int* info, long long* prof8) {
  if (n <= 0 || m <= 0) {
    for (int i = 0; i < n; ++i) x[i] = -1;
    for (int j = 0; j < m; ++j) y[j] = -1;
    if (info) *info = 2;
    return MOT_OK;
  }
  DBuf dc, di, dx, dy, dinfo, dwork, dt, dprof, drl;
  MOT_HIP(c, dc.alloc(static_cast<size_t>(n) * m * 4)); MOT_HIP(c, dx.alloc(n * 4)); MOT_HIP(c, dy.alloc(m * 4));
  MOT_HIP(c, drl.alloc(mot_lap_rowlist_bytes(n)));
  MOT_HIP(c, dinfo.alloc(16)); MOT_HIP(c, dwork.alloc(mot_lap_work_bytes(n, m))); MOT_HIP(c, dt.alloc(sizeof(mot_lap_task)));
  MOT_HIP(c, hipMemcpyAsync(dc.p, cost, static_cast<size_t>(n) * m * 4, hipMemcpyHostToDevice, c->stream));
  if (iou) {
    MOT_HIP(c, di.alloc(static_cast<size_t>(n) * m * 4));
    MOT_HIP(c, hipMemcpyAsync(di.p, iou, static_cast<size_t>(n) * m * 4, hipMemcpyHostToDevice, c->stream));
  }
  mot_lap_task t{};
  t.n = n; t.m = m; t.cost = dc.as<float>(); t.ldc = m; t.thresh = thresh; t.x = dx.as<int>(); t.y = dy.as<int>();
  t.mode = mode; t.iou = iou ? di.as<float>() : nullptr; t.ldi = m; t.gate = gate; t.info = dinfo.as<int>();
  t.work = dwork.p;
  t.rowlist = drl.p;
  if (prof8) {  // per-phase shader cycles of the exact solver (the sparse solver leaves a task that asks for them alone)
    MOT_HIP(c, dprof.alloc(288));
    MOT_HIP(c, hipMemsetAsync(dprof.p, 0, 288, c->stream));
    t.prof = dprof.as<long long>();
  }
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_lap(dt.as<mot_lap_task>(), 1, n, m, false, false, false, c->stream));
  MOT_HIP(c, hipMemcpyAsync(x, dx.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(y, dy.p, m * 4, hipMemcpyDeviceToHost, c->stream));
  int inf = 0;
  MOT_HIP(c, hipMemcpyAsync(&inf, dinfo.p, 4, hipMemcpyDeviceToHost, c->stream));
  if (prof8) MOT_HIP(c, hipMemcpyAsync(prof8, dprof.p, 288, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (info) *info = inf;
  return MOT_OK;
}

int mot_lap_geom_host(mot_ctx* c, const float* a, int n, const float* b, int m, const float* bconf, int cost_mode, float thresh,
                      int lap_mode, float gate, int* x, int* y, float* xval, int* info, long long* prof8) {
  if (n <= 0 || m <= 0) {
    for (int i = 0; i < n; ++i) { x[i] = -1; if (xval) xval[i] = 0.f; }
    for (int j = 0; j < m; ++j) y[j] = -1;
    if (info) *info = 2;
    return MOT_OK;
  }
  std::vector<float> sa, sb;
  to_soa4(a, n, 4, 4, sa);
  to_soa4(b, m, 4, 4, sb);
  DBuf da, db, dc, dx, dy, dv, dinfo, dwork, dt;
  MOT_HIP(c, da.alloc(sa.size() * 4)); MOT_HIP(c, db.alloc(sb.size() * 4)); MOT_HIP(c, dc.alloc(m * 4));
  MOT_HIP(c, dx.alloc(n * 4)); MOT_HIP(c, dy.alloc(m * 4)); MOT_HIP(c, dv.alloc(n * 4)); MOT_HIP(c, dinfo.alloc(16));
  MOT_HIP(c, dwork.alloc(mot_lap_work_bytes(n, m))); MOT_HIP(c, dt.alloc(sizeof(mot_lap_task)));
  MOT_HIP(c, hipMemcpyAsync(da.p, sa.data(), sa.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(db.p, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (bconf) MOT_HIP(c, hipMemcpyAsync(dc.p, bconf, m * 4, hipMemcpyHostToDevice, c->stream));
  mot_lap_task t{};
  t.n = n; t.m = m; t.thresh = thresh; t.x = dx.as<int>(); t.y = dy.as<int>(); t.mode = lap_mode; t.gate = gate;
  t.xval = dv.as<float>(); t.info = dinfo.as<int>(); t.work = dwork.p;
  DBuf dprof;
  MOT_HIP(c, dprof.alloc(288));
  MOT_HIP(c, hipMemsetAsync(dprof.p, 0, 288, c->stream));
  t.prof = prof8 ? dprof.as<long long>() : nullptr;
  t.geom.n = n; t.geom.m = m; t.geom.a = da.as<float>(); t.geom.lda = n; t.geom.b = db.as<float>(); t.geom.ldb = m;
  t.geom.bconf = bconf ? dc.as<float>() : nullptr; t.geom.mode = cost_mode;
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_lap(dt.as<mot_lap_task>(), 1, n, m, true, false, cost_mode != MOT_COST_BOTSORT, c->stream));
  MOT_HIP(c, hipMemcpyAsync(x, dx.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(y, dy.p, m * 4, hipMemcpyDeviceToHost, c->stream));
  std::vector<float> hv(n);
  int inf = 0;
  MOT_HIP(c, hipMemcpyAsync(hv.data(), dv.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(&inf, dinfo.p, 4, hipMemcpyDeviceToHost, c->stream));
  if (prof8) MOT_HIP(c, hipMemcpyAsync(prof8, dprof.p, 288, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (xval) for (int i = 0; i < n; ++i) xval[i] = hv[i];
  if (info) *info = inf;
  return MOT_OK;
}

static int kf_host(mot_ctx* c, int kind, int op, int n, const float* meas4, const float* q3, const unsigned char* flags,
                   const float* warp9, float* mean, float* cov, float* boxes4, const float* conf = nullptr) {
  if (n <= 0) return MOT_OK;
  const int D = mot_kf_dim(kind);
  const int RS = D + D * D;  // one record per track: mean then covariance (the slab layout of mot_kf_task)
  std::vector<float> sm(static_cast<size_t>(RS) * n), sz;
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < D; ++k) sm[static_cast<size_t>(i) * RS + k] = mean[static_cast<size_t>(i) * D + k];
    for (int k = 0; k < D * D; ++k) sm[static_cast<size_t>(i) * RS + D + k] = cov[static_cast<size_t>(i) * D * D + k];
  }
  DBuf dm, dz, df, db, dt, dcf;
  MOT_HIP(c, dm.alloc(sm.size() * 4)); MOT_HIP(c, dz.alloc(static_cast<size_t>(4) * n * 4));
  if (conf) { MOT_HIP(c, dcf.alloc(static_cast<size_t>(n) * 4)); MOT_HIP(c, hipMemcpyAsync(dcf.p, conf, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, c->stream)); }
  MOT_HIP(c, df.alloc(n)); MOT_HIP(c, db.alloc(static_cast<size_t>(4) * n * 4)); MOT_HIP(c, dt.alloc(sizeof(mot_kf_task)));
  MOT_HIP(c, hipMemcpyAsync(dm.p, sm.data(), sm.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (meas4) {
    to_soa4(meas4, n, 4, 4, sz);
    MOT_HIP(c, hipMemcpyAsync(dz.p, sz.data(), sz.size() * 4, hipMemcpyHostToDevice, c->stream));
  }
  if (flags) MOT_HIP(c, hipMemcpyAsync(df.p, flags, n, hipMemcpyHostToDevice, c->stream));
  mot_kf_task t{};
  t.mean = dm.as<float>(); t.cov = dm.as<float>() + D; t.cap = n; t.n = n; t.flags = flags ? df.as<uint8_t>() : nullptr;
  t.meas = dz.as<float>(); t.ldm = n; t.boxes = boxes4 ? db.as<float>() : nullptr; t.ldb = n;
  t.q[0] = q3 ? q3[0] : 0.01f; t.q[1] = q3 ? q3[1] : 0.01f; t.q[2] = q3 ? q3[2] : 0.0001f;
  if (warp9) std::memcpy(t.warp, warp9, sizeof(t.warp));
  t.conf = conf ? dcf.as<float>() : nullptr;
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_kf_op(op, kind, dt.as<mot_kf_task>(), 1, n, c->stream));
  MOT_HIP(c, hipMemcpyAsync(sm.data(), dm.p, sm.size() * 4, hipMemcpyDeviceToHost, c->stream));
  std::vector<float> sb(static_cast<size_t>(4) * n);
  if (boxes4) MOT_HIP(c, hipMemcpyAsync(sb.data(), db.p, sb.size() * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < D; ++k) mean[static_cast<size_t>(i) * D + k] = sm[static_cast<size_t>(i) * RS + k];
    for (int k = 0; k < D * D; ++k) cov[static_cast<size_t>(i) * D * D + k] = sm[static_cast<size_t>(i) * RS + D + k];
    if (boxes4) for (int k = 0; k < 4; ++k) boxes4[static_cast<size_t>(i) * 4 + k] = sb[static_cast<size_t>(k) * n + i];
  }
  return MOT_OK;
}

// XYAH update of n tracks whose covariances are in block form (mot_kf_task.cov_blocks), through the two launches ByteTrack's device lifecycle
// uses: block kernel, then the dense kernel over the tracks it handed on. mean [n][8] in/out; blocks [n][16] in/out; upd_flags [n] MOT_KF_* bits;
// dense_flag [n] out (1: the track left the block form, its state is then in cov_dense [n][64], mean still in `mean`).
int mot_kf_update_blocks_host(mot_ctx* c, int n, const float* meas4, const unsigned char* upd_flags, float* mean, float* blocks,
                              unsigned char* dense_flag, float* cov_dense) {
  if (n <= 0) return MOT_OK;
  if (!meas4 || !mean || !blocks || !dense_flag || !cov_dense) return MOT_ERR_INVALID;
  std::vector<float> sz;
  to_soa4(meas4, n, 4, 4, sz);
  std::vector<int32_t> ident(n);
  for (int i = 0; i < n; ++i) ident[i] = i;
  DBuf dm, db, dr, dz, df, dfl, dt, dfi, dff;
  MOT_HIP(c, dm.alloc(static_cast<size_t>(n) * 8 * 4)); MOT_HIP(c, db.alloc(static_cast<size_t>(n) * 16 * 4)); MOT_HIP(c, dr.alloc(static_cast<size_t>(n) * 64 * 4));
  MOT_HIP(c, dz.alloc(sz.size() * 4)); MOT_HIP(c, df.alloc(n)); MOT_HIP(c, dfl.alloc(n)); MOT_HIP(c, dt.alloc(2 * sizeof(mot_kf_task)));
  MOT_HIP(c, dfi.alloc(static_cast<size_t>(n) * 3 * 4)); MOT_HIP(c, dff.alloc(n));
  MOT_HIP(c, hipMemcpyAsync(dm.p, mean, static_cast<size_t>(n) * 8 * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(db.p, blocks, static_cast<size_t>(n) * 16 * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemsetAsync(dr.p, 0, static_cast<size_t>(n) * 64 * 4, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dz.p, sz.data(), sz.size() * 4, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemsetAsync(df.p, 0, n, c->stream));
  if (upd_flags) MOT_HIP(c, hipMemcpyAsync(dfl.p, upd_flags, n, hipMemcpyHostToDevice, c->stream));
  mot_kf_task t[2] = {};
  t[0].mean = dr.as<float>(); t[0].cap = n; t[0].n = n; t[0].meas = dz.as<float>(); t[0].ldm = n; t[0].flags = upd_flags ? dfl.as<uint8_t>() : nullptr;
  t[0].mean_dense = dm.as<float>(); t[0].cov_blocks = db.as<float>(); t[0].dense_flag = df.as<unsigned char>();
  t[1] = t[0];
  t[1].n = 0; t[1].src = dfi.as<int32_t>(); t[1].dst = dfi.as<int32_t>() + n; t[1].midx = dfi.as<int32_t>() + 2 * n; t[1].flags = dff.as<uint8_t>();
  MOT_HIP(c, hipMemcpyAsync(dt.p, t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_kf_update_blocks(dt.as<mot_kf_task>(), dt.as<mot_kf_task>() + 1, 1, n, c->stream));
  MOT_HIP(c, hipMemcpyAsync(mean, dm.p, static_cast<size_t>(n) * 8 * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(blocks, db.p, static_cast<size_t>(n) * 16 * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(dense_flag, df.p, n, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(cov_dense, dr.p, static_cast<size_t>(n) * 64 * 4, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

int mot_kf_update_conf_host(mot_ctx* c, int kind, int n, const float* meas4, const float* conf, float* mean, float* cov) {
  if (conf && kind != MOT_KF_XYAH) return MOT_ERR_INVALID;
  return kf_host(c, kind, 2, n, meas4, nullptr, nullptr, nullptr, mean, cov, nullptr, conf);
}

int mot_feat_update_host_alpha(mot_ctx* c, int mode, float alpha, const float* alpha_i, int n, int d, float* feat, const float* src) {
  if (n <= 0 || d <= 0) return MOT_OK;
  if (mode < 0 || mode > 3 || !feat || !src) return MOT_ERR_INVALID;
  const size_t bytes = static_cast<size_t>(n) * d * 4;
  DBuf df, ds, dt, da;
  MOT_HIP(c, df.alloc(bytes)); MOT_HIP(c, ds.alloc(bytes)); MOT_HIP(c, dt.alloc(sizeof(mot_feat_task)));
  MOT_HIP(c, hipMemcpyAsync(df.p, feat, bytes, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(ds.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  if (alpha_i) {
    MOT_HIP(c, da.alloc(static_cast<size_t>(n) * 4));
    MOT_HIP(c, hipMemcpyAsync(da.p, alpha_i, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, c->stream));
  }
  mot_feat_task t{};
  t.n = n; t.d = d; t.feat = df.as<float>(); t.ldf = d; t.src = ds.as<float>(); t.lds = d; t.mode = mode; t.alpha = alpha;
  t.alpha_i = alpha_i ? da.as<float>() : nullptr;
  MOT_HIP(c, hipMemcpyAsync(dt.p, &t, sizeof(t), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, mot::launch_feat(dt.as<mot_feat_task>(), 1, n, c->stream));
  MOT_HIP(c, hipMemcpyAsync(feat, df.p, bytes, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}
int mot_feat_update_host(mot_ctx* c, int mode, float alpha, int n, int d, float* feat, const float* src) {
  return mot_feat_update_host_alpha(c, mode, alpha, nullptr, n, d, feat, src);
}

int mot_kf_apply_host(mot_ctx* c, int kind, int op, int n, const float* meas4, const float* q3, const unsigned char* flags,
                      float* mean, float* cov, float* boxes4) {
  if (op < 0 || op > 2) return MOT_ERR_INVALID;
  return kf_host(c, kind, op, n, meas4, q3, flags, nullptr, mean, cov, boxes4);
}
int mot_kf_warp_host(mot_ctx* c, int kind, int n, const float* warp9, int predict_first, const float* q3, float* mean, float* cov,
                     float* boxes4) {
  if (!warp9 || kind == MOT_KF_XYAH) { c->err = "mot_kf_warp_host: needs a warp and an XYSR or XYWH filter"; return MOT_ERR_INVALID; }
  return kf_host(c, kind, predict_first ? 5 : 4, n, nullptr, q3, nullptr, warp9, mean, cov, boxes4);
}

}  // extern "C"
