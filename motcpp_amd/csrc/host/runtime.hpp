// Host runtime under the tracker classes: owns one mot_ctx per GPU, mirrored bump arenas for
// per-frame uploads/downloads, and the per-stage task lists that turn "every tracker of a batch
// wants an IoU matrix now" into ONE kernel launch. Plain C++17 — the only thing it knows about the
// GPU is the C ABI in motcpp_amd.h (no HIP headers, no device code here).
//
// A frame is processed as a few stages. In each stage every tracker of the batch appends tasks
// (detections to prepare, track slots to predict/update, cost matrices, assignments) and
// Device::flush() uploads the stage's inputs in one copy, launches each non-empty kernel family
// once over all tasks (grid dimension = task), downloads the results in one copy and syncs.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "motcpp_amd.h"

namespace motcpp::rt {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

template <class T>
struct Span {
  T* h = nullptr;  // host mirror (pinned) or nullptr for device-only memory
  T* d = nullptr;  // device address
  size_t n = 0;
};

// Chunked bump arena. Device addresses stay valid for the whole frame (chunks are never moved). alloc() is lock-free
// on its fast path (one atomic add), so the stage machines of a batch can run on many host threads.
class Arena {
 public:
  Arena(mot_ctx* ctx, size_t chunk_bytes, bool host_mirror);
  ~Arena();
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  template <class T>
  Span<T> alloc(size_t n) {
    void *h, *d;
    raw_alloc(n * sizeof(T), &h, &d);
    return Span<T>{static_cast<T*>(h), static_cast<T*>(d), n};
  }
  void reset();       // frame boundary (single-threaded)
  void upload();      // host -> device for everything allocated since the last transfer (single-threaded)
  void download();    // device -> host for everything allocated since the last transfer (single-threaded)
  void clear_pending(); // memset(0) on the device for everything allocated since the last transfer (no mark change)
  size_t bytes_in_flight() const;
 private:
  struct Chunk {
    char* h = nullptr; char* d = nullptr; size_t cap = 0;
    std::atomic<size_t> top{0};
    size_t mark = 0;
  };
  struct alignas(64) Lease {  // per host thread: private slice of a chunk
    char* h = nullptr; char* d = nullptr;
    size_t off = 0, end = 0;
    uint64_t epoch = 0;
  };
  static constexpr size_t kLeaseBytes = size_t(128) << 10;
  void raw_alloc(size_t bytes, void** h, void** d);
  void shared_alloc(size_t bytes, void** h, void** d);
  std::vector<Lease> leases_;
  std::atomic<uint64_t> epoch_{1};  // bumped at every reset/transfer (single-threaded points)
  mot_ctx* ctx_;
  size_t chunk_bytes_;
  bool host_;
  std::vector<std::unique_ptr<Chunk>> chunks_;  // grows under grow_mu_ only; readers index below n_chunks_
  std::atomic<size_t> cur_{0};
  std::atomic<size_t> n_chunks_{0};
  std::mutex grow_mu_;
};

struct StageCounters {
  long flushes = 0, launches = 0;
  double ms_begin = 0, ms_flush = 0, ms_advance = 0, ms_sync_wait = 0;  // host wall time by phase (run_frame)
};

// Kernel families timed with HIP events when Device::profile is on (bench.py's roofline leg).
enum Family { F_DET = 0, F_FEAT, F_KF_INIT, F_KF_UPDATE, F_KF_PREDICT, F_KF_BOXES, F_COSINE, F_IOU, F_OCSORT, F_LAP, F_COUNT };
struct FamilyStat {
  double ms = 0.0;      // summed launch durations (event pairs on the stream)
  long launches = 0;
  long tasks = 0;
  double bytes = 0.0;   // algorithmic bytes (DESIGN.md "kernels")
  double flops = 0.0;   // cosine only
};

class Device {
 public:
  explicit Device(int device_index);
  ~Device();
  static std::shared_ptr<Device> shared(int device_index);  // process-wide default per GPU

  mot_ctx* ctx = nullptr;
  int index = 0;
  // One frame (run_frame) or one utils:: call at a time per Device: its arenas, task lists and event pool are per-frame state.
  // Trackers that share the process-wide Device of a GPU may be updated from different host threads (one tracker per camera
  // thread, as the reference allows); their frames then serialise here. A batch created with a private Device has its own.
  std::mutex frame_mu;
  std::unique_ptr<Arena> up, down, tmp;
  std::unique_ptr<Arena> zdown;  // like `down`, but the device side is zeroed before the stage's kernels run (atomic counters)

  // Task lists of the current stage (host copies; flush() concatenates them and moves them to the device). One set
  // per host thread: a stage machine appends to the set of the thread it runs on, without locking.
  struct TaskLists {
    std::vector<mot_det_task> det[4];  // by mot_det_kind
    std::vector<mot_kf_task> kf_init[3], kf_upd[3], kf_pred[3], kf_box[3], kf_warp[3], kf_predw[3];
    std::vector<mot_feat_task> feat_set, feat_ema;
    std::vector<mot_feat_task> feat_late;  // after feat_ema: rows that read what the two lists before wrote (StrongSORT's sample library)
    std::vector<mot_ss_nn_task> ss_nn;     // StrongSORT: minimum over a track's samples (after the raw inner products, before the gate)
    std::vector<mot_gate_task> gate;       // XYAH motion gate + blend (+ clamp) on a cost matrix
    std::vector<mot_ss_iou_task> ss_iou;   // StrongSORT's IoU cost on tlwh boxes
    std::vector<mot_hyb_task> hyb[5];      // HybridSORT's nine-state filter and pairwise costs, by op (MOT_HYB_*)
    std::vector<mot_boost_task> boost[6];  // BoostTrack's constant-noise filter and costs, by op (MOT_BOOST_*)
    std::vector<mot_ucmc_task> ucmc[5];    // UCMCTrack's double-precision ground-plane filter, by op (MOT_UCMC_*): map with the detections, births /
                                           // updates with the Kalman updates, predict with the predicts, the cost matrices in front of the assignments
    std::vector<mot_cos_task> cos;
    std::vector<mot_cos_task> dot;    // raw inner products (DeepOC-SORT's embedding similarity)
    std::vector<mot_deep_task> deep;  // ... and its weighting into the association cost (after the OC-SORT cost, before the LAP)
    std::vector<mot_iou_task> iou;
    std::vector<mot_ocsort_task> oc;
    std::vector<mot_lap_task> lap;
    bool lap_geom = false;  // some queued LAP task carries on-the-fly geometry
    bool lap_assoc = false; // ... with an association measure other than IoU
    bool lap_appearance = false;  // ... with BoT-SORT's gated appearance cost (MOT_COST_BOTSORT)
    bool empty() const;
    void clear();
    void append(TaskLists& o);  // moves o's tasks behind this one's
  };
  static constexpr int kMaxHostThreads = 256;
  std::vector<TaskLists> lists;  // [kMaxHostThreads]
  TaskLists& q();                // the calling host thread's lists

  void begin_frame();
  bool pending() const;
  void flush();
  void check(int rc, const char* what);
  StageCounters counters;
  bool profile = false;
  FamilyStat stats[F_COUNT];
  void reset_stats();
 private:
  struct Timed { int family; void* e0; void* e1; };
  std::vector<void*> event_pool_;
  std::vector<Timed> timed_;
  void* get_event();
  void time_begin(int family);
  void time_end();
};

// Per-tracker device state + task-building helpers.
class Core {
 public:
  Core(std::shared_ptr<Device> dev, int kf_kind);
  ~Core();
  Device& dev() { return *dev_; }
  int kf_kind() const { return kind_; }

  // ---- Kalman slab (persistent slots + per-frame scratch slots) ----
  int new_slot();
  void release_slot(int s);
  void reserve(int extra_persistent, int scratch);  // call at frame start, before any task is queued
  int scratch_slot(int i) const { return pcap_ + i; }
  void clear_slots();
  float q[3] = {0.01f, 0.01f, 0.0001f};  // XYSR process noise tail
  int box_style = 0;                      // mot_kf_task.reserved of the boxes() tasks (StrongSORT: MOT_KF_BOX_TLWH_SUM)

  // ---- detections of the current frame ----
  struct Dets {
    int n = 0;
    const float* d_raw = nullptr;  // SoA [6][ld_raw]
    int ld_raw = 0;
    float* d_box = nullptr;        // [4][n]
    float* d_meas = nullptr;       // [4][n]
    const float* d_conf() const { return d_raw + static_cast<size_t>(4) * ld_raw; }
  };
  // resident != nullptr: the detections already sit in HBM as SoA [6][resident_ld]; nothing is uploaded
  Dets upload_dets(const float* colmajor, int n, int ld, int det_kind, const float* resident = nullptr, int resident_ld = 0);

  Span<int32_t> ints(const std::vector<int>& v);
  Span<uint8_t> bytes(const std::vector<uint8_t>& v);
  Span<float> floats(const std::vector<float>& v);

  // predict src slots into dst slots (nullptr: in place); returns device boxes [4][n] (ld = n)
  // warp9: optional 3x3 row-major camera-motion warp applied to every predicted state (mot_kf_predict_warp)
  float* predict(const std::vector<int>& src, const std::vector<int>* dst, const std::vector<uint8_t>* flags, Span<float>* boxes_dl,
                 const float* warp9 = nullptr);
  void warp(const std::vector<int>& slots, const float* warp9);  // camera-motion warp of stored states, no predict
  float* boxes(const std::vector<int>& slots, Span<float>* boxes_dl);  // state -> xyxy, [4][n]
  void update(const std::vector<int>& src, const std::vector<int>& dst, const std::vector<int>& midx, const Dets& dets);
  void initiate(const std::vector<int>& dst, const std::vector<int>& midx, const Dets& dets);

  struct Lap {
    int n = 0, m = 0;
    Span<int32_t> x, y, info;
    Span<float> xval;
    bool queued = false;
  };
  // rows: boxes [4][lda] (+ optional gather), cols: boxes [4][ldb] (+ optional gather) -> cost -> assignment
  struct IouArgs {
    const float* a = nullptr; int lda = 0; const int32_t* aidx = nullptr; int n = 0;
    const float* b = nullptr; int ldb = 0; const int32_t* bidx = nullptr; int m = 0;
    const float* bconf = nullptr;
    int mode = MOT_COST_IOU_DIST;
    const float* emb = nullptr; int lde = 0; float prox = 0.f, app = 0.f; int fuse = 0;
    int assoc = MOT_ASSOC_IOU; float frame_diag = 1.f;  // similarity measure (AssociationFunction mode)
  };
  float* iou_cost(const IouArgs& a, int* ldc);  // returns device cost matrix (tmp arena)
  Lap lap(const float* cost, int ldc, int n, int m, float thresh, int mode = MOT_LAP_PLAIN, const float* iou = nullptr,
          int ldi = 0, float gate = 0.f, bool want_xval = false);

  // assignment straight from boxes: the solver recomputes the IoU-family cost on the fly (no N x M matrix in memory)
  Lap lap_geom(const IouArgs& a, float thresh, int mode = MOT_LAP_PLAIN, float gate = 0.f, bool want_xval = false);

  float* d_mean() const { return mean_; }
  float* d_cov() const { return cov_; }
  int cap() const { return cap_; }

 private:
  void grow(int pcap, int scap);
  std::shared_ptr<Device> dev_;
  int kind_, D_;
  float* mean_ = nullptr;
  float* cov_ = nullptr;
  int cap_ = 0, pcap_ = 0, scap_ = 0;
  std::vector<int> free_;
  int next_ = 0;
};

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace motcpp::rt
