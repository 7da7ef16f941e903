/* motcpp_amd.h — C ABI of the MI355X-native association hot path of motcpp.
 *
 * Drop-in boundary: these entry points are what a binding of the reference's
 * internal primitive seam would call instead of the Eigen/CPU code. Each one cites the reference
 * interface it replaces (paths relative to the motcpp repository):
 *
 *   mot_det_prepare     STrack/BotSTrack/SortTrack ctors' box chains  src/trackers/bytetrack.cpp:18-37,
 *                       botsort.cpp:23-36, sort.cpp:21-41; include/motcpp/utils/ops.hpp:15-211
 *   mot_kf_initiate     BaseKalmanFilter::initiate src/motion/kalman_filter.cpp:29-42,
 *                       KalmanFilterXYWH::initiate include/motcpp/motion/kalman_filters/xywh_kf.hpp:41-62,
 *                       KalmanFilterXYSR ctor src/motion/kalman_filters/xysr_kf.cpp:10-69
 *   mot_kf_predict      ::predict kalman_filter.cpp:44-58, xysr_kf.cpp:71-77, xywh_kf.hpp:70-94 (+ the
 *                       per-track glue STrack::multi_predict bytetrack.cpp:97-115, KalmanBoxTracker::predict
 *                       ocsort.cpp:132-148)
 *   mot_kf_update       ::update kalman_filter.cpp:77-112, xysr_kf.cpp:79-112, xywh_kf.hpp:103-135
 *   mot_kf_boxes        STrack::xyxy bytetrack.cpp:117-127, BotSTrack::xyxy botsort.cpp:171-181,
 *                       utils::xysr2xyxy ops.hpp:202-211
 *   mot_iou_cost        utils::iou_batch include/motcpp/utils/iou.hpp:63-100, utils::iou_distance
 *                       src/utils/matching.cpp:62-65, utils::fuse_score matching.cpp:130-143, BoT-SORT's
 *                       gate/min botsort.cpp:433-466, ByteTrack::remove_duplicate_stracks bytetrack.cpp:659-706
 *   mot_cosine_cost     utils::embedding_distance src/utils/matching.cpp:67-92
 *   mot_feat_update     BotSTrack ctor / update_features botsort.cpp:38-46,158-169
 *   mot_ocsort_cost     ocsort_assoc::associate cost construction src/trackers/ocsort.cpp:624-679,699
 *   mot_lap_solve       utils::linear_assignment src/utils/matching.cpp:14-60 →
 *                       LAPSolver::linearAssignment include/motcpp/association/lap_solver.hpp:251-332
 *                       (+ OC-SORT's trivial-case shortcut and max-IoU gates ocsort.cpp:681-714,443,499)
 *
 * Conventions: every pointer inside a *_task is a DEVICE pointer; task arrays are device arrays;
 * all calls are asynchronous on the context's HIP stream unless suffixed _host (those take host
 * pointers, copy, run and synchronise — convenience for bindings and tests). Functions return
 * MOT_OK (0) or a negative mot_status and never throw. One mot_ctx per host thread / stream.
 * There is no CPU fallback: without a usable gfx950 device mot_ctx_create fails.
 */
#ifndef MOTCPP_AMD_H_
#define MOTCPP_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mot_ctx mot_ctx;

typedef enum mot_status {
  MOT_OK = 0,
  MOT_ERR_INVALID = -1,   /* bad argument */
  MOT_ERR_HIP = -2,       /* a HIP runtime call failed; see mot_ctx_last_error */
  MOT_ERR_NOMEM = -3,
  MOT_ERR_NODEVICE = -4,  /* no gfx950 device visible */
  MOT_ERR_CAPACITY = -5   /* a fixed device-side capacity was exceeded (mot_bt_*) */
} mot_status;

/* ---- context, stream, memory --------------------------------------------------------- */
const char* mot_version(void);
int mot_device_count(void);
int mot_ctx_create(int device, void* hip_stream /* hipStream_t or NULL: own stream */, mot_ctx** out);
int mot_ctx_destroy(mot_ctx* ctx);
int mot_ctx_sync(mot_ctx* ctx);
/* makes the context's GPU the calling host thread's current device (HIP keeps that per thread): a thread other than the one that
 * created the context calls this before it allocates or launches through it */
int mot_ctx_bind(mot_ctx* ctx);
void* mot_ctx_stream(mot_ctx* ctx);
const char* mot_ctx_last_error(mot_ctx* ctx);

int mot_malloc(mot_ctx* ctx, size_t bytes, void** dptr);
int mot_free(mot_ctx* ctx, void* dptr);
int mot_host_alloc(mot_ctx* ctx, size_t bytes, void** hptr); /* pinned */
int mot_host_free(mot_ctx* ctx, void* hptr);
int mot_memcpy_h2d(mot_ctx* ctx, void* d, const void* h, size_t bytes); /* async on the stream */
int mot_memcpy_d2h(mot_ctx* ctx, void* h, const void* d, size_t bytes); /* async on the stream */
int mot_memcpy_d2d(mot_ctx* ctx, void* dst, const void* src, size_t bytes);
int mot_memset(mot_ctx* ctx, void* d, int value, size_t bytes);
/* stream-ordered timing (hipEvent pair): returns elapsed milliseconds between two marks */
int mot_timer_start(mot_ctx* ctx);
int mot_timer_stop(mot_ctx* ctx, float* ms); /* synchronises */
/* HIP events on the context's stream (used to time individual kernel launches for the roofline report) */
int mot_event_create(mot_ctx* ctx, void** ev);
int mot_event_destroy(mot_ctx* ctx, void* ev);
int mot_event_record(mot_ctx* ctx, void* ev);
int mot_event_elapsed(mot_ctx* ctx, void* ev_start, void* ev_stop, float* ms); /* both events must have completed */

/* ---- detections ---------------------------------------------------------------------- */
typedef enum mot_det_kind {
  MOT_DET_XYSR = 0, /* SORT / OC-SORT: box = raw xyxy, meas = xyxy2xysr                     */
  MOT_DET_XYAH = 1, /* ByteTrack: xywh -> tlwh -> xyah chain, box = xywh2xyxy(xywh)          */
  MOT_DET_XYWH = 2, /* BoT-SORT: xywh = (x1+w/2, y1+h/2, w, h), box = (cx-w/2, ...)          */
  MOT_DET_TLWH = 3  /* StrongSORT (strongsort.cpp:33-40, 948-956): box = tlwh (x1, y1, x2-x1, y2-y1), meas = Detection::to_xyah
                       (x + w/2, y + h/2, w/h, h)                                                  */
} mot_det_kind;

typedef struct mot_det_task {
  const float* dets; /* SoA planes [6][ld]: x1,y1,x2,y2,conf,cls (= a column-major N x 6 matrix) */
  int32_t ld, n;
  float* box;  int32_t ldb; /* out [4][ldb] association boxes                                  */
  float* meas; int32_t ldm; /* out [4][ldm] Kalman measurements                                */
} mot_det_task;
int mot_det_prepare(mot_ctx* ctx, int det_kind, const mot_det_task* tasks, int ntasks, int max_n);

/* ---- Kalman filters ------------------------------------------------------------------ */
typedef enum mot_kf_kind { MOT_KF_XYSR = 0 /* d=7 */, MOT_KF_XYAH = 1 /* d=8 */, MOT_KF_XYWH = 2 /* d=8 */ } mot_kf_kind;
enum {
  MOT_KF_ZERO_V7 = 1,     /* predict: mean[7] = 0 first (ByteTrack, non-Tracked: bytetrack.cpp:108-110) */
  MOT_KF_OCSORT_CLAMP = 2,/* predict: if x6 + x2 <= 0 then x6 = 0 (ocsort.cpp:134-136)                  */
  MOT_KF_NO_STORE = 4,    /* predict: only the boxes are wanted, the predicted state is not written      */
  MOT_KF_BOX_TLWH_SUM = 1, /* (value of mot_kf_task.reserved, not a flag bit) */
  MOT_KF_PREDICT_FIRST = 8/* update: predict the loaded state first (honouring MOT_KF_ZERO_V7), then update: with NO_STORE
                             predictions this replaces "predict into a scratch slot, update from it" without the round trip */
};
/* Track-state slab: an array of `cap` records, the record of slot s at mean + s*(d + d*d) = mean[d] followed by the d x d
 * covariance, row-major (288 bytes for the 8-state filters, 224 for XYSR; `mean` 16-byte aligned). The items of a launch are a
 * gather over slots, and a record is one contiguous run whatever the slot is. `cov` is informational (= mean + d). */
typedef struct mot_kf_task {
  float* mean; float* cov; int32_t cap;
  int32_t n;                 /* items                                                         */
  const int32_t* src;        /* [n] slot read  (NULL: item index)                             */
  const int32_t* dst;        /* [n] slot written (NULL: same as src)                          */
  const uint8_t* flags;      /* [n] MOT_KF_* bits or NULL                                     */
  const float* meas; int32_t ldm; /* [4][ldm] measurements (initiate / update)               */
  const int32_t* midx;       /* [n] measurement column per item (NULL: item index)            */
  float* boxes; int32_t ldb; /* optional out [4][ldb]: xyxy of the written state, column = item */
  float q[3];                /* XYSR only: Q(4,4), Q(5,5), Q(6,6)                             */
  int32_t reserved;          /* box style of `boxes`: 0 the tracker family's own (centre +- half size), MOT_KF_BOX_TLWH_SUM (XYAH, mot_kf_boxes /
                                mot_kf_predict): x2 = x1 + w, y2 = y1 + h — StrongSORT's Track::to_tlbr, src/trackers/strongsort.cpp:94-111 */
  float warp[9];             /* camera-motion warp, 3x3 row-major (mot_kf_warp, mot_kf_predict_warp) */
  const float* conf;         /* optional, MOT_KF_XYAH update: detection confidences, indexed like `meas` — the NSA Kalman rule of
                                BaseKalmanFilter::project, R = ((1 - conf) * std)^2 (src/motion/kalman_filter.cpp:60-75; StrongSORT's
                                Track::update passes the detection's confidence, strongsort.cpp:153). NULL: confidence 0 */
  float* mean_dense;         /* optional (mot_kf_update, 8-state filters): the means live in a dense array of their own, [cap][8] floats, and `mean` then
                                points at covariance-only records of 64 floats per slot (256 bytes on a 256-byte boundary). The device lifecycles
                                read boxes (the mean, of every track, several times per frame) from the dense array: one 64-byte line serves two
                                tracks, where a record's mean is 32 bytes of a 288-byte stride. NULL: records of 8 + 64 floats, the mean first */
  float* cov_blocks;         /* ByteTrack's device lifecycle (internal: mot::launch_kf_update_blocks): the covariance of a track whose state was
                                initiated and only ever predicted / updated by these filters is four 2 x 2 blocks — component c couples with its
                                velocity c + 4 and with nothing else; every other entry is an exact zero, the innovation covariance is diagonal —
                                stored as [cap][4][4] floats {P(c,c), P(c,c+4), P(c+4,c), P(c+4,c+4)}: 64 bytes instead of 256 */
  unsigned char* dense_flag; /* [cap] 1: the track left the block form (a non-positive or non-finite innovation variance, a non-finite input: the
                                dense arithmetic then spreads NaN / inf over the structural zeros) and lives in its 64-float record from then on */
  const float* meas4;        /* optional (kf_update_blocks_kernel): the same measurements as [n][4] floats — the four components of a detection in one 16-byte
                                access instead of four lines of the [4][ldm] planes */
} mot_kf_task;
int mot_kf_dim(int kf_kind);
int mot_kf_initiate(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);
int mot_kf_predict(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);
int mot_kf_update(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);
/* boxes of the PREDICTED states, nothing stored (what mot_kf_predict does for items flagged MOT_KF_NO_STORE, for a launch
 * in which every item is): the box depends on the predicted mean only, so 32 bytes of a record are read instead of all of it.
 * flags: MOT_KF_ZERO_V7 / MOT_KF_OCSORT_CLAMP as in mot_kf_predict. ByteTrack's pool copies, bytetrack.cpp:251-265. */
int mot_kf_predict_boxes(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);
int mot_kf_boxes(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);
/* Camera-motion compensation of the stored states with a caller-supplied warp (the image registration that produces it
 * is outside this library). MOT_KF_XYWH: BotSTrack::multi_gmc (src/trackers/botsort.cpp:60-91) — both corners of the
 * state's box go through the 3x3 warp (projective divide included), mean[0..3] = the new cx,cy,w,h, covariance untouched.
 * MOT_KF_XYSR: KalmanFilterXYSR::apply_affine_correction (src/motion/kalman_filters/xysr_kf.cpp:114-141) with
 * m = warp[0:2,0:2], t = warp[0:2,2] — centre, velocity and the position/velocity covariance blocks. MOT_KF_XYAH has no
 * such step in the reference: MOT_ERR_INVALID. `boxes`, if set, receives the xyxy of the warped states.
 * mot_kf_predict_warp = mot_kf_predict followed by mot_kf_warp on the same items, in one launch (BoT-SORT's pool,
 * botsort.cpp:314-322). */
int mot_kf_warp(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);
int mot_kf_predict_warp(mot_ctx* ctx, int kf_kind, const mot_kf_task* tasks, int ntasks, int max_n);

/* ---- gating distances (StrongSORT's motion gate) ---------------------------------------- */
/* Squared distance between every track's projected state and every measurement of the frame:
 *   MOT_KF_XYAH  BaseKalmanFilter::gating_distance (src/motion/kalman_filter.cpp:148-176): S = H P H^T + R (project(), :60-75,
 *                confidence 0), metric 0 "maha": z = LLT(S[:dim,:dim]).solve(d), |z|^2 — plain |d|^2 when the factorisation
 *                fails; metric 1 "gaussian": |d|^2
 *   MOT_KF_XYWH  KalmanFilterXYWH::gating_distance (include/motcpp/motion/kalman_filters/xywh_kf.hpp:140-176):
 *                d^T S^-1 d with the partial-pivot LU inverse (only_position: its leading 2 x 2 block)
 * and, fused into the same pass, what the two callers do with it:
 *   MOT_GATE_FUSE_MOTION  utils::fuse_motion (include/motcpp/utils/matching.hpp:60-94): +inf above chi2inv95[dim-1],
 *                         else lambda * cost + (1 - lambda) * distance
 *   MOT_GATE_STRONGSORT   gate_cost_matrix (src/trackers/strongsort.cpp:449-492): cost replaced by gated_cost above 9.4877,
 *                         then EVERY entry blended lambda * cost + (1 - lambda) * distance
 * dim = 2 (only_position) or 4. meas: SoA [4][ldm] (xyah for XYAH, xywh for XYWH), what mot_det_prepare writes. */
typedef enum mot_gate_mode { MOT_GATE_DISTANCE = 0, MOT_GATE_FUSE_MOTION = 1, MOT_GATE_STRONGSORT = 2, MOT_GATE_CLAMP = 0x100 /* flag, OR'ed in */ } mot_gate_mode;
typedef struct mot_gate_task {
  int32_t n, m;                        /* tracks x measurements                                              */
  const float* mean; const int32_t* src; /* Kalman records (see mot_kf_task) and the slot of each row, NULL = identity */
  const float* meas; int32_t ldm;
  const float* cost; int32_t ldc;      /* n x m row-major input (modes 1, 2)                                 */
  float* out; int32_t ldo;             /* n x m row-major                                                    */
  int32_t mode, only_position, metric; /* metric 0 = "maha", 1 = "gaussian" (XYAH only)                      */
  float lambda, gated_cost;
  float clamp_above; /* with MOT_GATE_CLAMP in `mode`: a result above this becomes clamp_above + 1e-5 (min_cost_matching, strongsort.cpp:376-379) */
} mot_gate_task;
int mot_gate_cost(mot_ctx* ctx, int kf_kind, const mot_gate_task* tasks, int ntasks, int max_n, int max_m);
/* host-pointer form: mean8 [n][8], cov [n][64], meas row-major [m][4], cost/out row-major n x m (cost NULL for mode 0) */
int mot_gate_cost_host(mot_ctx* ctx, int kf_kind, int mode, int n, int m, const float* mean8, const float* cov64, const float* meas4,
                       const float* cost_or_null, int only_position, int metric, float lambda, float gated_cost, float* out);

/* ---- N x M box costs ----------------------------------------------------------------- */
typedef enum mot_cost_mode {
  MOT_COST_IOU = 0,           /* iou                                                          */
  MOT_COST_IOU_DIST = 1,      /* 1 - iou                                                      */
  MOT_COST_IOU_DIST_FUSE = 2, /* 1 - (1 - (1 - iou)) * conf_j                                 */
  MOT_COST_NEG_IOU = 3,       /* -iou (OC-SORT rematch)                                       */
  MOT_COST_BOTSORT = 4,       /* min(fuse?(1-iou), gate(emb/2)) — botsort.cpp:433-466          */
  MOT_COST_FUSE_IOU = 5       /* fuse_iou (matching.cpp:109-128): 1 - (1 - emb_ij) * (1 + (1 - (1 - iou))) / 2, emb = ReID cost n x m */
} mot_cost_mode;
/* the pairwise similarity the cost is built from (AssociationFunction modes, include/motcpp/utils/iou.hpp:371-414);
 * "iou" in the cost-mode formulas above means this value */
typedef enum mot_assoc {
  MOT_ASSOC_IOU = 0, MOT_ASSOC_HMIOU = 1, MOT_ASSOC_GIOU = 2, MOT_ASSOC_CIOU = 3, MOT_ASSOC_DIOU = 4, MOT_ASSOC_CENTROID = 5
} mot_assoc;
typedef struct mot_iou_task {
  int32_t n, m;
  const float* a; int32_t lda; const int32_t* aidx; /* row boxes [4][lda], optional gather       */
  const float* b; int32_t ldb; const int32_t* bidx; /* column boxes                              */
  const float* bconf;                               /* column confidences, indexed like b        */
  float* cost; int32_t ldc;                         /* out n x m row-major (may be NULL)         */
  int32_t mode;
  const float* emb; int32_t lde;                    /* BOTSORT: cosine distances n x m (emb NULL and lde < 0: constant 1) */
  float prox_thresh, app_thresh; int32_t fuse;      /* BOTSORT                                   */
  int32_t* pairs; int32_t* npairs; int32_t pairs_cap; float pair_thresh; /* optional: (i,j) with value < thresh */
  /* optional duplicate marking instead of a pair list (ByteTrack remove_duplicate_stracks, bytetrack.cpp:659-706): for
   * every pair with value < pair_thresh, dup_b[j] = 1 if age_a[i] > age_b[j], else dup_a[i] = 1 (flags zeroed by the caller) */
  const int32_t* age_a; const int32_t* age_b; uint8_t* dup_a; uint8_t* dup_b;
  int32_t assoc;     /* a mot_assoc value, 0 = IoU                                                          */
  float frame_diag;  /* CENTROID: sqrt(w*w + h*h) of the frame                                       */
} mot_iou_task;
int mot_iou_cost(mot_ctx* ctx, const mot_iou_task* tasks, int ntasks, int max_n, int max_m);
/* flags: the caller promises properties of the whole task array so that a leaner kernel instance can be launched */
enum { MOT_COST_F_IOU_ONLY = 1 /* every task has assoc == MOT_ASSOC_IOU */ };
int mot_iou_cost_ex(mot_ctx* ctx, const mot_iou_task* tasks, int ntasks, int max_n, int max_m, int flags);

typedef struct mot_ocsort_task {
  int32_t nd, nt;
  const float* dbox; int32_t ldd; const int32_t* didx; const float* dconf; /* detections (gathered)   */
  const float* tbox; int32_t ldt;  /* predicted track boxes [4][ldt]                                   */
  const float* vel;  int32_t ldv;  /* [2][ldv] velocity direction (dy, dx)                             */
  const float* prev; int32_t ldp;  /* [5][ldp] k-previous observation x1,y1,x2,y2,score (-1: none)     */
  float vdc_weight;
  float* cost; float* iou; int32_t ldc; /* out nd x nt row-major: -(iou + angle) and iou               */
  int32_t assoc; float frame_diag;      /* similarity used as "iou" (mot_assoc), frame diagonal for CENTROID */
} mot_ocsort_task;
int mot_ocsort_cost(mot_ctx* ctx, const mot_ocsort_task* tasks, int ntasks, int max_nd, int max_nt);
int mot_ocsort_cost_ex(mot_ctx* ctx, const mot_ocsort_task* tasks, int ntasks, int max_nd, int max_nt, int flags);

/* ---- DeepOC-SORT: the embedding term of the association cost (src/trackers/deepocsort.cpp:294-346, 419-441) ---------- */
/* emb: nd x nt similarities dets_embs . trk_embs^T (mot_embedding_cost, MOT_EMB_DOT); iou, cost: what mot_ocsort_cost wrote
 * (cost = -(iou + angle)). Entries with iou <= 0 count as 0; with aw_off == 0 row i is weighted by
 * rw_i = 1 - max(second_i / max_i - aw_param, 0) / (1 - aw_param) from the two largest entries of the row (0 if the largest
 * is 0, 1 if the row has fewer than two entries), columns likewise, final = ((w * rw_i) * cw_j) * emb; with aw_off it is
 * emb * w. cost becomes -((iou + angle) + final). rw / cw: scratch [nd] / [nt]. */
typedef struct mot_deep_task {
  int32_t nd, nt;
  const float* emb; int32_t lde;
  const float* iou; int32_t ldi;
  float* cost; int32_t ldc;
  float* rw; float* cw;
  float w, aw_param; int32_t aw_off;
} mot_deep_task;
int mot_deepoc_cost(mot_ctx* ctx, const mot_deep_task* tasks, int ntasks, int max_nd, int max_nt);

/* ---- appearance ---------------------------------------------------------------------- */
typedef struct mot_cos_task {
  int32_t n, m, d;
  const float* a; int32_t lda; const int32_t* aidx; /* track features, rows of length d (gathered) */
  const float* b; int32_t ldb; const int32_t* bidx; /* detection features                          */
  float* out; int32_t ldo;                          /* n x m: max(0, 1 - a.b / (|a||b| + 1e-10))   */
  float* norm_a; float* norm_b;                     /* unused (the row norms are accumulated inside the kernel); may be NULL */
} mot_cos_task;
int mot_cosine_cost(mot_ctx* ctx, const mot_cos_task* tasks, int ntasks, int max_n, int max_m);
/* the other metrics over the same task layout: MOT_EMB_COSINE = mot_cosine_cost; MOT_EMB_DOT: out = a_i . b_j (the raw
 * similarity DeepOC-SORT builds its embedding cost from, src/trackers/deepocsort.cpp:404; fp32 MFMA, k-ordered);
 * MOT_EMB_EUCLIDEAN: out = |a_i - b_j| (utils::embedding_distance(metric = "euclidean"), src/utils/matching.cpp:93-101) */
typedef enum mot_emb_metric { MOT_EMB_COSINE = 0, MOT_EMB_DOT = 1, MOT_EMB_EUCLIDEAN = 2 } mot_emb_metric;
int mot_embedding_cost(mot_ctx* ctx, int metric, const mot_cos_task* tasks, int ntasks, int max_n, int max_m);

typedef struct mot_feat_task {
  int32_t n, d;
  float* feat; int32_t ldf; const int32_t* slot;          /* destination rows feat[slot[i]*ldf ..]   */
  const float* src; int32_t lds; const int32_t* sidx;     /* raw detection feature rows              */
  const float* alpha_i;                                   /* optional [n]: per-item EMA weight (DeepOC-SORT's dets_alpha,
                                                             deepocsort.cpp:646-648); NULL: `alpha` for every item */
  int32_t mode;                                           /* 0: set = src/|src|, 1: EMA then renormalise,
                                                             2: src/|src| only where |src| > 1e-6 (ReIDBackend::normalize_features,
                                                                src/appearance/reid_backend.cpp:72-88; DeepOCSortKalmanBoxTracker
                                                                ctor deepocsort.cpp:73-79), 3: EMA then normalise where the norm
                                                                exceeds 1e-6 (update_emb, deepocsort.cpp:132-150),
                                                             4: StrongSORT's Track::update (strongsort.cpp:165-182): EMA of the stored feature with an already
                                                                normalised src, renormalised when the norm exceeds 1e-10 — else the stored feature stays,
                                                             5: src/|src| where |src| > 1e-10, else src (cosine_distance's row normalisation :317-331; a new
                                                                track's feature :84-91), 6: plain copy */
  float alpha;                                            /* EMA weight of the old feature (0.9)     */
} mot_feat_task;
int mot_feat_update(mot_ctx* ctx, const mot_feat_task* tasks, int ntasks, int max_n);

/* ---- StrongSORT's cost matrices (src/trackers/strongsort.cpp) ------------------------------ */
/* mot_ss_nn_cost: NearestNeighborDistanceMetric::distance (:239-275) — cost[i][j] = min over the samples s of track i of 1 - dots[s][j],
 * dots = inner products of the re-normalised sample rows with the re-normalised detection features (mot_embedding_cost, MOT_EMB_DOT);
 * soff [n + 1]: track i's samples are the rows soff[i] .. soff[i+1]-1 of dots; no sample: 1e5.
 * mot_ss_iou_cost: iou_matching::iou_cost (:500-583) + min_cost_matching's clamp (:376-379): 1 - IoU between the tlwh box of track i's
 * Kalman state (record `src[i]` of `mean`, Track::to_tlwh :94-100) and detection didx[j] of the tlwh planes dtlwh [4][ldd]
 * (mot_det_prepare, MOT_DET_TLWH); stale[i] != 0 (time_since_update > 1): 1e5; results above max_dist become max_dist + 1e-5. */
typedef struct mot_ss_nn_task {
  int32_t n, m;
  const float* dots; int32_t ldd;
  const int32_t* soff;
  float* cost; int32_t ldc;
} mot_ss_nn_task;
int mot_ss_nn_cost(mot_ctx* ctx, const mot_ss_nn_task* tasks, int ntasks, int max_n, int max_m);
typedef struct mot_ss_iou_task {
  int32_t n, m;
  const float* mean; const int32_t* src; const uint8_t* stale;
  const float* dtlwh; int32_t ldd; const int32_t* didx;
  float* cost; int32_t ldc;
  float max_dist;
} mot_ss_iou_task;
int mot_ss_iou_cost(mot_ctx* ctx, const mot_ss_iou_task* tasks, int ntasks, int max_n, int max_m);

/* ---- UCMCTrack's ground-plane filter (src/trackers/ucmc.cpp) ------------------------------- */
/* The reference keeps this tracker's state in double precision: [x, vx, y, vy] and a 4 x 4 covariance per track (UCMCKalmanFilter
 * :15-49), measurements = a detection's foot point mapped to the ground plane with a 2 x 2 covariance (CameraMapper :57-146). One task
 * = one stream's share of a launch; `x` [cap][4] and `P` [cap][16] (row-major) are the stream's state slabs, `y` [nd][2] / `R` [nd][4]
 * its mapped detections. Arithmetic: IEEE double, the operation order of the reference's expressions (sums of more than two non-zero
 * terms in index order, no fused multiply-add); S^-1 as Eigen forms a 2 x 2 inverse (1 / det, scaled adjugate); log = the device's
 * double-precision log (the one value that is not correctly rounded).
 *   MOT_UCMC_MAP     n detections: didx[i] (NULL: i) = column of `dets` (SoA [6][ld] floats x1,y1,x2,y2,conf,cls) -> y[i], R[i]
 *                    (mapped != 0: uv2xy with invA, else mapToImageSpace's scaled image coordinates)
 *   MOT_UCMC_PREDICT n tracks slots[i]: x = F x, P = F P F^T + Q (UCMCKalmanFilter::predict; F = I + dt at (0,1), (2,3))
 *   MOT_UCMC_COST    cost[i * ldc + j] = (float)(Mahalanobis + log det S) of track slots[i] and detection didx[j]
 *                    (UCMCSingleTrack::distance :213-223), n x m
 *   MOT_UCMC_UPDATE  n pairs: track slots[i] with detection didx[i] (UCMCKalmanFilter::update, Joseph form :33-49)
 *   MOT_UCMC_INIT    n births: track slots[i] from detection didx[i] (UCMCSingleTrack ctor :152-201: zero velocity, P = diag(1, vmax^2/3, 1, vmax^2/3)) */
enum { MOT_UCMC_MAP = 0, MOT_UCMC_PREDICT = 1, MOT_UCMC_COST = 2, MOT_UCMC_UPDATE = 3, MOT_UCMC_INIT = 4 };
typedef struct mot_ucmc_task {
  int32_t n, m, mapped, ld;
  double* x; double* P;
  const int32_t* slots; const int32_t* didx;
  double* y; double* R;
  const float* dets;
  float* cost; int32_t ldc, reserved;
  double dt, vmax;
  double Q[16];
  double invA[9];
} mot_ucmc_task;
int mot_ucmc_run(mot_ctx* ctx, int op, const mot_ucmc_task* tasks, int ntasks, int max_n, int max_m);

/* ---- BoostTrack's filter and costs (src/trackers/boosttrack.cpp) ----------------------------- */
/* State [cx, cy, h, r, vx, vy, vh, vr] with constant noise (BoostKalmanFilter :22-75): records of 72 floats (mean[8], covariance[64]) in
 * `slab`, like the other 8-state filters. Detections: SoA planes [6][ldd] (x1, y1, x2, y2, conf, cls).
 *   MOT_BOOST_PREDICT n tracks slots[i]: x = F x, P = F P F^T + Q; boxes[i*4..] = get_state() of the predicted state (:107-115)
 *   MOT_BOOST_DLO     n detections x m tracks: max_s[i] = max_j iou_batch(det i, box j) (:376-391), vt[i] = 1 when some j has
 *                     iou > max(0.95 - (tsu[j] - 1), 0.8) (:409-424); boxes = the predicted boxes [m][4]
 *   MOT_BOOST_COST    cost[i * ldc + j] = (1 - IoU as get_iou_matrix computes it, :297-329) - lambda_mhd * (limit - min(mh, limit)) / limit,
 *                     mh = the diagonal Mahalanobis distance of detection didx[i] to track slots[j] (:331-359, :598-611); n detections x m tracks
 *   MOT_BOOST_UPDATE  n pairs: track slots[i] with detection didx[i] (BoostKalmanFilter::update :61-75; S^-1 = partial-pivot LU inverse)
 *   MOT_BOOST_INIT    n births: track slots[i] from detection didx[i] (:22-54)
 *   MOT_BOOST_BOXES   boxes[i*4..] = get_state() of track slots[i] */
enum { MOT_BOOST_PREDICT = 0, MOT_BOOST_DLO = 1, MOT_BOOST_COST = 2, MOT_BOOST_UPDATE = 3, MOT_BOOST_INIT = 4, MOT_BOOST_BOXES = 5 };
typedef struct mot_boost_task {
  int32_t n, m, ldd, ldc;
  float* slab;
  const int32_t* slots; const int32_t* didx;
  const float* dets;
  float* boxes;
  const int32_t* tsu;
  float* max_s; int32_t* vt;
  float* cost;
  float lambda_mhd, lambda_emb;
  const float* emb; int32_t lde, reserved;  /* COST, optional: emb[i * lde + j] = <raw embedding of detection i, stored embedding of track j>;
                                             cost -= lambda_emb * (emb + 1) / 2 (:613-618) */
} mot_boost_task;
int mot_boost_run(mot_ctx* ctx, int op, const mot_boost_task* tasks, int ntasks, int max_n, int max_m);

/* ---- HybridSORT's filter and pairwise costs (src/trackers/hybridsort.cpp) -------------------- */
/* State [u, v, s, c, r, du, dv, ds, dc] (HybridKalmanFilter :26-88; c = the detection confidence), records of 90 floats (mean[9],
 * covariance[81]) in `slab`. Detections: SoA planes [6][ldd].
 *   MOT_HYB_PREDICT n tracks slots[i]: HybridKalmanBoxTracker::predict's guard (ds + s <= 0 -> ds = 0, :257-259), x = F x, P = F P F^T + Q
 *   MOT_HYB_UPDATE  n tracks slots[i] with measurement convert_bbox_to_z(detection didx[i]) (:181-193), or the ALL-ZERO measurement when
 *                   didx[i] < 0 (what the reference feeds an unmatched track, :315-320); K = P H^T S^-1 with the partial-pivot LU inverse
 *                   of the 5 x 5, P = (I - K H) P (:72-88)
 *   MOT_HYB_INIT    n births slots[i] from detection didx[i]
 *   MOT_HYB_BOXES   boxes[i*4..] = convert_x_to_bbox of track slots[i] (:195-201)
 *   MOT_HYB_PAIR    n x m: sim[i*ldc+j] = IoU (:529-556; hmiou != 0: x the height overlap, :558-577) of box a[i] and box b[j] (plain [k][4]
 *                   arrays) - score_w * |b_score[j] - a_score[i]| (score_w = 0: no score term); cost = (1 - sim) * 1 + add_const */
enum { MOT_HYB_PREDICT = 0, MOT_HYB_UPDATE = 1, MOT_HYB_INIT = 2, MOT_HYB_BOXES = 3, MOT_HYB_PAIR = 4 };
typedef struct mot_hyb_task {
  int32_t n, m, ldd, ldc;
  float* slab;
  const int32_t* slots; const int32_t* didx;
  const float* dets;
  float* boxes;
  const float* a; const float* b; const float* a_score; const float* b_score;
  float* sim; float* cost;
  int32_t hmiou, scale_first;
  float score_w, add_const;
} mot_hyb_task;
int mot_hyb_run(mot_ctx* ctx, int op, const mot_hyb_task* tasks, int ntasks, int max_n, int max_m);

/* ---- linear assignment ---------------------------------------------------------------- */
typedef enum mot_lap_mode {
  MOT_LAP_PLAIN = 0,
  MOT_LAP_GATE_MIN = 1, /* solve only if min(cost) < gate, else leave everything unmatched (ocsort.cpp:443,499) */
  MOT_LAP_OCSORT = 2    /* if every row/col of (iou > gate) has <= 1 hit and one exists: take those pairs (ocsort.cpp:684-696) */
} mot_lap_mode;
typedef struct mot_lap_task {
  int32_t n, m;
  const float* cost; int32_t ldc;
  float thresh;
  int32_t* x;   /* out [n]: column matched to row i or -1                       */
  int32_t* y;   /* out [m]: row matched to column j or -1                       */
  int32_t mode;
  const float* iou; int32_t ldi; float gate;
  float* xval;  /* optional out [n]: iou (if given, else cost) at (i, x[i])     */
  int32_t* info;/* optional out [1]: 0 lapjv, 1 trivial shortcut, 2 gated off,
                 * -1 refused (MOT_LAP_F_PLAIN given but geom.mode is MOT_COST_BOTSORT: everything unmatched) */
  void* work;   /* REQUIRED scratch of mot_lap_work_bytes(n, m) bytes (cold path arrays, staged boxes, overflow of the LDS state) */
  /* on-the-fly cost: when geom.a != NULL the cost of pair (i,j) is recomputed inside the solver from geom's row /
   * column boxes with mot_iou_cost's arithmetic for geom.mode (cost, ldc are ignored; geom.cost/pairs unused): the
   * N x M matrix is never materialised. geom.emb (BOTSORT) is still read from memory, for overlapping pairs only. */
  mot_iou_task geom;
  long long* prof; /* optional out [36]: shader cycles per solver phase, pass counts, shortest-path scan counters and cycles (diagnostics) */
  void* rowlist;   /* optional scratch of mot_lap_rowlist_bytes(n) bytes for a task with a cost MATRIX (geom.a == NULL): the exact
                    * solver gathers, per row, the entries below thresh/2 there and runs lapjv's shortest-path scans over those
                    * lists (many SCAN members per step) instead of one dense row sweep per member — same decisions, proved in
                    * lap_core.hpp; what OC-SORT's first association (4096 x 2048, src/trackers/ocsort.cpp:700-701) needs. NULL:
                    * dense sweeps. */
} mot_lap_task;
enum {
  MOT_LAP_F_GEOM = 1, /* some task carries geom: reserve LDS for the staged boxes */
  MOT_LAP_F_ASSOC = 2, /* some geom.assoc != MOT_ASSOC_IOU: run the variants compiled with every association measure
                         (the default variants evaluate plain IoU only and ignore geom.assoc) */
  MOT_LAP_F_PLAIN = 4 /* no geom.mode is MOT_COST_BOTSORT: the variants without the gated appearance term may run */
};
size_t mot_lap_work_bytes(int n, int m);
size_t mot_lap_rowlist_bytes(int n);
/* BoT-SORT's appearance term (src/trackers/botsort.cpp:433-466): the reference forces the distance of every pair whose IoU distance exceeds
 * proximity_thresh to 1 (:439-447), so only the other pairs' cosine distances can reach the assignment. tasks[s] (mot_cosine_cost's layout) is paired
 * with lap[s * lap_stride]: that task's geom (boxes, MOT_COST_BOTSORT, prox_thresh) says which pairs pass the test — evaluated with the arithmetic
 * the solvers use — and out[i][j] is written for exactly those (bit-identical to mot_cosine_cost's entry); every other entry of out is left
 * untouched (no solver reads it). A geom without boxes, with another cost mode, or with prox_thresh >= 1 gets the whole matrix. */
int mot_cosine_cost_gated(mot_ctx* ctx, const mot_cos_task* tasks, const mot_lap_task* lap, int lap_stride, int ntasks, int max_n, int max_m);
/* mot_lap_solve runs two kernels over the task array: a fast path (viable pairs only, shortest augmenting paths, and a
 * certificate that the optimum is unique — then it IS lapjv's answer) and, for the problems the fast path does not certify, the
 * step-by-step lapjv emulation that reproduces the reference's tie-breaks. Diagnostics: outcome counts of the fast path on this
 * device since the last reset (synchronises the context's stream): [0] finished by the fast path, [1] declined while listing the
 * viable pairs (a cost within 1e-9 of the threshold, NaN/inf, more than 16 viable pairs in a column, ...), [2] a path search
 * grew too large, [3] certificate arithmetic, [4] too many tight pairs, [5] optimum not unique (a tie), [6] not attempted
 * (prof requested, cost flavour), [7] empty problems; [8..11] summed shader cycles of the fast path's stages (listing the viable
 * pairs, initial matching, path searches, certificate), [12] path searches, [13] column scans inside them, [14] declined: a column with more than 16 viable
 * pairs, [15] declined: more viable pairs than the list holds; [16..18] cycles inside [8]: bucketing
 * the rows, the candidate sweep, the list build; [19] candidate rows looked at by lane 0, [20] pairs it evaluated; declined: [21] NaN / inf /
 * out-of-range input, [22] pairs that do not intersect would be viable, [23] a cost within 1e-9 of the threshold; [24..27] cycles inside [10]
 * (four-wavefront launches): fetching a column's pairs, delivering labels, picking the nearest row, dual update + augmentation;
 * [28..31] reserved. */
int mot_lap_fast_stats(mot_ctx* ctx, unsigned long long* out32, int reset);
/* Diagnostics of the problems the exact emulation solved BEHIND the fast path on this device since the last reset: out80[0..35] = the
 * counters of mot_lap_task.prof summed over those problems, out80[39] = how many; out80[40..75] = the counters of the slowest one,
 * out80[79] = its shader cycles (phases 1a + 1b + 2 + 3). Synchronises the context's stream. */
int mot_lap_behind_stats(mot_ctx* ctx, long long* out80, int reset);
int mot_lap_solve(mot_ctx* ctx, const mot_lap_task* tasks, int ntasks, int max_n, int max_m, int flags);

/* ---- ByteTrack with the per-stream lifecycle on the device ------------------------------ */
/* S independent ByteTrack streams whose whole update() (src/trackers/bytetrack.cpp:166-706: detection split, pools,
 * three associations, Kalman updates, list algebra, duplicate removal, output table) runs on the GPU; the host only
 * enqueues a fixed sequence of launches per frame. params: [min_conf, track_thresh, match_thresh, track_buffer, frame_rate].
 * cap_tracks bounds tracked+lost tracks per stream, max_dets the detections per frame (exceeding either sets an error
 * flag that mot_bt_step returns as MOT_ERR_CAPACITY). */
typedef struct mot_bt_batch mot_bt_batch;
int mot_bt_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, const float* params5, mot_bt_batch** out);
void mot_bt_destroy(mot_bt_batch* b);
int mot_bt_reset(mot_bt_batch* b);
/* d_dets: device, SoA [S][6][max_dets]; h_counts: host [S]; out (host) [S][cap_out][8] rows x1,y1,x2,y2,id,conf,cls,det_ind;
 * out_counts (host) [S]. Synchronous: returns when the outputs are in host memory. */
int mot_bt_step(mot_bt_batch* b, const float* d_dets, const int* h_counts, float* out, int* out_counts, int cap_out);
/* Same frame, packed output: the emitted rows of all streams back to back (stream s's rows start at the sum of the counts
 * before it), so that only rows that exist cross PCIe and no stream has a row limit short of cap_tracks (the reference's
 * ByteTrack can emit more rows than the frame has detections). rows (host) [rows_cap][8]; out_counts (host) [S];
 * *total_rows = rows written. MOT_ERR_CAPACITY if the frame has more than rows_cap rows (nothing is copied). */
int mot_bt_step_packed(mot_bt_batch* b, const float* d_dets, const int* h_counts, float* rows, int rows_cap, int* out_counts,
                       int* total_rows);
/* device-resident result of the last mot_bt_step_packed: packed rows [total][8], offsets [S+1] (offsets[S] = total), counts [S] —
 * what a gather over xGMI reads directly (no copy through the host) */
int mot_bt_device_output(mot_bt_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts);
/* Frames in flight: mot_bt_step_packed split in two so that ONE host thread overlaps the result copy of frame f with the
 * kernels of frame f + 1. mot_bt_enqueue_packed queues a frame's launches and returns (at most two frames may be pending;
 * h_counts is copied before the call returns, but d_dets — and d_embs for mot_bot_* — are read by the queued kernels: the buffers must
 * stay unmodified until the matching collect has returned, i.e. a caller with two frames in flight needs two input buffers);
 * mot_bt_collect_packed waits for the OLDEST pending frame only and delivers its
 * packed rows (copied on a second stream while the next frame runs); MOT_ERR_CAPACITY when the frame has more rows than the rows_cap
 * of ITS enqueue call or of this collect call. rows == NULL: nothing is copied — the frame's table stays on the device, where
 * mot_bt_device_output finds it (a consumer on the GPU, or the RCCL gather, reads it there). Frames come back in the order they went in; mixing with mot_bt_step /
 * mot_bt_step_packed while frames are pending is an error (MOT_ERR_INVALID); mot_bt_reset drops the frames still pending (and, like
 * ByteTrack::reset, keeps the id counters running: clear_count() is empty in the reference — same for mot_oc_reset, mot_sort_reset;
 * mot_bot_reset restarts them). A frame whose table does not fit the enqueue call's rows_cap, or the collect call's buffer, is reported
 * by the collect (MOT_ERR_CAPACITY) and consumed; the tracks are not affected (tests/test_gpu_inflight_misuse.py). */
int mot_bt_enqueue_packed(mot_bt_batch* b, const float* d_dets, const int* h_counts, int rows_cap);
int mot_bt_collect_packed(mot_bt_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows);
/* parity hook: ids and Kalman states of stream s's live tracks in list order (active then lost): ids [cap], mean [cap][8],
 * cov [cap][64]; returns the number of tracks */
int mot_bt_dump(mot_bt_batch* b, int s, int* ids, float* mean, float* cov, int cap);
/* HIP-event timing of the assignment launches (enable = 1 resets). out8: [0] summed ms of the first-association launches,
 * [1] of the second/unconfirmed launches, [2] of whole frames on the stream, [3] frames; problems queued and the sum of their
 * n + m: [4],[5] first association, [6],[7] second + unconfirmed */
int mot_bt_profile(mot_bt_batch* b, int enable);
int mot_bt_profile_stats(mot_bt_batch* b, double* out8);
/* the achieved problem sizes behind those counts: summed rows (tracks) and columns (detections) of the queued problems,
 * [0],[1] first association, [2],[3] second + unconfirmed (divide by out8[4] / out8[6] for the mean N x M) */
/* [0] summed HIP-event ms of the FIRST association's sparse-solver kernel alone (lap_sparse_kernel; the exact kernel behind it is
 * not included), [1] its launches, since profiling was switched on */
int mot_bt_profile_lap_sparse(mot_bt_batch* b, double* out2);
int mot_bt_profile_dims(mot_bt_batch* b, double* out4);
/* the Kalman launches of the same frames: summed ms and items of [0],[1] the box-only prediction of the pool (32 B of mean read,
 * 16 B box written per track), [2],[3] initiations, [4],[5] predict-first updates (one 288-byte record read and written) */
int mot_bt_profile_kalman(mot_bt_batch* b, double* out6);

/* ---- BoT-SORT with the per-stream lifecycle on the device -------------------------------- */
/* Same contract as mot_bt_* for BotSort::update (src/trackers/botsort.cpp:260-764): S independent streams, one fixed launch
 * sequence per frame, packed output rows [x1,y1,x2,y2,id,conf,cls,det_ind]. params10: [track_high_thresh, track_low_thresh,
 * new_track_thresh, track_buffer, match_thresh, proximity_thresh, appearance_thresh, frame_rate, fuse_first_associate,
 * with_reid] (botsort.hpp:108-144 defaults when NULL). emb_dim > 0 reserves the smooth-feature slab [S][cap_tracks][emb_dim].
 * Per frame: d_dets SoA [S][6][max_dets] (device), h_counts [S] (host; a stream with 0 detections is left untouched,
 * botsort.cpp:267-269), d_embs [S][max_dets][emb_dim] row-major raw detection features (device; NULL: no features this frame),
 * h_warps6 [S][6] + h_has_warp [S]: the 2x3 camera-motion warp of streams that have one for this frame (what
 * cmc_->apply(img, dets) returns, :317-324; NULL: none). mot_bot_dump: live tracks of stream s, tracked list then lost list:
 * ids, mean [cap][8], cov [cap][64], feats [cap][emb_dim] (may be NULL), has_feat [cap] (may be NULL). */
typedef struct mot_bot_batch mot_bot_batch;
int mot_bot_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, int emb_dim, const float* params10, mot_bot_batch** out);
void mot_bot_destroy(mot_bot_batch* b);
int mot_bot_reset(mot_bot_batch* b);
int mot_bot_step_packed(mot_bot_batch* b, const float* d_dets, const int* h_counts, const float* d_embs, const float* h_warps6,
                        const unsigned char* h_has_warp, float* rows, int rows_cap, int* out_counts, int* total_rows);
int mot_bot_device_output(mot_bot_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts);
/* frames in flight, as mot_bt_enqueue_packed / mot_bt_collect_packed (h_counts and the warps are copied before the call returns) */
int mot_bot_enqueue_packed(mot_bot_batch* b, const float* d_dets, const int* h_counts, const float* d_embs, const float* h_warps6,
                           const unsigned char* h_has_warp, int rows_cap);
int mot_bot_collect_packed(mot_bot_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows);
int mot_bot_dump(mot_bot_batch* b, int s, int* ids, float* mean, float* cov, float* feats, unsigned char* has_feat, int cap);
/* HIP-event timing (enable = 1 resets). out8: [0] summed ms of the three assignment launches, [1] of the first cosine launch,
 * [2] of whole frames, [3] frames, [4] assignment problems queued, [5] sum of their n + m, [6] sum of n*m over the cosine
 * problems, [7] emb_dim */
int mot_bot_profile(mot_bot_batch* b, int enable);
int mot_bot_profile_stats(mot_bot_batch* b, double* out8);
/* the appearance-feature maintenance of the profiled frames (feat_kernel: normalise the detections' rows, set the new tracks' features, blend
 * the matched ones — botsort.cpp:38-46, :158-169). out2: [0] summed ms of its three launches per frame, [1] feature rows moved (a read or a
 * write of emb_dim floats each: normalise / set = 2 per row, blend = 3) */
int mot_bot_profile_feat(mot_bot_batch* b, double* out2);

/* ---- OC-SORT with the per-stream lifecycle on the device --------------------------------- */
/* Same contract as mot_bt_* for OCSort::update (src/trackers/ocsort.cpp:285-606): S independent streams, one fixed launch
 * sequence per frame (predict, NaN-row rule, IoU + velocity-direction cost, trivial-case shortcut or lapjv, optional BYTE
 * stage, observation-centric rematch, spawns, Kalman updates, output rows newest tracker first), packed output rows
 * [x1,y1,x2,y2,id,conf,cls,det_ind]. params14: [det_thresh, max_age, max_obs (unused), min_hits, iou_threshold, min_conf,
 * delta_t, inertia, use_byte, Q_xy_scaling, Q_s_scaling, asso (mot_assoc), frame_w, frame_h] (ocsort.hpp:88-108 defaults when
 * NULL). A track that quirk Q4 has updated more than 4 times in one frame raises MOT_ERR_CAPACITY (never observed).
 * mot_oc_dump: the tracker list of stream s in list order: ids, mean [cap][7], cov [cap][49]. */
typedef struct mot_oc_batch mot_oc_batch;
int mot_oc_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, const float* params14, mot_oc_batch** out);
void mot_oc_destroy(mot_oc_batch* b);
int mot_oc_reset(mot_oc_batch* b);
int mot_oc_step_packed(mot_oc_batch* b, const float* d_dets, const int* h_counts, float* rows, int rows_cap, int* out_counts, int* total_rows);
int mot_oc_device_output(mot_oc_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts);
/* frames in flight, as mot_bt_enqueue_packed / mot_bt_collect_packed (same contract: at most two pending, d_dets unmodified until the
 * matching collect, the enqueue call's rows_cap bounds the frame). OCSort::update, src/trackers/ocsort.cpp:285-606. */
int mot_oc_enqueue_packed(mot_oc_batch* b, const float* d_dets, const int* h_counts, int rows_cap);
int mot_oc_collect_packed(mot_oc_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows);
int mot_oc_dump(mot_oc_batch* b, int s, int* ids, float* mean, float* cov, int cap);
/* out8: [0] summed ms of the first-association assignment launches, [1] of the cost-matrix launches, [2] of whole frames,
 * [3] frames, [4] first-association problems queued, [5] sum of their n*m */
int mot_oc_profile(mot_oc_batch* b, int enable);
int mot_oc_profile_stats(mot_oc_batch* b, double* out8);

/* ---- SORT with the per-stream lifecycle on the device ----------------------------------- */
/* Same contract as mot_bt_* for Sort::update (src/trackers/sort.cpp:102-255). params: [det_thresh, max_age, max_obs (unused),
 * min_hits, iou_threshold]. mot_sort_reset keeps the id counters running (sort.cpp:97-100). mot_sort_dump: mean [cap][7],
 * cov [cap][49] of the live tracks in list order. */
typedef struct mot_sort_batch mot_sort_batch;
int mot_sort_create(mot_ctx* ctx, int nstreams, int cap_tracks, int max_dets, const float* params5, mot_sort_batch** out);
void mot_sort_destroy(mot_sort_batch* b);
int mot_sort_reset(mot_sort_batch* b);
int mot_sort_step(mot_sort_batch* b, const float* d_dets, const int* h_counts, float* out, int* out_counts, int cap_out);
/* packed output and frames in flight, as mot_bt_step_packed / mot_bt_enqueue_packed / mot_bt_collect_packed / mot_bt_device_output
 * (same contracts). Sort::update, src/trackers/sort.cpp:102-255. */
int mot_sort_step_packed(mot_sort_batch* b, const float* d_dets, const int* h_counts, float* rows, int rows_cap, int* out_counts, int* total_rows);
int mot_sort_enqueue_packed(mot_sort_batch* b, const float* d_dets, const int* h_counts, int rows_cap);
int mot_sort_collect_packed(mot_sort_batch* b, float* rows, int rows_cap, int* out_counts, int* total_rows);
int mot_sort_device_output(mot_sort_batch* b, const float** d_rows, const int** d_offsets, const int** d_counts);
int mot_sort_dump(mot_sort_batch* b, int s, int* ids, float* mean, float* cov, int cap);
int mot_sort_profile(mot_sort_batch* b, int enable);         /* same layout as mot_bt_profile_stats ([1], [6], [7] = 0) */
int mot_sort_profile_stats(mot_sort_batch* b, double* out8);

/* ---- pooled streams (round 4): the streams of a device-lifecycle batch as independent tracker objects ----------------- */
/* The reference's model is one BaseTracker object per camera, each updated from its own thread
 * (include/motcpp/tracker.hpp:67-69, docs/guides/architecture.md:242-255). The host runtime (csrc/host/pool.cpp) gives every such object
 * one stream of a shared batch and merges the update() calls that arrive together into ONE launch sequence; these entry points are what
 * it needs beyond the lockstep API above:
 *   mot_*_enqueue_frame  a frame in which only the streams with h_counts[s] >= 0 take part (the others keep their state untouched and
 *                        report 0 rows), detections packed back to back: stream s's SoA planes [6][h_det_ld[s]] start at
 *                        d_dets + h_det_off[s] floats (a column-major N x 6 matrix as the caller holds it, ld = N rounded up as it likes).
 *                        The host arrays are copied before the call returns; d_dets (and d_embs) must stay unmodified until the
 *                        matching collect. At most two frames in flight, as mot_*_enqueue_packed.
 *   mot_*_collect_view   waits for the oldest frame in flight; rows / counts / alive point into page-locked memory owned by the batch —
 *                        the kernels wrote the rows there directly, no copy is made — valid until the second mot_*_enqueue_* after
 *                        this call. Stream s's rows start at the sum of counts[0..s). alive[s] = live tracks of stream s after the frame.
 *   mot_*_reset_stream   fresh == 0: BaseTracker::reset() of one stream as the reference does it — the tracks go; the id counter keeps
 *                        counting for SORT / ByteTrack / OC-SORT (sort.cpp:97-100; STrack::clear_count and KalmanBoxTracker::clear_count
 *                        are empty, bytetrack.hpp:38-40, ocsort.hpp:37-39) and restarts for BoT-SORT (botsort.cpp:252-258). fresh != 0: the
 *                        stream as created (ids from the start: the slot now belongs to a new tracker object). Asynchronous on the batch's stream.
 *   mot_*_move_stream    moves one stream into a batch with larger capacities (same parameters, same device): how a tracker object
 *                        outgrows cap_tracks / max_dets without ever seeing MOT_ERR_CAPACITY. Synchronous. */
typedef struct mot_frame_in {
  const float* d_dets;
  const int* h_counts;
  const int* h_det_ld;
  const long long* h_det_off;
  const float* d_embs;              /* BoT-SORT: raw detection features, row-major [n][emb_dim] per stream (NULL: no stream has any) */
  const long long* h_emb_off;       /* BoT-SORT: float offset of stream s's rows in d_embs, < 0: none for this stream in this frame */
  const float* h_warps6;            /* BoT-SORT: [S][6] camera-motion warps, h_has_warp [S] (NULL: none) */
  const unsigned char* h_has_warp;
} mot_frame_in;
typedef struct mot_frame_view {
  const float* rows;
  const int* counts;
  const int* alive;  /* [S] live tracks after the frame; a stream that raised an error (capacity) reports -(its error code) here: when
                        mot_*_collect_view returns MOT_ERR_CAPACITY the view is still filled, the rows of the streams with alive >= 0 are
                        valid, and only the callers of the others need to see the error (the reference's exceptions are per object:
                        src/tracker.cpp:108-125) */
  int total;
} mot_frame_view;
int mot_bt_enqueue_frame(mot_bt_batch* b, const mot_frame_in* in, int rows_cap);
int mot_bt_collect_view(mot_bt_batch* b, mot_frame_view* out);
int mot_bt_reset_stream(mot_bt_batch* b, int s, int fresh);
int mot_bt_move_stream(mot_bt_batch* src, int s, mot_bt_batch* dst, int s2);
int mot_sort_enqueue_frame(mot_sort_batch* b, const mot_frame_in* in, int rows_cap);
int mot_sort_collect_view(mot_sort_batch* b, mot_frame_view* out);
int mot_sort_reset_stream(mot_sort_batch* b, int s, int fresh);
int mot_sort_move_stream(mot_sort_batch* src, int s, mot_sort_batch* dst, int s2);
int mot_oc_enqueue_frame(mot_oc_batch* b, const mot_frame_in* in, int rows_cap);
int mot_oc_collect_view(mot_oc_batch* b, mot_frame_view* out);
int mot_oc_reset_stream(mot_oc_batch* b, int s, int fresh);
int mot_oc_move_stream(mot_oc_batch* src, int s, mot_oc_batch* dst, int s2);
int mot_bot_enqueue_frame(mot_bot_batch* b, const mot_frame_in* in, int rows_cap);
int mot_bot_collect_view(mot_bot_batch* b, mot_frame_view* out);
int mot_bot_reset_stream(mot_bot_batch* b, int s, int fresh);
int mot_bot_move_stream(mot_bot_batch* src, int s, mot_bot_batch* dst, int s2);

/* ---- multi-GPU: gather of the track tables over RCCL ------------------------------------ */
/* one process per GPU, rank r owns streams [r*S, (r+1)*S) and never exchanges tracker state; the only
 * collective is the gather of the output tables, done on the device tables (mot_bt_device_output) with RCCL on the context's
 * stream — the counterpart of nothing in the single-process reference (its trackers return one Eigen matrix per call).
 * mot_comm_unique_id: rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks through whatever channel the
 * host program has (torch.distributed, MPI, a file). mot_comm_create: collective over all ranks.
 * mot_comm_gather_tables: d_rows [sum(counts)][8] + d_counts [nstreams] of this rank -> d_rows_all (every rank's rows,
 * rank-major, then stream-major inside a rank, packed), h_counts_all [world][nstreams] on the host, h_rank_rows [world]
 * (optional) = rows per rank. Enqueued on the context's stream; returns after the counts are known (one stream
 * synchronisation), the row exchange itself completes in stream order. RCCL is loaded at first use (dlopen). */
typedef struct mot_comm mot_comm;
int mot_comm_unique_id(void* id128);
int mot_comm_create(mot_ctx* ctx, int world, int rank, const void* id128, mot_comm** out);
int mot_comm_destroy(mot_comm* comm);
int mot_comm_gather_tables(mot_comm* comm, const float* d_rows, const int* d_counts, int nstreams, float* d_rows_all, int rows_cap,
                           int* h_counts_all, int* h_rank_rows);

/* ---- host-pointer conveniences (synchronous; row-major matrices) ---------------------- */
int mot_iou_cost_host(mot_ctx* ctx, const float* a_xyxy, int n, const float* b_xyxy, int m,
                      const float* bconf_or_null, int mode, float* cost);
/* same with the similarity measure chosen (mot_assoc; frame size only matters for CENTROID):
 * replaces utils::AssociationFunction(w, h, name)(a, b), include/motcpp/utils/iou.hpp:371-414, with mode = MOT_COST_IOU */
int mot_assoc_cost_host(mot_ctx* ctx, const float* a_xyxy, int n, const float* b_xyxy, int m, const float* bconf_or_null,
                        int mode, int assoc, int frame_w, int frame_h, float* cost);
/* utils::fuse_iou(reid_cost, tracks_xyxy, detections_xyxy, confs) — src/utils/matching.cpp:109-128 (the confidences are unused there) */
int mot_fuse_iou_host(mot_ctx* ctx, const float* reid_cost, const float* a_xyxy, int n, const float* b_xyxy, int m, float* cost);
int mot_cosine_cost_host(mot_ctx* ctx, const float* a, int n, const float* b, int m, int d, float* out);
int mot_embedding_cost_host(mot_ctx* ctx, int metric, const float* a, int n, const float* b, int m, int d, float* out);
/* mot_cosine_cost_gated on host arrays (boxes row-major [n][4] / [m][4]); out is read first: entries of pairs that fail the test come back as given */
int mot_cosine_cost_gated_host(mot_ctx* ctx, const float* a, int n, const float* b, int m, int d, const float* a_xyxy, const float* b_xyxy,
                               int cost_mode, float prox_thresh, float* out);
/* mot_feat_update on host rows: feat [n][d] in/out (read by the EMA modes 1 and 3), src [n][d]; modes 0-3 as in mot_feat_task;
 * alpha_i: optional [n] per-row EMA weights (DeepOC-SORT's dets_alpha), NULL = `alpha` for every row */
int mot_feat_update_host(mot_ctx* ctx, int mode, float alpha, int n, int d, float* feat, const float* src);
int mot_feat_update_host_alpha(mot_ctx* ctx, int mode, float alpha, const float* alpha_i, int n, int d, float* feat, const float* src);
int mot_ocsort_cost_host(mot_ctx* ctx, const float* dets5, int nd, const float* trks4, int nt,
                         const float* vel2, const float* prev5, float vdc_weight, float* cost, float* iou);
int mot_lap_solve_host(mot_ctx* ctx, const float* cost, int n, int m, float thresh, int mode,
                       const float* iou_or_null, float gate, int* x, int* y, int* info_or_null);
/* assignment straight from boxes (on-the-fly cost, no matrix): cost_mode is a mot_cost_mode */
/* the same with the exact solver's per-phase shader cycles (diagnostics; see mot_lap_task.prof): [0] column minima, [1] reduction
 * transfer, [2] augmenting row reduction, [3] augmentation, [4..6] their pass counts, [7] n + m */
int mot_lap_solve_prof_host(mot_ctx* ctx, const float* cost, int n, int m, float thresh, int mode,
                            const float* iou_or_null, float gate, int* x, int* y, int* info, long long* prof8);
int mot_lap_geom_host(mot_ctx* ctx, const float* a_xyxy, int n, const float* b_xyxy, int m, const float* bconf_or_null,
                      int cost_mode, float thresh, int lap_mode, float gate, int* x, int* y, float* xval_or_null,
                      int* info_or_null, long long* prof8_or_null);
/* mean: n x d, cov: n x d x d row-major (AoS); op: 0 initiate (mean/cov out), 1 predict, 2 update */
int mot_kf_apply_host(mot_ctx* ctx, int kf_kind, int op, int n, const float* meas4, const float* q3_or_null,
                      const unsigned char* flags_or_null, float* mean, float* cov, float* boxes4_or_null);
/* mot_kf_warp on AoS host states; predict_first != 0: mot_kf_predict_warp instead (predict, then warp, one launch) */
/* mot_kf_update for host arrays with per-measurement confidences (NSA Kalman; XYAH only, conf NULL = 0) */
/* The XYAH update on block-form covariances (mot_kf_task.cov_blocks; ByteTrack's device lifecycle, kf_kernels.hip::kf_update_blocks_kernel): the same
 * operations as KalmanFilterXYAH::update on the non-zero terms (src/motion/kalman_filter.cpp:77-112), 96 bytes per track instead of 576. Host
 * convenience for tests: mean [n][8] and blocks [n][16] in/out, upd_flags [n] MOT_KF_* bits (or NULL), dense_flag [n] out, cov_dense [n][64] out
 * (the 8 x 8 covariance of the tracks the block kernel handed to the dense one). */
int mot_kf_update_blocks_host(mot_ctx* ctx, int n, const float* meas4, const unsigned char* upd_flags, float* mean, float* blocks,
                              unsigned char* dense_flag, float* cov_dense);
int mot_kf_update_conf_host(mot_ctx* ctx, int kf_kind, int n, const float* meas4, const float* conf_or_null, float* mean, float* cov);
int mot_kf_warp_host(mot_ctx* ctx, int kf_kind, int n, const float* warp9, int predict_first, const float* q3_or_null,
                     float* mean, float* cov, float* boxes4_or_null);

#ifdef __cplusplus
}
#endif
#endif /* MOTCPP_AMD_H_ */
