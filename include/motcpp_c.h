/* motcpp_c.h — flat C handles over the C++ tracker classes of libmotcpp.so (for ctypes / cgo / JNI style
 * bindings and for this repository's tests and bench.py). Matrices are ROW-major here.
 *
 * kind: 0 SORT, 1 ByteTrack, 2 OC-SORT, 3 BoT-SORT, 4 DeepOC-SORT, 5 StrongSORT. Parameter vectors (missing tail = reference defaults):
 *  SORT      [det_thresh, max_age, max_obs, min_hits, iou_threshold]
 *  ByteTrack [min_conf, track_thresh, match_thresh, track_buffer, frame_rate, max_age, max_obs]
 *  OC-SORT   [det_thresh, max_age, max_obs, min_hits, iou_threshold, min_conf, delta_t, inertia, use_byte, Q_xy, Q_s,
 *             asso (mot_assoc: 0 iou, 1 hmiou, 2 giou, 3 ciou, 4 diou, 5 centroid)]
 *  BoT-SORT  [track_high, track_low, new_track, track_buffer, match_thresh, proximity, appearance,
 *             frame_rate, fuse_first_associate, with_reid, max_age, max_obs]
 *  StrongSORT [min_conf, max_cos_dist, max_iou_dist, n_init, nn_budget, mc_lambda, ema_alpha, max_age]
 *  HybridSORT (8) [det_thresh, max_age, min_hits, iou_threshold, asso (0 iou, 1 hmiou), low_thresh, use_byte, track_thresh, EG_weight_high_score,
 *              EG_weight_low_score, TCM_first_step, TCM_byte_step, TCM_byte_step_weight, with_reid (zero-feature behaviour: no embeddings)]
 *  BoostTrack (7) [det_thresh, max_age, min_hits, iou_threshold, min_box_area, aspect_ratio_thresh, lambda_iou, lambda_mhd,
 *              lambda_shape, use_dlo_boost, use_duo_boost, dlo_boost_coef, use_sb, use_vt, with_reid (embeddings with update())]
 *  UCMCTrack (6) [det_thresh, max_age, a1, a2, wx, wy, vmax, fps, high_score] (dt = 1.0 / fps in double precision)
 * Functions return >= 0 on success and a negative value on error (motcpp_last_error() has the message).
 */
#ifndef MOTCPP_C_H_
#define MOTCPP_C_H_
#ifdef __cplusplus
extern "C" {
#endif

typedef struct motcpp_tracker motcpp_tracker;
typedef struct motcpp_batch motcpp_batch;

const char* motcpp_last_error(void);

motcpp_tracker* motcpp_tracker_create(int kind, const float* params, int nparams, int device);
/* The same tracker as a stream of a shared device-lifecycle batch (what the C++ classes Sort / ByteTrack / OCSort / BotSort are by
 * default since round 4, csrc/host/pool.hpp): update() calls that arrive together from different host threads — one handle per
 * thread, like one BaseTracker per camera thread in the reference — run as ONE launch sequence on the GPU. Same parameter vectors;
 * the assignment hooks (lap_count / lap_get) report nothing for such a handle. */
motcpp_tracker* motcpp_tracker_create_pooled(int kind, const float* params, int nparams, int device);
int motcpp_tracker_pool_level(motcpp_tracker* t); /* capacity level the stream sits on (0: 512 tracks x 256 detections, x4 per level), -1: not attached yet */
/* combiner counters of the process: [0] launch sequences (rounds) run, [1] stream-frames they carried, [2] streams moved to a larger
 * level, [3] most streams merged into one round; leader wall time in microseconds: [4] batching windows, [5] waiting for the callers'
 * copies and the previous table's readers, [6] the rounds themselves (uploads, launches, GPU), [7] of that: queueing the launches */
int motcpp_pool_stats(long* out8, int reset);
/* T tracker OBJECTS of the C++ classes (kind as above) on T host threads, each calling BaseTracker::update(dets, img) on its own
 * stream of host detections: dets [T][frames][max_n][6] row-major, counts [T][frames]. `warm` untimed frames per thread first, then
 * the threads meet at a barrier and the remaining frames are timed. out: [0] seconds of the timed part (first start to last end),
 * [1] frames timed, [2] output rows, [3] mean update() latency (ms), [4] worst update() latency (ms). checksum (optional, [T]): sum of
 * the ids of every output row per tracker (parity against a single-threaded run). Returns 0. */
int motcpp_bench_threads(int kind, const float* params, int nparams, int T, int frames, int warm, const float* dets, const int* counts,
                         int max_n, int device, double* out5, double* checksum);
void motcpp_tracker_destroy(motcpp_tracker* t);
int motcpp_tracker_reset(motcpp_tracker* t);
/* dets: n x 6 [x1,y1,x2,y2,conf,cls]; embs: n x d or NULL; out: cap x 8. Returns rows, or -(rows needed) - 1000000 if cap is too small. */
int motcpp_tracker_update(motcpp_tracker* t, const float* dets, int n, const float* embs, int d, float* out, int cap);
/* BoT-SORT: the 2x3 row-major warp the reference's cmc_->apply(img, dets) would return for the NEXT update (applied to
 * the predicted track states, botsort.cpp:317-324, BotSTrack::multi_gmc :60-91); it is consumed by that update. NULL
 * withdraws it. The image registration that estimates the warp (ECC/ORB/SOF) stays with the caller. Returns 0, or -1
 * for the other trackers. A tracker inside a batch takes its warp through motcpp_batch_tracker(b, s). */
int motcpp_tracker_set_camera_motion(motcpp_tracker* t, const float* warp2x3);
/* parity hooks: assignments solved during the last update, and the Kalman states of the live tracks */
int motcpp_tracker_lap_count(motcpp_tracker* t);
int motcpp_tracker_lap_get(motcpp_tracker* t, int k, int* n, int* m, int* x, int* y, int cap);
int motcpp_tracker_dump_states(motcpp_tracker* t, float* out, int cap_floats, int* width);
/* UCMCTrack: creation with a calibrated camera (Ki 3 x 4, Ko 4 x 4, row-major; NULL: image-space fallback) and its double-precision
 * states: rows of 26 doubles [id, state, death_count, birth_count, det_idx, age, x(4), P(16)] in list order. */
motcpp_tracker* motcpp_ucmc_create(const float* params, int nparams, const double* Ki12, const double* Ko16, int device);
int motcpp_tracker_dump_f64(motcpp_tracker* t, double* out, int cap_rows);
/* BoT-SORT parity hook: smooth features (botsort.hpp smooth_feat_) of the live tracks in dump_states order, *dim floats per row
 * (zeros for a track that has none yet; *dim = 0 when the tracker holds no features); returns the number of rows */
int motcpp_tracker_dump_features(motcpp_tracker* t, float* out, int cap_floats, int* dim); /* rows [id, mean(d), cov(d*d)] */

/* S independent streams stepped in lockstep on one GPU (one kernel launch per kernel family per stage). */
motcpp_batch* motcpp_batch_create(int kind, const float* params, int nparams, int nstreams, int device);
/* same, but with its own device context (HIP stream + arenas): several batches can then be stepped concurrently from
 * different host threads, overlapping one batch's host lifecycle with another's kernels */
motcpp_batch* motcpp_batch_create_private(int kind, const float* params, int nparams, int nstreams, int device);
int motcpp_batch_profile(motcpp_batch* b, int enable);
int motcpp_batch_profile_stats(motcpp_batch* b, double* out_rows5, int cap_rows);
void motcpp_batch_destroy(motcpp_batch* b);
/* dets: [S][max_n][6] with counts[s] valid rows; embs: [S][max_n][d] or NULL; out: [S][cap][8]; out_counts: [S].
 * threads: host threads used for the per-stream lifecycle work (<= 1: caller thread only). */
int motcpp_batch_step(motcpp_batch* b, const float* dets, const int* counts, int max_n, const float* embs, int d,
                      float* out, int* out_counts, int cap);
/* same, with the detections ALSO already resident in HBM as SoA [S][6][max_n] (device pointer): nothing but the small
 * per-stage index lists crosses PCIe inside the call. The host copy is still needed for the lifecycle decisions. */
int motcpp_batch_step_resident(motcpp_batch* b, const float* dets, const int* counts, int max_n, const void* d_dets_soa,
                               const float* embs, int d, float* out, int* out_counts, int cap);
/* same, with the embeddings resident too: d_embs = device [S][max_n][d] row-major (BoT-SORT); no payload crosses PCIe */
int motcpp_batch_step_resident_embs(motcpp_batch* b, const float* dets, const int* counts, int max_n, const void* d_dets_soa,
                                    const void* d_embs, int d, float* out, int* out_counts, int cap);
int motcpp_batch_set_threads(motcpp_batch* b, int threads);
/* Pins the CALLING thread's worker team (set_threads many) to consecutive CPUs of the process's allowed set, starting at
 * its first_cpu-th one; call it from the thread that will call motcpp_batch_step. Sub-batches stepped from different
 * threads should get disjoint ranges. Optional: unpinned teams work, pinned ones step a frame about 1.5x faster. */
int motcpp_batch_pin_threads(motcpp_batch* b, int first_cpu);
int motcpp_batch_record_laps(motcpp_batch* b, int on); /* keep per-frame assignment records for the parity hooks (default on) */
/* per-kernel-family HIP-event timing on the device's stream. enable=1 resets the counters. Families (rows):
 * 0 det_prepare 1 feat 2 kf_initiate 3 kf_update 4 kf_predict 5 kf_boxes 6 cosine 7 iou 8 ocsort_cost 9 lap.
 * stats row = [summed ms, launches, tasks, algorithmic bytes, flops]. */
int motcpp_profile(int device, int enable);
int motcpp_profile_stats(int device, double* out_rows5, int cap_rows);
/* counters since creation: [0] frames stepped, [1] flushes, [2] kernel launches */
int motcpp_batch_counters(motcpp_batch* b, long* out3);
/* host wall time since creation, ms: [0] begin() of all trackers, [1] flush (upload+launch+download+sync), [2] advance(), [3] the sync wait inside flush */
int motcpp_batch_host_ms(motcpp_batch* b, double* out4);
int motcpp_batch_tracker_count(motcpp_batch* b);
motcpp_tracker* motcpp_batch_tracker(motcpp_batch* b, int s); /* borrowed handle (for the parity hooks) */

#ifdef __cplusplus
}
#endif
#endif
