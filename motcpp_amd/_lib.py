"""ctypes loader for the gfx950 kernel library (C ABI: include/motcpp_amd.h).

There is no CPU fallback: if the library is missing it must be built (motcpp_amd.build()), and on a box
without a gfx950 device creating a context raises."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("MOTCPP_LIB_DIR") or os.path.join(HERE, "lib")  # (MOTCPP_LIB_DIR: a diagnostic build, e.g. the host library under AddressSanitizer)
HIP_LIB = os.path.join(LIBDIR, "libmotcpp_hip.so")
# diagnostics only (tools/): a differently instrumented build of the SAME sources, e.g. lib/libmotcpp_hip_fineprof.so (-DMOT_LAP_FINE_PROF)
if os.environ.get("MOTCPP_HIP_LIB_DIAG"):
    HIP_LIB = os.path.join(LIBDIR, os.path.basename(os.environ["MOTCPP_HIP_LIB_DIAG"]))
HOST_LIB = os.path.join(LIBDIR, "libmotcpp.so")

KF_XYSR, KF_XYAH, KF_XYWH = 0, 1, 2
COST_IOU, COST_IOU_DIST, COST_IOU_DIST_FUSE, COST_NEG_IOU, COST_BOTSORT = range(5)
LAP_PLAIN, LAP_GATE_MIN, LAP_OCSORT = 0, 1, 2


class MotError(RuntimeError):
    pass


def build(jobs=8):
    """Compile every HIP kernel for gfx950 and the C++ host library, in-tree (motcpp_amd/lib)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-j", str(jobs)])
    return HIP_LIB, HOST_LIB


_hip = None


def hip():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise MotError(f"{HIP_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        L = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        L.mot_version.restype = C.c_char_p
        L.mot_ctx_last_error.restype = C.c_char_p
        L.mot_ctx_last_error.argtypes = [C.c_void_p]
        L.mot_lap_work_bytes.restype = C.c_size_t
        _hip = L
    return _hip


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One mot_ctx (device + HIP stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = hip()
        self.h = C.c_void_p()
        rc = self.lib.mot_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self.h))
        if rc != 0:
            raise MotError(f"mot_ctx_create failed ({rc}): a gfx950 (MI355X) device is required; no CPU fallback")

    def close(self):
        if self.h:
            self.lib.mot_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MotError(f"motcpp_amd call failed ({rc}): {self.lib.mot_ctx_last_error(self.h).decode()}")

    # ---- synchronous host-pointer conveniences (row-major numpy in/out) ----
    def iou_cost(self, a, b, mode=COST_IOU, conf=None):
        a, b = f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        c = f32(conf) if conf is not None else None
        self._chk(self.lib.mot_iou_cost_host(self.h, _p(a), a.shape[0], _p(b), b.shape[0], _p(c) if c is not None else None,
                                             int(mode), _p(out)))
        return out

    def assoc_cost(self, a, b, assoc, frame=(1920, 1080), mode=COST_IOU, conf=None):
        """AssociationFunction(w, h, name)(a, b): assoc 0 iou, 1 hmiou, 2 giou, 3 ciou, 4 diou, 5 centroid."""
        a, b = f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        c = f32(conf) if conf is not None else None
        self._chk(self.lib.mot_assoc_cost_host(self.h, _p(a), a.shape[0], _p(b), b.shape[0], _p(c) if c is not None else None,
                                               int(mode), int(assoc), int(frame[0]), int(frame[1]), _p(out)))
        return out

    def cosine_cost(self, a, b):
        a, b = f32(a), f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self._chk(self.lib.mot_cosine_cost_host(self.h, _p(a), a.shape[0], _p(b), b.shape[0], a.shape[1], _p(out)))
        return out

    def ocsort_cost(self, dets5, trks4, vel2, prev5, vdc):
        dets5, trks4, vel2, prev5 = f32(dets5), f32(trks4), f32(vel2), f32(prev5)
        nd, nt = dets5.shape[0], trks4.shape[0]
        cost, iou = np.zeros((nd, nt), np.float32), np.zeros((nd, nt), np.float32)
        self._chk(self.lib.mot_ocsort_cost_host(self.h, _p(dets5), nd, _p(trks4), nt, _p(vel2), _p(prev5), C.c_float(vdc),
                                                _p(cost), _p(iou)))
        return cost, iou

    def lap(self, cost, thresh, mode=LAP_PLAIN, iou=None, gate=0.0):
        cost = f32(cost)
        n, m = cost.shape
        x, y = np.full(max(n, 1), -1, np.int32), np.full(max(m, 1), -1, np.int32)
        info = C.c_int(0)
        i = f32(iou) if iou is not None else None
        self._chk(self.lib.mot_lap_solve_host(self.h, _p(cost), n, m, C.c_float(thresh), int(mode),
                                              _p(i) if i is not None else None, C.c_float(gate), _p(x), _p(y), C.byref(info)))
        return x[:n], y[:m], info.value

    def embedding_cost(self, metric, a, b):
        """metric 0 cosine distance, 1 raw dot product, 2 euclidean distance (mot_embedding_cost_host)"""
        a, b = f32(a), f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.mot_embedding_cost_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self._chk(self.lib.mot_embedding_cost_host(self.h, int(metric), _p(a), a.shape[0], _p(b), b.shape[0], a.shape[1], _p(out)))
        return out

    def cosine_cost_gated(self, a, b, a_xyxy, b_xyxy, prox, out, cost_mode=COST_BOTSORT):
        """mot_cosine_cost_gated_host: cosine distances of the pairs that pass BoT-SORT's proximity test written into `out` (other entries stay)"""
        a, b, a_xyxy, b_xyxy = f32(a), f32(b), f32(a_xyxy).reshape(-1, 4), f32(b_xyxy).reshape(-1, 4)
        out = np.ascontiguousarray(out, np.float32).copy()
        self.lib.mot_cosine_cost_gated_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                        C.c_int, C.c_float, C.c_void_p]
        self._chk(self.lib.mot_cosine_cost_gated_host(self.h, _p(a), a.shape[0], _p(b), b.shape[0], a.shape[1], _p(a_xyxy), _p(b_xyxy),
                                                      int(cost_mode), C.c_float(prox), _p(out)))
        return out

    def gate_cost(self, kind, mode, mean, cov, meas, cost=None, only_position=False, metric=0, lam=0.98, gated_cost=1e5):
        """mot_gate_cost_host: kind KF_XYAH / KF_XYWH; mode 0 gating distances, 1 utils::fuse_motion, 2 StrongSORT's gate_cost_matrix"""
        mean, cov, meas = f32(mean).reshape(-1, 8), f32(cov).reshape(-1, 64), f32(meas).reshape(-1, 4)
        n, m = mean.shape[0], meas.shape[0]
        out = np.zeros((n, m), np.float32)
        cost = f32(cost) if cost is not None else None
        self.lib.mot_gate_cost_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        self._chk(self.lib.mot_gate_cost_host(self.h, int(kind), int(mode), n, m, _p(mean), _p(cov), _p(meas),
                                              _p(cost) if cost is not None else None, int(only_position), int(metric),
                                              C.c_float(lam), C.c_float(gated_cost), _p(out)))
        return out

    def fuse_iou(self, reid_cost, a, b):
        """utils::fuse_iou (mot_fuse_iou_host)"""
        reid_cost, a, b = f32(reid_cost), f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        out = np.zeros_like(reid_cost)
        self.lib.mot_fuse_iou_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.lib.mot_fuse_iou_host(self.h, _p(reid_cost), _p(a), a.shape[0], _p(b), b.shape[0], _p(out)))
        return out

    def feat_update(self, mode, feat, src, alpha=0.9, alpha_i=None):
        """mot_feat_update on host rows: mode 0 set+normalise, 1 EMA+normalise, 2 ReID normalise (norm > 1e-6), 3 EMA then normalise
        where the norm exceeds 1e-6 (DeepOC-SORT's update_emb); alpha_i: optional per-row EMA weights. Returns new feat."""
        feat, src = f32(feat).copy(), f32(src)
        n, d = src.shape
        ai = f32(alpha_i) if alpha_i is not None else None
        self.lib.mot_feat_update_host_alpha.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._chk(self.lib.mot_feat_update_host_alpha(self.h, int(mode), C.c_float(alpha), _p(ai) if ai is not None else None, n, d, _p(feat), _p(src)))
        return feat

    def lap_behind_stats(self, reset=False):
        """Counters of the problems the exact emulation solved behind the fast path since the last reset (see mot_lap_behind_stats):
        sums over the problems and the slowest problem's own figures."""
        o = np.zeros(80, np.int64)
        self._chk(self.lib.mot_lap_behind_stats(self.h, _p(o), 1 if reset else 0))
        names = ("cyc_phase1_columns", "cyc_phase1_transfer", "cyc_row_reduction", "cyc_augmentation", "n_unique_rows", "serial_row_rounds",
                 "serial_augmentations", "n_extended", "scan_steps", "scan_step_members", "scan_step_real_rows", "tie_events", "single_sweeps",
                 "steps_refused", "find_dense_calls", "row_lists", "cyc_step_classify", "cyc_step_dry", "cyc_step_apply", "cyc_event_sort",
                 "cyc_event_replay", "cyc_find_dense", "cyc_single_sweeps", "cyc_search_setup")
        r = {"problems": int(o[39]), "sum": dict(zip(names, (int(v) for v in o[:24]))),
             "slowest": dict(zip(names, (int(v) for v in o[40:64]))), "slowest_cycles": int(o[79])}
        if os.environ.get("MOT_BEHIND_RAW"):  # (tools/ns_behind_fine.py: the twelve counters only a -DMOT_LAP_FINE_PROF build fills)
            r["raw_fine"] = [int(v) for v in o[24:36]]
        return r

    def lap_fast_stats(self, reset=False):
        """Outcome counts of the assignment fast path on this device since the last reset (see mot_lap_fast_stats)."""
        o = np.zeros(32, np.uint64)
        self._chk(self.lib.mot_lap_fast_stats(self.h, _p(o), 1 if reset else 0))
        d = dict(zip(("fast", "declined_enum", "search_too_large", "certificate_arith", "too_many_tight", "not_unique", "not_attempted", "empty",
                         "cycles_enumerate", "cycles_init", "cycles_search", "cycles_certificate", "searches", "column_scans",
                         "declined_column_list_full", "declined_pair_list_full", "cycles_enum_bucket_rows", "cycles_enum_candidates",
                         "cycles_enum_list", "lane0_candidates", "lane0_pairs_evaluated", "declined_nonfinite", "declined_viable_disjoint",
                         "declined_threshold_tie", "cycles_search_fetch", "cycles_search_deliver", "cycles_search_pick", "cycles_search_augment",
                         "search_retries", "cycles_kernel", "wall_ticks_kernel_100MHz", "cycles_prologue"),
                        (int(v) for v in o[:32])))
        d["short_searches"] = d["search_retries"] >> 32  # searches finished by one lane each (lap_sparse.hpp::sparse_short_searches; the wide kernel counts them)
        d["search_retries"] &= 0xffffffff
        return d

    def lap_geom(self, a, b, thresh, cost_mode=COST_IOU_DIST, conf=None, lap_mode=LAP_PLAIN, gate=0.0, prof=False):
        """Assignment straight from boxes (on-the-fly IoU-family cost inside the solver; no matrix in memory)."""
        a, b = f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        n, m = a.shape[0], b.shape[0]
        x, y = np.full(max(n, 1), -1, np.int32), np.full(max(m, 1), -1, np.int32)
        xv = np.zeros(max(n, 1), np.float32)
        info = C.c_int(0)
        self._prof = np.zeros(36, np.int64)
        c = f32(conf) if conf is not None else None
        self._chk(self.lib.mot_lap_geom_host(self.h, _p(a), n, _p(b), m, _p(c) if c is not None else None, int(cost_mode),
                                             C.c_float(thresh), int(lap_mode), C.c_float(gate), _p(x), _p(y), _p(xv), C.byref(info),
                                             _p(self._prof) if prof else None))
        return x[:n], y[:m], xv[:n], info.value

    def kf_apply(self, kind, op, mean, cov, meas=None, q=None, flags=None, want_boxes=False):
        """op: 0 initiate, 1 predict, 2 update. mean [n,d], cov [n,d,d]; returns (mean, cov[, boxes])."""
        d = 7 if kind == KF_XYSR else 8
        if op == 0:
            meas = f32(meas).reshape(-1, 4)
            n = meas.shape[0]
            mean, cov = np.zeros((n, d), np.float32), np.zeros((n, d, d), np.float32)
        else:
            mean, cov = f32(mean).copy(), f32(cov).copy()
            n = mean.shape[0]
        z = f32(meas).reshape(-1, 4) if meas is not None else None
        qq = f32(q) if q is not None else None
        fl = np.ascontiguousarray(flags, np.uint8) if flags is not None else None
        boxes = np.zeros((n, 4), np.float32) if want_boxes else None
        self._chk(self.lib.mot_kf_apply_host(self.h, int(kind), int(op), n, _p(z) if z is not None else None,
                                             _p(qq) if qq is not None else None, _p(fl) if fl is not None else None,
                                             _p(mean), _p(cov), _p(boxes) if boxes is not None else None))
        return (mean, cov, boxes) if want_boxes else (mean, cov)

    def kf_update_conf(self, mean, cov, meas, conf):
        """XYAH Kalman update with per-measurement confidences (NSA Kalman, mot_kf_update_conf_host)"""
        mean, cov, meas, conf = f32(mean).copy(), f32(cov).copy(), f32(meas).reshape(-1, 4), f32(conf)
        self.lib.mot_kf_update_conf_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.mot_kf_update_conf_host(self.h, KF_XYAH, mean.shape[0], _p(meas), _p(conf), _p(mean), _p(cov)))
        return mean, cov

    def kf_update_blocks(self, mean, blocks, meas, flags=None):
        """XYAH update on block-form covariances (mot_kf_update_blocks_host): returns (mean [n,8], blocks [n,4,4], dense_flag [n], cov_dense [n,8,8])"""
        mean, blocks, meas = f32(mean).copy(), f32(blocks).reshape(-1, 16).copy(), f32(meas).reshape(-1, 4)
        n = mean.shape[0]
        fl = np.ascontiguousarray(flags, np.uint8) if flags is not None else None
        dense = np.zeros(n, np.uint8)
        cov = np.zeros((n, 64), np.float32)
        self.lib.mot_kf_update_blocks_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.mot_kf_update_blocks_host(self.h, n, _p(meas), _p(fl) if fl is not None else None, _p(mean), _p(blocks), _p(dense), _p(cov)))
        return mean, blocks.reshape(n, 4, 4), dense, cov.reshape(n, 8, 8)

    def kf_warp(self, kind, mean, cov, warp9, predict_first=False, q=None, want_boxes=False):
        """mot_kf_warp on AoS states (predict_first: one predict launch with the warp applied after it)."""
        mean, cov = f32(mean).copy(), f32(cov).copy()
        n = mean.shape[0]
        w = f32(warp9).reshape(9)
        qq = f32(q) if q is not None else None
        boxes = np.zeros((n, 4), np.float32) if want_boxes else None
        self.lib.mot_kf_warp_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        self._chk(self.lib.mot_kf_warp_host(self.h, int(kind), n, _p(w), 1 if predict_first else 0, _p(qq) if qq is not None else None,
                                            _p(mean), _p(cov), _p(boxes) if boxes is not None else None))
        return (mean, cov, boxes) if want_boxes else (mean, cov)


# ---- tracker handles (libmotcpp.so: C++17 host library over the C ABI) ------------------------------------
SORT, BYTETRACK, OCSORT, BOTSORT = 0, 1, 2, 3
KIND = {"sort": SORT, "bytetrack": BYTETRACK, "ocsort": OCSORT, "botsort": BOTSORT, "deepocsort": 4, "strongsort": 5, "ucmc": 6, "boosttrack": 7, "hybridsort": 8}
_host = None


def host():
    global _host
    if _host is None:
        hip()
        if not os.path.exists(HOST_LIB):
            raise MotError(f"{HOST_LIB} is missing: run __graft_entry__.build() (there is no CPU fallback)")
        H = C.CDLL(HOST_LIB)
        H.motcpp_last_error.restype = C.c_char_p
        H.motcpp_tracker_create.restype = C.c_void_p
        H.motcpp_tracker_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        H.motcpp_tracker_create_pooled.restype = C.c_void_p
        H.motcpp_tracker_create_pooled.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        H.motcpp_tracker_pool_level.argtypes = [C.c_void_p]
        H.motcpp_pool_stats.argtypes = [C.c_void_p, C.c_int]
        H.motcpp_bench_threads.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p]
        H.motcpp_bench_threads_ex.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        H.motcpp_tracker_destroy.argtypes = [C.c_void_p]
        H.motcpp_tracker_reset.argtypes = [C.c_void_p]
        H.motcpp_tracker_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        H.motcpp_tracker_lap_count.argtypes = [C.c_void_p]
        H.motcpp_tracker_lap_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
        H.motcpp_tracker_dump_states.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        H.motcpp_batch_create.restype = C.c_void_p
        H.motcpp_batch_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        H.motcpp_batch_create_private.restype = C.c_void_p
        H.motcpp_batch_create_private.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        H.motcpp_batch_profile.argtypes = [C.c_void_p, C.c_int]
        H.motcpp_batch_profile_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        H.motcpp_batch_destroy.argtypes = [C.c_void_p]
        H.motcpp_batch_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        H.motcpp_batch_step_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_int]
        H.motcpp_batch_set_threads.argtypes = [C.c_void_p, C.c_int]
        H.motcpp_batch_pin_threads.argtypes = [C.c_void_p, C.c_int]
        H.motcpp_batch_record_laps.argtypes = [C.c_void_p, C.c_int]
        H.motcpp_profile.argtypes = [C.c_int, C.c_int]
        H.motcpp_profile_stats.argtypes = [C.c_int, C.c_void_p, C.c_int]
        H.motcpp_batch_counters.argtypes = [C.c_void_p, C.c_void_p]
        H.motcpp_batch_host_ms.argtypes = [C.c_void_p, C.c_void_p]
        H.motcpp_batch_tracker.restype = C.c_void_p
        H.motcpp_batch_tracker.argtypes = [C.c_void_p, C.c_int]
        _host = H
    return _host


def _err():
    return host().motcpp_last_error().decode()


class _Hooks:
    """parity hooks shared by Tracker and the borrowed per-stream handles of a Batch"""

    def laps(self):
        H, out, cap = host(), [], 1 << 15
        x, y = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        for k in range(H.motcpp_tracker_lap_count(self.h)):
            n, m = C.c_int(), C.c_int()
            if H.motcpp_tracker_lap_get(self.h, k, C.byref(n), C.byref(m), _p(x), _p(y), cap) != 0:
                raise MotError("lap_get failed")
            out.append((x[:n.value].copy(), y[:m.value].copy()))
        return out

    def dump_states(self):
        H, w = host(), C.c_int()
        buf = np.zeros(1 << 20, np.float32)
        while True:
            r = H.motcpp_tracker_dump_states(self.h, _p(buf), buf.size, C.byref(w))
            if r >= 0:
                return buf[:r * w.value].reshape(r, w.value).copy() if r else np.zeros((0, 0), np.float32)
            if r > -1000000:
                raise MotError(_err())
            buf = np.zeros((-r - 1000000 + 8) * max(w.value, 1), np.float32)

    def dump_f64(self):
        """UCMCTrack: [rows, 26] float64 — id, state, death_count, birth_count, det_idx, age, x(4), P(16) per track in list order"""
        H = host()
        H.motcpp_tracker_dump_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        buf = np.zeros((4096, 26), np.float64)
        r = H.motcpp_tracker_dump_f64(self.h, _p(buf), buf.shape[0])
        if r < 0:
            raise MotError(_err())
        return buf[:r].copy()

    def dump_features(self):
        """BoT-SORT: smooth features of the live tracks in dump_states order, [rows, dim] (dim 0: no features)."""
        H, d = host(), C.c_int()
        H.motcpp_tracker_dump_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        buf = np.zeros(1 << 22, np.float32)
        r = H.motcpp_tracker_dump_features(self.h, _p(buf), buf.size, C.byref(d))
        if r < 0:
            raise MotError(_err())
        return buf[:r * d.value].reshape(r, d.value).copy() if d.value else np.zeros((r, 0), np.float32)


class Tracker(_Hooks):
    """One tracker instance on one GPU. update() mirrors motcpp::BaseTracker::update (row-major numpy in/out)."""

    def __init__(self, kind, params=None, device=0, pooled=False, camera=None):
        """pooled: the tracker is a stream of a shared device-lifecycle batch (what the C++ classes are by default): handles updated
        from different host threads at the same time run as one launch sequence. The ctypes call releases the GIL."""
        kind = KIND.get(kind, kind)
        p = f32(params if params is not None else [])
        create = host().motcpp_tracker_create_pooled if pooled else host().motcpp_tracker_create
        if camera is not None:  # UCMCTrack with a calibrated camera: (Ki 3 x 4, Ko 4 x 4)
            assert kind == KIND["ucmc"] and not pooled
            ki, ko = (np.ascontiguousarray(a, np.float64).reshape(-1) for a in camera)
            assert ki.size == 12 and ko.size == 16
            host().motcpp_ucmc_create.restype = C.c_void_p
            host().motcpp_ucmc_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
            self.h = host().motcpp_ucmc_create(_p(p) if p.size else None, int(p.size), _p(ki), _p(ko), int(device))
        else:
            self.h = create(int(kind), _p(p) if p.size else None, int(p.size), int(device))
        if not self.h:
            raise MotError("tracker create failed: " + _err())
        self._out = np.zeros((8192, 8), np.float32)

    def pool_level(self):
        return host().motcpp_tracker_pool_level(self.h)

    def close(self):
        if getattr(self, "h", None):
            host().motcpp_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        host().motcpp_tracker_reset(self.h)

    def set_camera_motion(self, warp2x3):
        """BoT-SORT: 2x3 warp of the next update (BotSTrack::multi_gmc, botsort.cpp:60-91,317-324); None withdraws it."""
        H = host()
        H.motcpp_tracker_set_camera_motion.argtypes = [C.c_void_p, C.c_void_p]
        w = f32(warp2x3).reshape(6) if warp2x3 is not None else None
        if H.motcpp_tracker_set_camera_motion(self.h, _p(w) if w is not None else None) != 0:
            raise MotError(H.motcpp_last_error().decode())

    def update(self, dets, embs=None):
        dets = f32(dets).reshape(-1, 6)
        e, d = None, 0
        if embs is not None and np.size(embs):
            embs = f32(embs)
            e, d = _p(embs), embs.shape[1]
        while True:
            r = host().motcpp_tracker_update(self.h, _p(dets), dets.shape[0], e, d, _p(self._out), self._out.shape[0])
            if r >= 0:
                return self._out[:r].copy()
            if r > -1000000:
                raise MotError(_err())
            self._out = np.zeros((-r - 1000000 + 64, 8), np.float32)


def pool_stats(reset=False):
    """combiner counters of this process: launch sequences run, stream-frames carried, streams moved up a level, largest round"""
    a = (C.c_long * 8)()
    host().motcpp_pool_stats(a, 1 if reset else 0)
    return {"rounds": a[0], "frames": a[1], "moves": a[2], "max_round": a[3], "us_window": a[4], "us_gather": a[5], "us_run": a[6],
            "us_enqueue": a[7]}


def bench_threads(kind, dets, counts, warm, params=None, device=0, frames=None):
    """T tracker objects of the C++ classes on T host threads, each calling BaseTracker::update on host detections
    (motcpp_bench_threads_ex). dets [T, F, N, 6], counts [T, F]; frames (default F): updates per object, the F given frames played back
    and forth when it is larger. Returns a dict (seconds of the timed part, frames, rows, latencies incl. p50 / p99 / p99.9 over all timed
    calls) and the per-tracker id checksums."""
    kind = KIND.get(kind, kind)
    dets, counts = f32(dets), np.ascontiguousarray(counts, np.int32)
    T, F, N = dets.shape[0], dets.shape[1], dets.shape[2]
    p = f32(params if params is not None else [])
    out = np.zeros(5, np.float64)
    pct = np.zeros(3, np.float64)
    cs = np.zeros(T, np.float64)
    rc = host().motcpp_bench_threads_ex(int(kind), _p(p) if p.size else None, int(p.size), T, int(frames or F), int(warm), _p(dets), _p(counts), N,
                                        int(device), _p(out), _p(cs), F, _p(pct))
    if rc != 0:
        raise MotError("motcpp_bench_threads failed")
    return {"seconds": out[0], "frames": int(out[1]), "rows": int(out[2]), "latency_ms_mean": out[3], "latency_ms_max": out[4],
            "latency_ms_p50": pct[0], "latency_ms_p99": pct[1], "latency_ms_p999": pct[2],
            "frames_per_s": out[1] / out[0] if out[0] > 0 else 0.0}, cs


class _Borrowed(_Hooks):
    def __init__(self, h):
        self.h = h


class Batch:
    """S independent streams stepped in lockstep on one GPU (motcpp::StreamBatch)."""

    def __init__(self, kind, nstreams, params=None, device=0, threads=1, record_laps=True, private_device=False):
        kind = KIND.get(kind, kind)
        p = f32(params if params is not None else [])
        self.S = int(nstreams)
        create = host().motcpp_batch_create_private if private_device else host().motcpp_batch_create
        self.h = create(int(kind), _p(p) if p.size else None, int(p.size), self.S, int(device))
        if not self.h:
            raise MotError("batch create failed: " + _err())
        host().motcpp_batch_set_threads(self.h, int(threads))
        host().motcpp_batch_record_laps(self.h, 1 if record_laps else 0)
        self._out = None
        self._cnt = np.zeros(self.S, np.int32)

    def pin_threads(self, first_cpu):
        """Pin the calling thread's worker team to consecutive allowed CPUs (call from the thread that steps the batch)."""
        host().motcpp_batch_pin_threads(self.h, int(first_cpu))

    def close(self):
        if getattr(self, "h", None):
            host().motcpp_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def profile(self, enable):
        host().motcpp_batch_profile(self.h, 1 if enable else 0)

    def profile_stats(self):
        a = np.zeros((len(FAMILIES), 5), np.float64)
        host().motcpp_batch_profile_stats(self.h, _p(a), len(FAMILIES))
        return {FAMILIES[i]: {"ms": a[i, 0], "launches": int(a[i, 1]), "tasks": int(a[i, 2]), "bytes": a[i, 3], "flops": a[i, 4]}
                for i in range(len(FAMILIES))}

    def stream(self, s):
        return _Borrowed(host().motcpp_batch_tracker(self.h, int(s)))

    def counters(self):
        a = (C.c_long * 3)()
        host().motcpp_batch_counters(self.h, a)
        h = (C.c_double * 4)()
        host().motcpp_batch_host_ms(self.h, h)
        return {"frames": a[0], "flushes": a[1], "launches": a[2], "ms_begin": h[0], "ms_flush": h[1], "ms_advance": h[2],
                "ms_sync_wait": h[3]}

    def step(self, dets, counts=None, embs=None, cap=None, resident_ptr=None, resident_embs=None):
        """dets [S, N, 6] (counts[s] valid rows each); returns (out [S, cap, 8], out_counts [S]).
        resident_ptr: device address of the same detections as SoA [S, 6, N] already in HBM (no payload upload);
        resident_embs: (device address of [S, N, D] row-major embeddings, D) — then `embs` is not read."""
        dets = f32(dets)
        S, N = dets.shape[0], dets.shape[1]
        assert S == self.S
        counts = np.full(S, N, np.int32) if counts is None else np.ascontiguousarray(counts, np.int32)
        cap = cap or max(2 * N, 64)
        if self._out is None or self._out.shape[1] < cap:
            self._out = np.zeros((S, cap, 8), np.float32)
        e, d = None, 0
        if embs is not None:
            embs = f32(embs)
            e, d = _p(embs), embs.shape[2]
        if resident_ptr and resident_embs:
            H = host()
            H.motcpp_batch_step_resident_embs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                          C.c_void_p, C.c_void_p, C.c_int]
            r = H.motcpp_batch_step_resident_embs(self.h, _p(dets), _p(counts), N, C.c_void_p(int(resident_ptr)),
                                                  C.c_void_p(int(resident_embs[0])), int(resident_embs[1]), _p(self._out), _p(self._cnt),
                                                  self._out.shape[1])
        elif resident_ptr:
            r = host().motcpp_batch_step_resident(self.h, _p(dets), _p(counts), N, C.c_void_p(int(resident_ptr)), e, d,
                                                  _p(self._out), _p(self._cnt), self._out.shape[1])
        else:
            r = host().motcpp_batch_step(self.h, _p(dets), _p(counts), N, e, d, _p(self._out), _p(self._cnt), self._out.shape[1])
        if r < 0:
            raise MotError("batch step failed: " + _err())
        return self._out, self._cnt


FAMILIES = ["det_prepare", "feat", "kf_initiate", "kf_update", "kf_predict", "kf_boxes", "cosine", "iou_cost", "ocsort_cost", "lap"]


def profile(enable, device=0):
    """Turn per-kernel-family HIP-event timing on (resets the counters) or off."""
    if host().motcpp_profile(int(device), 1 if enable else 0) != 0:
        raise MotError(_err())


def profile_stats(device=0):
    a = np.zeros((len(FAMILIES), 5), np.float64)
    if host().motcpp_profile_stats(int(device), _p(a), len(FAMILIES)) < 0:
        raise MotError(_err())
    return {FAMILIES[i]: {"ms": a[i, 0], "launches": int(a[i, 1]), "tasks": int(a[i, 2]), "bytes": a[i, 3], "flops": a[i, 4]}
            for i in range(len(FAMILIES))}


def pinned_array(ctx, shape, dtype):
    """numpy array over page-locked host memory (mot_host_alloc): device -> host copies into it run at PCIe speed and
    asynchronously; pageable memory goes through a staging buffer at a fraction of that. Freed with the context's process."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = C.c_void_p()
    ctx._chk(ctx.lib.mot_host_alloc(ctx.h, C.c_size_t(max(n, 16)), C.byref(ptr)))
    buf = (C.c_char * max(n, 16)).from_address(ptr.value)
    a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    a[...] = 0
    return a


class DeviceByteTrack:
    """S ByteTrack streams whose whole per-frame lifecycle runs on the GPU (mot_bt_*, csrc/bt_device.hip): the host
    enqueues a fixed launch sequence per frame. params = [min_conf, track_thresh, match_thresh, track_buffer, frame_rate]."""

    def __init__(self, nstreams, cap_tracks, max_dets, params=None, device=0):
        self.ctx = Context(device)
        self.lib = self.ctx.lib
        self.S, self.CAP, self.D = int(nstreams), int(cap_tracks), int(max_dets)
        p = f32(params if params is not None else [0.1, 0.45, 0.8, 25, 30])
        self.h = C.c_void_p()
        self.ctx._chk(self.lib.mot_bt_create(self.ctx.h, self.S, self.CAP, self.D, _p(p), C.byref(self.h)))
        self._ddets = C.c_void_p()
        self.ctx._chk(self.lib.mot_malloc(self.ctx.h, C.c_size_t(self.S * 6 * self.D * 4), C.byref(self._ddets)))
        self._soa = np.zeros((self.S, 6, self.D), np.float32)
        self._out = None
        self._cnt = pinned_array(self.ctx, (self.S,), np.int32)
        self.lib.mot_bt_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.mot_bt_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.mot_bt_profile.argtypes = [C.c_void_p, C.c_int]
        self.lib.mot_bt_profile_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.mot_bt_reset.argtypes = [C.c_void_p]
        self.lib.mot_bt_destroy.argtypes = [C.c_void_p]

    def step(self, dets=None, counts=None, cap=None, resident_ptr=None, out=None, out_counts=None):
        """dets [S, N, 6] host rows (uploaded as SoA), or resident_ptr = device SoA [S][6][max_dets] + counts.
        out [S, cap, 8] / out_counts [S] (C-contiguous float32 / int32): write the tables there instead of an internal buffer."""
        if resident_ptr is None:
            dets = f32(dets)
            n = dets.shape[1]
            assert dets.shape[0] == self.S and n <= self.D
            counts = np.full(self.S, n, np.int32) if counts is None else np.ascontiguousarray(counts, np.int32)
            self._soa[:, :, :n] = dets.transpose(0, 2, 1)
            self.ctx._chk(self.lib.mot_memcpy_h2d(self.ctx.h, self._ddets, _p(self._soa), C.c_size_t(self._soa.nbytes)))
            ptr = self._ddets
        else:
            counts = np.ascontiguousarray(counts, np.int32)
            ptr = C.c_void_p(int(resident_ptr))
        if out is not None:
            assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.shape[0] == self.S and out.shape[2] == 8
            assert out_counts is not None and out_counts.dtype == np.int32 and out_counts.flags["C_CONTIGUOUS"]
            self.ctx._chk(self.lib.mot_bt_step(self.h, ptr, _p(counts), _p(out), _p(out_counts), int(out.shape[1])))
            return out, out_counts
        cap = int(cap or max(2 * self.D, 64))
        if self._out is None or self._out.shape[1] != cap:
            self._out = pinned_array(self.ctx, (self.S, cap, 8), np.float32)
        self.ctx._chk(self.lib.mot_bt_step(self.h, ptr, _p(counts), _p(self._out), _p(self._cnt), cap))
        return self._out, self._cnt

    def step_packed(self, resident_ptr, counts, rows, out_counts):
        """One frame, packed output (mot_bt_step_packed): rows [cap, 8] float32 and out_counts [S] int32 are caller buffers
        (page-locked for speed); returns the number of rows written — stream s's rows start at out_counts[:s].sum()."""
        counts = np.ascontiguousarray(counts, np.int32)
        assert rows.dtype == np.float32 and rows.flags["C_CONTIGUOUS"] and rows.shape[1] == 8
        assert out_counts.dtype == np.int32 and out_counts.flags["C_CONTIGUOUS"] and out_counts.shape[0] == self.S
        total = C.c_int(0)
        self.lib.mot_bt_step_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bt_step_packed(self.h, C.c_void_p(int(resident_ptr)), _p(counts), _p(rows), int(rows.shape[0]),
                                                  _p(out_counts), C.byref(total)))
        return total.value

    def enqueue_packed(self, resident_ptr, counts, rows_cap):
        """queue one frame and return at once (mot_bt_enqueue_packed); at most two frames may be pending"""
        counts = np.ascontiguousarray(counts, np.int32)
        self.lib.mot_bt_enqueue_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.ctx._chk(self.lib.mot_bt_enqueue_packed(self.h, C.c_void_p(int(resident_ptr)), _p(counts), int(rows_cap)))

    def collect_packed(self, rows, out_counts):
        """wait for the oldest pending frame and fetch its packed rows (mot_bt_collect_packed); returns the number of rows"""
        total = C.c_int(0)
        self.lib.mot_bt_collect_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bt_collect_packed(self.h, _p(rows) if rows is not None else None, int(rows.shape[0]) if rows is not None else 0,
                                                     _p(out_counts), C.byref(total)))  # rows None: the table stays on the device
        return total.value

    def device_output(self):
        """(rows ptr, offsets ptr, counts ptr): device addresses of the last packed result (mot_bt_device_output)."""
        r, o, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.mot_bt_device_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bt_device_output(self.h, C.byref(r), C.byref(o), C.byref(c)))
        return r.value, o.value, c.value

    def dump(self, s):
        """(ids [n], mean [n,8], cov [n,8,8]) of stream s's live tracks, active list then lost list."""
        cap = 2 * self.CAP
        ids = np.zeros(cap, np.int32)
        mean, cov = np.zeros((cap, 8), np.float32), np.zeros((cap, 64), np.float32)
        n = self.lib.mot_bt_dump(self.h, int(s), _p(ids), _p(mean), _p(cov), cap)
        if n < 0:
            raise MotError("mot_bt_dump failed")
        return ids[:n].copy(), mean[:n].copy(), cov[:n].reshape(-1, 8, 8).copy()

    def reset(self):
        self.ctx._chk(self.lib.mot_bt_reset(self.h))

    def profile(self, on):
        self.ctx._chk(self.lib.mot_bt_profile(self.h, 1 if on else 0))

    def profile_stats(self):
        o = np.zeros(8, np.float64)
        self.ctx._chk(self.lib.mot_bt_profile_stats(self.h, _p(o)))
        return {"lap1_ms": o[0], "lap23_ms": o[1], "frame_ms": o[2], "frames": int(o[3]), "lap1_problems": o[4], "lap1_nm": o[5],
                "lap23_problems": o[6], "lap23_nm": o[7]}

    def profile_lap_sparse(self):
        """HIP-event ms and launches of the first association's sparse-solver kernel alone"""
        o = np.zeros(2, np.float64)
        self.lib.mot_bt_profile_lap_sparse.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bt_profile_lap_sparse(self.h, _p(o)))
        return {"ms": o[0], "launches": int(o[1])}

    def profile_dims(self):
        """summed rows / columns of the queued assignment problems: [first N, first M, second+unconfirmed N, second+unconfirmed M]"""
        o = np.zeros(4, np.float64)
        self.lib.mot_bt_profile_dims.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bt_profile_dims(self.h, _p(o)))
        return o

    def profile_kalman(self):
        """HIP-event ms and items of the frame's Kalman launches: predicted boxes of the pool, initiations, updates"""
        o = np.zeros(6, np.float64)
        self.lib.mot_bt_profile_kalman.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bt_profile_kalman(self.h, _p(o)))
        return {"predict_boxes_ms": o[0], "predict_boxes_items": o[1], "initiate_ms": o[2], "initiate_items": o[3],
                "update_ms": o[4], "update_items": o[5]}

    def close(self):
        if getattr(self, "h", None):
            self.lib.mot_bt_destroy(self.h)
            self.h = None
            self.lib.mot_free(self.ctx.h, self._ddets)
            self.ctx.close()


class DeviceBotSort:
    """S BoT-SORT streams whose whole per-frame lifecycle runs on the GPU (mot_bot_*, csrc/bot_device.hip).
    params = [track_high, track_low, new_track, track_buffer, match_thresh, proximity, appearance, frame_rate, fuse_first, with_reid]."""

    def __init__(self, nstreams, cap_tracks, max_dets, emb_dim=0, params=None, device=0):
        self.ctx = Context(device)
        self.lib = self.ctx.lib
        self.S, self.CAP, self.D, self.E = int(nstreams), int(cap_tracks), int(max_dets), int(emb_dim)
        p = f32(params if params is not None else [0.5, 0.1, 0.6, 30, 0.8, 0.5, 0.25, 30, 0, 1])
        self.h = C.c_void_p()
        self.lib.mot_bot_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        self.ctx._chk(self.lib.mot_bot_create(self.ctx.h, self.S, self.CAP, self.D, self.E, _p(p), C.byref(self.h)))
        self.lib.mot_bot_step_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p]
        self.lib.mot_bot_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.mot_bot_destroy.argtypes = [C.c_void_p]
        self.lib.mot_bot_reset.argtypes = [C.c_void_p]
        self.lib.mot_bot_profile.argtypes = [C.c_void_p, C.c_int]
        self.lib.mot_bot_profile_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._ddets = self._dembs = None
        self._rows = self._cnt = None

    def _dev(self, nbytes):
        p = C.c_void_p()
        self.ctx._chk(self.lib.mot_malloc(self.ctx.h, C.c_size_t(nbytes), C.byref(p)))
        return p

    def step_packed(self, dets_ptr, counts, rows, out_counts, embs_ptr=None, warps=None, has_warp=None):
        """device pointers in (dets SoA [S][6][D], embs [S][D][E] or None), caller buffers out; returns the number of rows"""
        counts = np.ascontiguousarray(counts, np.int32)
        total = C.c_int(0)
        w = f32(warps).reshape(self.S, 6) if warps is not None else None
        hw = np.ascontiguousarray(has_warp, np.uint8) if has_warp is not None else None
        self.ctx._chk(self.lib.mot_bot_step_packed(self.h, C.c_void_p(int(dets_ptr)), _p(counts), C.c_void_p(int(embs_ptr)) if embs_ptr else None,
                                                   _p(w) if w is not None else None, _p(hw) if hw is not None else None, _p(rows),
                                                   int(rows.shape[0]), _p(out_counts), C.byref(total)))
        return total.value

    def enqueue_packed(self, dets_ptr, counts, rows_cap, embs_ptr=None, warps=None, has_warp=None):
        """queue one frame and return at once (mot_bot_enqueue_packed); at most two frames may be pending"""
        counts = np.ascontiguousarray(counts, np.int32)
        w = f32(warps).reshape(self.S, 6) if warps is not None else None
        hw = np.ascontiguousarray(has_warp, np.uint8) if has_warp is not None else None
        self.lib.mot_bot_enqueue_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.ctx._chk(self.lib.mot_bot_enqueue_packed(self.h, C.c_void_p(int(dets_ptr)), _p(counts), C.c_void_p(int(embs_ptr)) if embs_ptr else None,
                                                      _p(w) if w is not None else None, _p(hw) if hw is not None else None, int(rows_cap)))

    def collect_packed(self, rows, out_counts):
        """wait for the oldest pending frame and fetch its packed rows (mot_bot_collect_packed); returns the number of rows"""
        total = C.c_int(0)
        self.lib.mot_bot_collect_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bot_collect_packed(self.h, _p(rows) if rows is not None else None, int(rows.shape[0]) if rows is not None else 0,
                                                     _p(out_counts), C.byref(total)))  # rows None: the table stays on the device
        return total.value

    def step(self, dets, counts, embs=None, warps=None, has_warp=None):
        """host convenience: dets [S, N, 6] rows, embs [S, N, E] or None -> list of per-stream tables"""
        dets = f32(dets)
        n = dets.shape[1]
        assert dets.shape[0] == self.S and n <= self.D
        if self._ddets is None:
            self._ddets = self._dev(self.S * 6 * self.D * 4)
            self._rows = pinned_array(self.ctx, (self.S * self.CAP, 8), np.float32)
            self._cnt = pinned_array(self.ctx, (self.S,), np.int32)
        soa = np.zeros((self.S, 6, self.D), np.float32)
        soa[:, :, :n] = dets.transpose(0, 2, 1)
        self.ctx._chk(self.lib.mot_memcpy_h2d(self.ctx.h, self._ddets, _p(soa), C.c_size_t(soa.nbytes)))
        eptr = None
        if embs is not None and self.E:
            if self._dembs is None:
                self._dembs = self._dev(self.S * self.D * self.E * 4)
            e = np.zeros((self.S, self.D, self.E), np.float32)
            e[:, :n] = f32(embs)
            self.ctx._chk(self.lib.mot_memcpy_h2d(self.ctx.h, self._dembs, _p(e), C.c_size_t(e.nbytes)))
            eptr = self._dembs.value
        self.ctx._chk(self.lib.mot_ctx_sync(self.ctx.h))
        total = self.step_packed(self._ddets.value, counts, self._rows, self._cnt, eptr, warps, has_warp)
        off = np.concatenate([[0], np.cumsum(self._cnt)])
        assert off[-1] == total
        return [self._rows[off[s]:off[s + 1]].copy() for s in range(self.S)]

    def dump(self, s):
        """(ids [n], mean [n,8], cov [n,8,8], feats [n,E], has_feat [n]) of stream s's live tracks, tracked list then lost list"""
        cap = 2 * self.CAP
        ids = np.zeros(cap, np.int32)
        mean, cov = np.zeros((cap, 8), np.float32), np.zeros((cap, 64), np.float32)
        feats = np.zeros((cap, max(self.E, 1)), np.float32)
        has = np.zeros(cap, np.uint8)
        n = self.lib.mot_bot_dump(self.h, int(s), _p(ids), _p(mean), _p(cov), _p(feats) if self.E else None, _p(has), cap)
        if n < 0:
            raise MotError("mot_bot_dump failed")
        return ids[:n].copy(), mean[:n].copy(), cov[:n].reshape(-1, 8, 8).copy(), feats[:n, :self.E].copy(), has[:n].copy()

    def device_output(self):
        """(rows ptr, offsets ptr, counts ptr): device addresses of the last packed result (mot_bot_device_output)."""
        r, o, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.mot_bot_device_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bot_device_output(self.h, C.byref(r), C.byref(o), C.byref(c)))
        return r.value, o.value, c.value

    def profile(self, on):
        self.ctx._chk(self.lib.mot_bot_profile(self.h, 1 if on else 0))

    def profile_stats(self):
        o = np.zeros(8, np.float64)
        self.ctx._chk(self.lib.mot_bot_profile_stats(self.h, _p(o)))
        f = np.zeros(2, np.float64)
        self.lib.mot_bot_profile_feat.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_bot_profile_feat(self.h, _p(f)))
        return {"lap_ms": o[0], "cos_ms": o[1], "frame_ms": o[2], "frames": int(o[3]), "lap_problems": o[4], "lap_nm": o[5], "cos_nm": o[6], "emb_dim": int(o[7]),
                "feat_ms": f[0], "feat_row_moves": f[1]}

    def reset(self):
        self.ctx._chk(self.lib.mot_bot_reset(self.h))

    def close(self):
        if self.h:
            self.lib.mot_bot_destroy(self.h)
            self.h = None


class DeviceOCSort:
    """S OC-SORT streams whose whole per-frame lifecycle runs on the GPU (mot_oc_*, csrc/oc_device.hip).
    params = [det_thresh, max_age, max_obs, min_hits, iou_threshold, min_conf, delta_t, inertia, use_byte, Q_xy, Q_s, asso, w, h]."""

    def __init__(self, nstreams, cap_tracks, max_dets, params=None, device=0):
        self.ctx = Context(device)
        self.lib = self.ctx.lib
        self.S, self.CAP, self.D = int(nstreams), int(cap_tracks), int(max_dets)
        p = f32(params if params is not None else [0.2, 30, 50, 3, 0.3, 0.1, 3, 0.2, 0, 0.01, 0.0001, 0, 1920, 1080])
        self.h = C.c_void_p()
        self.lib.mot_oc_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        self.ctx._chk(self.lib.mot_oc_create(self.ctx.h, self.S, self.CAP, self.D, _p(p), C.byref(self.h)))
        self.lib.mot_oc_step_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.mot_oc_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.mot_oc_destroy.argtypes = [C.c_void_p]
        self.lib.mot_oc_reset.argtypes = [C.c_void_p]
        self.lib.mot_oc_profile.argtypes = [C.c_void_p, C.c_int]
        self.lib.mot_oc_profile_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._ddets = None

    def step_packed(self, dets_ptr, counts, rows, out_counts):
        counts = np.ascontiguousarray(counts, np.int32)
        total = C.c_int(0)
        self.ctx._chk(self.lib.mot_oc_step_packed(self.h, C.c_void_p(int(dets_ptr)), _p(counts), _p(rows), int(rows.shape[0]), _p(out_counts),
                                                  C.byref(total)))
        return total.value

    def enqueue_packed(self, dets_ptr, counts, rows_cap):
        """queue one frame and return at once (mot_oc_enqueue_packed); at most two frames may be pending"""
        counts = np.ascontiguousarray(counts, np.int32)
        self.lib.mot_oc_enqueue_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.ctx._chk(self.lib.mot_oc_enqueue_packed(self.h, C.c_void_p(int(dets_ptr)), _p(counts), int(rows_cap)))

    def collect_packed(self, rows, out_counts):
        """wait for the oldest pending frame and fetch its packed rows (mot_oc_collect_packed); returns the number of rows"""
        total = C.c_int(0)
        self.lib.mot_oc_collect_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_oc_collect_packed(self.h, _p(rows) if rows is not None else None, int(rows.shape[0]) if rows is not None else 0,
                                                     _p(out_counts), C.byref(total)))  # rows None: the table stays on the device
        return total.value

    def step(self, dets, counts):
        """host convenience: dets [S, N, 6] rows -> list of per-stream tables"""
        dets = f32(dets)
        n = dets.shape[1]
        assert dets.shape[0] == self.S and n <= self.D
        if self._ddets is None:
            self._ddets = C.c_void_p()
            self.ctx._chk(self.lib.mot_malloc(self.ctx.h, C.c_size_t(self.S * 6 * self.D * 4), C.byref(self._ddets)))
            self._rows = pinned_array(self.ctx, (self.S * self.CAP, 8), np.float32)
            self._cnt = pinned_array(self.ctx, (self.S,), np.int32)
        soa = np.zeros((self.S, 6, self.D), np.float32)
        soa[:, :, :n] = dets.transpose(0, 2, 1)
        self.ctx._chk(self.lib.mot_memcpy_h2d(self.ctx.h, self._ddets, _p(soa), C.c_size_t(soa.nbytes)))
        self.ctx._chk(self.lib.mot_ctx_sync(self.ctx.h))
        total = self.step_packed(self._ddets.value, counts, self._rows, self._cnt)
        off = np.concatenate([[0], np.cumsum(self._cnt)])
        assert off[-1] == total
        return [self._rows[off[s]:off[s + 1]].copy() for s in range(self.S)]

    def device_output(self):
        r, o, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.mot_oc_device_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_oc_device_output(self.h, C.byref(r), C.byref(o), C.byref(c)))
        return r.value, o.value, c.value

    def dump(self, s):
        """(ids [n], mean [n,7], cov [n,7,7]) of stream s's trackers in list order"""
        cap = self.CAP
        ids = np.zeros(cap, np.int32)
        mean, cov = np.zeros((cap, 7), np.float32), np.zeros((cap, 49), np.float32)
        n = self.lib.mot_oc_dump(self.h, int(s), _p(ids), _p(mean), _p(cov), cap)
        if n < 0:
            raise MotError("mot_oc_dump failed")
        return ids[:n].copy(), mean[:n].copy(), cov[:n].reshape(-1, 7, 7).copy()

    def profile(self, on):
        self.ctx._chk(self.lib.mot_oc_profile(self.h, 1 if on else 0))

    def profile_stats(self):
        o = np.zeros(8, np.float64)
        self.ctx._chk(self.lib.mot_oc_profile_stats(self.h, _p(o)))
        return {"lap_ms": o[0], "cost_ms": o[1], "frame_ms": o[2], "frames": int(o[3]), "lap_problems": o[4], "lap_nm": o[5]}

    def reset(self):
        self.ctx._chk(self.lib.mot_oc_reset(self.h))

    def close(self):
        if self.h:
            self.lib.mot_oc_destroy(self.h)
            self.h = None


class DeviceSort:
    """S SORT streams with the whole per-frame lifecycle on the GPU (mot_sort_*, csrc/sort_device.hip).
    params = [det_thresh, max_age, max_obs, min_hits, iou_threshold]."""

    def __init__(self, nstreams, cap_tracks, max_dets, params=None, device=0):
        self.ctx = Context(device)
        self.lib = self.ctx.lib
        self.S, self.CAP, self.D = int(nstreams), int(cap_tracks), int(max_dets)
        p = f32(params if params is not None else [0.3, 1, 50, 3, 0.3])
        self.h = C.c_void_p()
        self.ctx._chk(self.lib.mot_sort_create(self.ctx.h, self.S, self.CAP, self.D, _p(p), C.byref(self.h)))
        self._ddets = C.c_void_p()
        self.ctx._chk(self.lib.mot_malloc(self.ctx.h, C.c_size_t(self.S * 6 * self.D * 4), C.byref(self._ddets)))
        self._soa = np.zeros((self.S, 6, self.D), np.float32)
        self._out = None
        self._cnt = pinned_array(self.ctx, (self.S,), np.int32)
        self.lib.mot_sort_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.mot_sort_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.mot_sort_reset.argtypes = [C.c_void_p]
        self.lib.mot_sort_destroy.argtypes = [C.c_void_p]

    def step(self, dets=None, counts=None, cap=None, resident_ptr=None, out=None, out_counts=None):
        if resident_ptr is None:
            dets = f32(dets)
            n = dets.shape[1]
            assert dets.shape[0] == self.S and n <= self.D
            counts = np.full(self.S, n, np.int32) if counts is None else np.ascontiguousarray(counts, np.int32)
            self._soa[:, :, :n] = dets.transpose(0, 2, 1)
            self.ctx._chk(self.lib.mot_memcpy_h2d(self.ctx.h, self._ddets, _p(self._soa), C.c_size_t(self._soa.nbytes)))
            ptr = self._ddets
        else:
            counts = np.ascontiguousarray(counts, np.int32)
            ptr = C.c_void_p(int(resident_ptr))
        if out is not None:
            self.ctx._chk(self.lib.mot_sort_step(self.h, ptr, _p(counts), _p(out), _p(out_counts), int(out.shape[1])))
            return out, out_counts
        cap = int(cap or max(2 * self.D, 64))
        if self._out is None or self._out.shape[1] != cap:
            self._out = pinned_array(self.ctx, (self.S, cap, 8), np.float32)
        self.ctx._chk(self.lib.mot_sort_step(self.h, ptr, _p(counts), _p(self._out), _p(self._cnt), cap))
        return self._out, self._cnt

    def step_packed(self, resident_ptr, counts, rows, out_counts):
        """one frame, packed output (mot_sort_step_packed): the emitted rows of all streams back to back; returns their number"""
        counts = np.ascontiguousarray(counts, np.int32)
        total = C.c_int(0)
        self.lib.mot_sort_step_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_sort_step_packed(self.h, C.c_void_p(int(resident_ptr)), _p(counts), _p(rows), int(rows.shape[0]), _p(out_counts),
                                                    C.byref(total)))
        return total.value

    def enqueue_packed(self, resident_ptr, counts, rows_cap):
        """queue one frame and return at once (mot_sort_enqueue_packed); at most two frames may be pending"""
        counts = np.ascontiguousarray(counts, np.int32)
        self.lib.mot_sort_enqueue_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.ctx._chk(self.lib.mot_sort_enqueue_packed(self.h, C.c_void_p(int(resident_ptr)), _p(counts), int(rows_cap)))

    def collect_packed(self, rows, out_counts):
        """wait for the oldest pending frame and fetch its packed rows (mot_sort_collect_packed); returns the number of rows"""
        total = C.c_int(0)
        self.lib.mot_sort_collect_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_sort_collect_packed(self.h, _p(rows) if rows is not None else None, int(rows.shape[0]) if rows is not None else 0,
                                                     _p(out_counts), C.byref(total)))  # rows None: the table stays on the device
        return total.value

    def device_output(self):
        r, o, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.mot_sort_device_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_sort_device_output(self.h, C.byref(r), C.byref(o), C.byref(c)))
        return r.value, o.value, c.value

    def dump(self, s):
        ids = np.zeros(self.CAP, np.int32)
        mean, cov = np.zeros((self.CAP, 7), np.float32), np.zeros((self.CAP, 49), np.float32)
        n = self.lib.mot_sort_dump(self.h, int(s), _p(ids), _p(mean), _p(cov), self.CAP)
        if n < 0:
            raise MotError("mot_sort_dump failed")
        return ids[:n].copy(), mean[:n].copy(), cov[:n].copy()

    def reset(self):
        self.ctx._chk(self.lib.mot_sort_reset(self.h))

    def profile(self, on):
        self.lib.mot_sort_profile.argtypes = [C.c_void_p, C.c_int]
        self.ctx._chk(self.lib.mot_sort_profile(self.h, 1 if on else 0))

    def profile_stats(self):
        o = np.zeros(8, np.float64)
        self.lib.mot_sort_profile_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._chk(self.lib.mot_sort_profile_stats(self.h, _p(o)))
        return {"lap1_ms": o[0], "lap23_ms": o[1], "frame_ms": o[2], "frames": int(o[3]), "lap1_problems": o[4], "lap1_nm": o[5],
                "lap23_problems": o[6], "lap23_nm": o[7]}

    def close(self):
        if getattr(self, "h", None):
            self.lib.mot_sort_destroy(self.h)
            self.h = None
            self.lib.mot_free(self.ctx.h, self._ddets)
            self.ctx.close()
