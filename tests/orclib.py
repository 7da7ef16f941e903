"""ctypes front-end of the CPU ORACLE (oracle/). Test infrastructure only: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORC_DIR, "build", "liborc.so")

SORT, BYTETRACK, OCSORT, BOTSORT, DEEPOCSORT, STRONGSORT, UCMC, BOOSTTRACK, HYBRIDSORT = 0, 1, 2, 3, 4, 5, 6, 7, 8
KF_XYSR, KF_XYAH, KF_XYWH = 0, 1, 2
KF_DIM = {0: 7, 1: 8, 2: 8}

_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    srcs = [os.path.join(ORC_DIR, f) for f in ("orc_capi.cpp", "orc_math.hpp", "orc_kf.hpp", "orc_trackers.hpp")]
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORC_DIR, "-B"], stdout=subprocess.DEVNULL)
    return LIB


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_tracker_create.restype = C.c_void_p
        L.orc_tracker_create.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.orc_tracker_destroy.argtypes = [C.c_void_p]
        L.orc_tracker_reset.argtypes = [C.c_void_p]
        L.orc_tracker_set_warp.restype = C.c_int
        L.orc_tracker_set_warp.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_tracker_update.restype = C.c_int
        L.orc_tracker_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_tracker_lap_count.restype = C.c_int
        L.orc_tracker_lap_count.argtypes = [C.c_void_p]
        L.orc_tracker_lap_get.restype = C.c_int
        L.orc_tracker_lap_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _i, _i, C.c_int]
        L.orc_tracker_dump_states.restype = C.c_int
        L.orc_tracker_dump_states.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]

    # ---- primitives ----
    def iou_batch(self, a, b):
        a, b = f32(a), f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.orc_iou_batch(a.ctypes, a.shape[0], a.shape[1] if a.ndim == 2 else 4, b.ctypes, b.shape[0],
                               b.shape[1] if b.ndim == 2 else 4, out.ctypes)
        return out

    def asso_batch(self, kind, a, b, frame=(1920, 1080)):
        a, b = f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        if out.size:
            self.lib.orc_asso_batch(int(kind), a.ctypes.data_as(C.c_void_p), a.shape[0], b.ctypes.data_as(C.c_void_p), b.shape[0],
                                    int(frame[0]), int(frame[1]), out.ctypes.data_as(C.c_void_p))
        return out

    def iou_distance(self, a, b):
        a, b = f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.orc_iou_distance(a.ctypes, a.shape[0], b.ctypes, b.shape[0], out.ctypes)
        return out

    def fuse_score(self, cost, conf):
        cost, conf = f32(cost), f32(conf)
        out = np.zeros_like(cost)
        self.lib.orc_fuse_score(cost.ctypes, cost.shape[0], cost.shape[1], conf.ctypes, out.ctypes)
        return out

    def cosine_distance(self, t, d):
        t, d = f32(t), f32(d)
        out = np.zeros((t.shape[0], d.shape[0]), np.float32)
        self.lib.orc_cosine_distance(t.ctypes, t.shape[0], d.ctypes, d.shape[0], t.shape[1], out.ctypes)
        return out

    def embedding_distance(self, metric, t, d):
        """metric 0 cosine, 1 raw dot product, 2 euclidean"""
        t, d = f32(t), f32(d)
        out = np.zeros((t.shape[0], d.shape[0]), np.float32)
        self.lib.orc_embedding_distance.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.orc_embedding_distance(int(metric), t.ctypes, t.shape[0], d.ctypes, d.shape[0], t.shape[1], out.ctypes)
        return out

    def gate_cost(self, kind, mode, mean, cov, meas, cost=None, only_position=False, metric=0, lam=0.98, gated_cost=1e5):
        """kind 1 XYAH / 2 XYWH; mode 0 gating distances, 1 utils::fuse_motion, 2 StrongSORT's gate_cost_matrix"""
        mean, cov, meas = f32(mean).reshape(-1, 8), f32(cov).reshape(-1, 64), f32(meas).reshape(-1, 4)
        n, m = mean.shape[0], meas.shape[0]
        out = np.zeros((n, m), np.float32)
        cost = f32(cost) if cost is not None else np.zeros((n, m), np.float32)
        self.lib.orc_gate_cost.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_float, C.c_float, C.c_void_p]
        self.lib.orc_gate_cost(int(kind), int(mode), n, m, mean.ctypes.data, cov.ctypes.data, meas.ctypes.data, cost.ctypes.data,
                               int(only_position), int(metric), C.c_float(lam), C.c_float(gated_cost), out.ctypes.data)
        return out

    def fuse_iou(self, reid, a, b):
        reid, a, b = f32(reid), f32(a).reshape(-1, 4), f32(b).reshape(-1, 4)
        out = np.zeros_like(reid)
        self.lib.orc_fuse_iou.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.orc_fuse_iou(reid.ctypes.data, a.shape[0], b.shape[0], a.ctypes.data, b.ctypes.data, out.ctypes.data)
        return out

    def linear_assignment(self, cost, thresh):
        cost = f32(cost)
        n, m = cost.shape
        x = np.full(max(n, 1), -1, np.int32)
        y = np.full(max(m, 1), -1, np.int32)
        self.lib.orc_linear_assignment(cost.ctypes, n, m, C.c_float(thresh), x.ctypes, y.ctypes)
        return x[:n], y[:m]

    def ocsort_cost(self, dets, trks, vel, prev, vdc):
        dets, trks, vel, prev = f32(dets), f32(trks), f32(vel), f32(prev)
        nd, nt = dets.shape[0], trks.shape[0]
        cost = np.zeros((nd, nt), np.float32)
        iou = np.zeros((nd, nt), np.float32)
        self.lib.orc_ocsort_cost(dets.ctypes, nd, trks.ctypes, nt, vel.ctypes, prev.ctypes, C.c_float(vdc),
                                 cost.ctypes, iou.ctypes)
        return cost, iou

    def ocsort_associate(self, dets, trks, vel, prev, thr, vdc):
        dets, trks, vel, prev = f32(dets), f32(trks), f32(vel), f32(prev)
        nd, nt = dets.shape[0], trks.shape[0]
        cap = 2 * (nd + nt) + 4
        matches = np.zeros(2 * cap, np.int32)
        umd, umt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n_umd, n_umt, used = C.c_int(), C.c_int(), C.c_int()
        k = self.lib.orc_ocsort_associate(dets.ctypes, nd, trks.ctypes, nt, vel.ctypes, prev.ctypes,
                                          C.c_float(thr), C.c_float(vdc), matches.ctypes, umd.ctypes,
                                          C.byref(n_umd), umt.ctypes, C.byref(n_umt), C.byref(used))
        return matches[:2 * k].reshape(-1, 2).copy(), umd[:n_umd.value].copy(), umt[:n_umt.value].copy(), bool(used.value)

    def feat_update(self, mode, feat, src, alpha=0.9, alpha_i=None):
        feat, src = f32(feat).copy(), f32(src)
        n, d = src.shape
        ai = f32(alpha_i) if alpha_i is not None else None
        self.lib.orc_feat_update_alpha.argtypes = [C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.orc_feat_update_alpha(int(mode), C.c_float(alpha), ai.ctypes if ai is not None else None, n, d, feat.ctypes, src.ctypes)
        return feat

    def kf_initiate(self, kind, meas):
        meas = f32(meas).reshape(-1, 4)
        n, d = meas.shape[0], KF_DIM[kind]
        mean = np.zeros((n, d), np.float32)
        cov = np.zeros((n, d, d), np.float32)
        self.lib.orc_kf_initiate(kind, n, meas.ctypes, None, mean.ctypes, cov.ctypes)
        return mean, cov

    def kf_predict(self, kind, mean, cov, q=None):
        mean, cov = f32(mean).copy(), f32(cov).copy()
        qp = f32(q).ctypes if q is not None else None
        self.lib.orc_kf_predict(kind, mean.shape[0], qp, mean.ctypes, cov.ctypes)
        return mean, cov

    def kf_update(self, kind, mean, cov, meas, q=None):
        mean, cov, meas = f32(mean).copy(), f32(cov).copy(), f32(meas).reshape(-1, 4)
        qp = f32(q).ctypes if q is not None else None
        self.lib.orc_kf_update(kind, mean.shape[0], meas.ctypes, qp, mean.ctypes, cov.ctypes)
        return mean, cov

    def kf_update_conf(self, mean, cov, meas, conf):
        """XYAH update with per-measurement confidences (NSA Kalman)"""
        mean, cov, meas, conf = f32(mean).copy(), f32(cov).copy(), f32(meas).reshape(-1, 4), f32(conf)
        self.lib.orc_kf_update_conf.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.orc_kf_update_conf(mean.shape[0], meas.ctypes.data, conf.ctypes.data, mean.ctypes.data, cov.ctypes.data)
        return mean, cov

    def kf_warp(self, kind, mean, cov, warp9):
        mean, cov, w = f32(mean).copy(), f32(cov).copy(), f32(warp9).reshape(9)
        if self.lib.orc_kf_warp(kind, mean.shape[0], w.ctypes, mean.ctypes, cov.ctypes) != 0:
            raise ValueError("orc_kf_warp: filter kind without a camera-motion step")
        return mean, cov

    def box_convert(self, op, boxes):
        boxes = f32(boxes).reshape(-1, 4)
        out = np.zeros_like(boxes)
        self.lib.orc_box_convert(op, boxes.shape[0], boxes.ctypes, out.ctypes)
        return out

    def set_arith_mode(self, mode):
        """0: the canonical summation orders (what the kernels reproduce bit for bit); 1: the alternative, equally plausible orders of
        everything that depends on Eigen in the reference (oracle/orc_kf.hpp) — process-wide"""
        self.lib.orc_set_arith_mode(int(mode))

    def tracker(self, kind, params=None):
        return OracleTracker(self, kind, params)

    def ucmc(self, params=None, camera=None):
        """UCMCTrack (kind 6): params [det_thresh, max_age, a1, a2, wx, wy, vmax, dt, high_score] in double precision, camera (Ki 3 x 4, Ko 4 x 4)"""
        return OracleTracker(self, UCMC, params, camera)

    def ucmc_distance(self, x, P, y, R):
        x, P, y, R = (np.ascontiguousarray(a, np.float64) for a in (x, P, y, R))
        n, m = x.shape[0], y.shape[0]
        out = np.zeros((n, m), np.float32)
        self.lib.orc_ucmc_distance.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.orc_ucmc_distance(n, x.ctypes.data, P.ctypes.data, m, y.ctypes.data, R.ctypes.data, out.ctypes.data)
        return out


class OracleTracker:
    def __init__(self, orc, kind, params=None, camera=None):
        self.orc, self.kind = orc, kind
        if kind == UCMC:
            p = np.ascontiguousarray(params if params is not None else [], np.float64)
            ki = ko = None
            if camera is not None:
                ki, ko = (np.ascontiguousarray(a, np.float64).reshape(-1) for a in camera)
            orc.lib.orc_ucmc_create.restype = C.c_void_p
            orc.lib.orc_ucmc_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            self.h = orc.lib.orc_ucmc_create(p.ctypes.data if p.size else None, int(p.size), ki.ctypes.data if ki is not None else None,
                                             ko.ctypes.data if ko is not None else None)
            assert self.h
            self._out = np.zeros((4096, 8), np.float32)
            return
        p = f32(params if params is not None else [])
        self.h = orc.lib.orc_tracker_create(kind, p.ctypes if p.size else None, int(p.size))
        assert self.h
        self._out = np.zeros((4096, 8), np.float32)

    def __del__(self):
        if getattr(self, "h", None):
            self.orc.lib.orc_tracker_destroy(self.h)
            self.h = None

    def reset(self):
        self.orc.lib.orc_tracker_reset(self.h)

    def set_camera_motion(self, warp2x3):
        w = f32(warp2x3).reshape(6)
        if self.orc.lib.orc_tracker_set_warp(self.h, w.ctypes) != 0:
            raise ValueError("only BoT-SORT takes a camera-motion warp")

    def update(self, dets, embs=None):
        dets = f32(dets).reshape(-1, 6)
        e, d = None, 0
        if embs is not None and np.size(embs):
            embs = f32(embs)
            e, d = embs.ctypes, embs.shape[1]
        while True:
            r = self.orc.lib.orc_tracker_update(self.h, dets.ctypes, dets.shape[0], e, d, self._out.ctypes,
                                                self._out.shape[0])
            if r >= 0:
                return self._out[:r].copy()
            self._out = np.zeros((-r + 64, 8), np.float32)

    def laps(self):
        out = []
        cap = 1 << 15
        x, y = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        for k in range(self.orc.lib.orc_tracker_lap_count(self.h)):
            n, m = C.c_int(), C.c_int()
            rc = self.orc.lib.orc_tracker_lap_get(self.h, k, C.byref(n), C.byref(m), x, y, cap)
            assert rc == 0
            out.append((x[:n.value].copy(), y[:m.value].copy()))
        return out

    def dump_states(self):
        w = C.c_int()
        buf = np.zeros(1 << 16, np.float32)
        while True:
            r = self.orc.lib.orc_tracker_dump_states(self.h, buf.ctypes, buf.size, C.byref(w))
            if r >= 0:
                return buf[:r * w.value].reshape(r, w.value).copy() if r else np.zeros((0, 0), np.float32)
            buf = np.zeros((-r + 8) * max(w.value, 1), np.float32)

    def dump_f64(self):
        """UCMCTrack: [rows, 26] float64 — id, state, death, birth, det_idx, age, x(4), P(16)"""
        buf = np.zeros((4096, 26), np.float64)
        self.orc.lib.orc_ucmc_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        r = self.orc.lib.orc_ucmc_dump(self.h, buf.ctypes.data, buf.shape[0])
        assert r >= 0
        return buf[:r].copy()

    def dump_features(self):
        d = C.c_int()
        buf = np.zeros(1 << 22, np.float32)
        self.orc.lib.orc_tracker_dump_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        r = self.orc.lib.orc_tracker_dump_features(self.h, buf.ctypes, buf.size, C.byref(d))
        assert r >= 0
        return buf[:r * d.value].reshape(r, d.value).copy() if d.value else np.zeros((r, 0), np.float32)


_cached = None


def load():
    global _cached
    if _cached is None:
        build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached
