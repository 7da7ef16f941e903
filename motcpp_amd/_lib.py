"""ctypes loader for the gfx950 kernel library (C ABI: include/motcpp_amd.h).

There is no CPU fallback: if the library is missing it must be built (motcpp_amd.build()), and on a box
without a gfx950 device creating a context raises."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(HERE, "lib")
HIP_LIB = os.path.join(LIBDIR, "libmotcpp_hip.so")
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
