"""Per-phase cycles of the exact assignment kernel on an OC-SORT first-association problem of the C4 shape (2048 detections x 4096
tracks, cost -(IoU + direction term), a few exactly duplicated tracks as quirk Q4 leaves them). GPU box: python tools/c4_lap_phases.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream

ctx = L.Context(0)
nt, nd = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 2048)
s = SynthStream(nt, nd, 5)
d, _ = s.next_frame()
trk = np.stack([s.c[:, 0] - s.w / 2, s.c[:, 1] - s.h / 2, s.c[:, 0] + s.w / 2, s.c[:, 1] + s.h / 2], 1).astype(np.float32)
r = np.random.default_rng(1)
trk[-20:] = trk[:20]  # duplicated tracks
vel = r.normal(0, 1, (nt, 2)).astype(np.float32)
vel /= np.linalg.norm(vel, axis=1, keepdims=True)
vel[-20:] = vel[:20]
prev = np.concatenate([trk + r.normal(0, 2, trk.shape).astype(np.float32), np.ones((nt, 1), np.float32)], 1)
prev[-20:] = prev[:20]
dets5 = d[:, :5].astype(np.float32)
cost, iou = ctx.ocsort_cost(dets5, trk, vel, prev, 0.2)
lib = ctx.lib
lib.mot_lap_solve_prof_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
x, y = np.zeros(nd, np.int32), np.zeros(nt, np.int32)
info = C.c_int(0)
prof = np.zeros(36, np.int64)
for rep in range(2):
    t0 = time.time()
    ctx._chk(lib.mot_lap_solve_prof_host(ctx.h, cost.ctypes.data, nd, nt, C.c_float(-0.3), L.LAP_OCSORT if hasattr(L, "LAP_OCSORT") else 2, iou.ctypes.data,
                                         C.c_float(0.3), x.ctypes.data, y.ctypes.data, C.byref(info), prof.ctypes.data))
    dt = time.time() - t0
p = prof
print("shape", nd, "x", nt, "host ms", round(dt * 1e3, 1), "info", info.value)
print("cycles: colmin", p[0], "transfer", p[1], "carr", p[2], "aug", p[3], "| passes: transfer", p[4], "carr", p[5], "aug", p[6], "n+m", p[7])
print("per pass: transfer", p[1] / max(p[4], 1), "carr", p[2] / max(p[5], 1), "aug", p[3] / max(p[6], 1), "total s @2.1GHz", p[:4].sum() / 2.1e9)
