"""The exact lapjv emulation on ONE north-star-sized problem: costs recomputed from the boxes inside the solver (what the declined problems of
the north-star path run through today) against the same costs read from a materialised matrix. Cycles from mot_lap_task.prof.
Usage (GPU): python tools/exact_geom_vs_matrix.py"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from motcpp_amd import _lib as L  # noqa: E402
from motcpp_amd.synth import SynthStream  # noqa: E402
from tests import orclib  # noqa: E402

o = orclib.load()
ctx = L.Context(0)
for P, M in ((1000, 500), (1024, 512)):
    s = SynthStream(P, M, 1)
    d, _ = s.next_frame()
    tb = np.stack([s.c[:, 0] - s.w / 2, s.c[:, 1] - s.h / 2, s.c[:, 0] + s.w / 2, s.c[:, 1] + s.h / 2], 1).astype(np.float32)
    hi = d[d[:, 4] > 0.45]
    for rep in range(2):
        t0 = time.time()
        xg, yg, xv, info = ctx.lap_geom(tb, hi[:, :4], 0.8, L.COST_IOU_DIST_FUSE, hi[:, 4], prof=True)
        dtg = time.time() - t0
    pg = ctx._prof.copy()
    cost = o.fuse_score(o.iou_distance(tb, hi[:, :4]), hi[:, 4])
    n, m = cost.shape
    x, y = np.full(n, -1, np.int32), np.full(m, -1, np.int32)
    info = C.c_int(0)
    prof = np.zeros(36, np.int64)
    ctx.lib.mot_lap_solve_prof_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
    for rep in range(2):
        t0 = time.time()
        ctx._chk(ctx.lib.mot_lap_solve_prof_host(ctx.h, cost.ctypes.data, n, m, C.c_float(0.8), 0, None, C.c_float(0.0), x.ctypes.data, y.ctypes.data,
                                                 C.byref(info), prof.ctypes.data))
        dtm = time.time() - t0
    print(P, M, "rows x cols", n, m, "equal", np.array_equal(x, xg) and np.array_equal(y, yg),
          "| geometry: host ms", round(dtg * 1e3, 2), "solver ms @2.4GHz", round(pg[:4].sum() / 2.4e6, 2),
          "| matrix: host ms", round(dtm * 1e3, 2), "solver ms @2.4GHz", round(prof[:4].sum() / 2.4e6, 2))
