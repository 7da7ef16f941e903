"""Exact assignment kernel on REAL first-association problems of an OC-SORT 4096 x 2048 run (dumped from the oracle with
ORC_LAP_DUMP, expected answers next to them as .npz): parity, per-phase shader cycles and the shortest-path scan counters.
GPU box: python tools/c4_real_problem.py scratch/c4 [repeats]"""
import ctypes as C, glob, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motcpp_amd import _lib as L

ctx = L.Context(0)
lib = ctx.lib
lib.mot_lap_solve_prof_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
d = sys.argv[1] if len(sys.argv) > 1 else "scratch/c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bad = 0
for f in sorted(glob.glob(os.path.join(d, "*.bin"))):
    nr, nc = map(int, f.split("_")[-1][:-4].split("x"))
    raw = np.fromfile(f, np.float32)
    th, cost = float(raw[0]), np.ascontiguousarray(raw[1:].reshape(nr, nc))
    exp = np.load(f + ".npz")
    x, y = np.zeros(nr, np.int32), np.zeros(nc, np.int32)
    info = C.c_int(0)
    prof = np.zeros(36, np.int64)
    for rep in range(reps):
        t0 = time.time()
        ctx._chk(lib.mot_lap_solve_prof_host(ctx.h, cost.ctypes.data, nr, nc, C.c_float(th), 0, None, C.c_float(0.0), x.ctypes.data, y.ctypes.data,
                                             C.byref(info), prof.ctypes.data))
        dt = time.time() - t0
    ok = np.array_equal(x, exp["x"]) and np.array_equal(y, exp["y"])
    bad += 0 if ok else 1
    p = prof
    print(os.path.basename(f), "OK" if ok else "MISMATCH", "host ms %.1f" % (dt * 1e3), "| Mcycles colmin %.2f transfer %.2f carr %.2f aug %.2f = %.1f ms @2.4GHz" %
          (p[0] / 1e6, p[1] / 1e6, p[2] / 1e6, p[3] / 1e6, p[:4].sum() / 2.4e6),
          "| carr %d paths %d finds %d | steps %d members %d real %d events %d one-at-a-time %d refused %d lists %d" %
          (p[5], p[6], p[14], p[8], p[9], p[10], p[11], p[12], p[13], p[15]),
          "| aug Mcycles: classify %.1f dry %.1f apply %.1f evsort %.1f evreplay %.1f find %.1f one-at-a-time %.1f setup %.1f" % tuple(p[16:24] / 1e6),
          "| sub: " + " ".join("%.1f" % (v / 1e6) for v in p[24:36]), "| raw sub:", " ".join(str(int(v)) for v in p[24:36]), flush=True)
print("mismatches", bad)
