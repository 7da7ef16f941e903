"""Per-frame latency of ONE tracker object of an f3 tracker (Batch of one stream, stepped frame by frame): which frames are slow, how many
rows they emit, how many device flushes they take.  python tools/f3_latency_probe.py [kind] [frames]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
kind = sys.argv[1] if len(sys.argv) > 1 else "deepocsort"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 120
P, M = 256, 128
dets, embs = SynthStream(P, M, 900, 32).frames(40)
b = L.Batch(kind, 1, threads=1, record_laps=False)
cnt = np.full(1, M, np.int32)
lat, rows, fl = [], [], []
prev = 0
for f in range(F):
    k = f % 78
    k = k if k < 40 else 78 - k  # back and forth over the 40 frames
    k = min(k, 39)
    d = np.ascontiguousarray(dets[k][None]); e = np.ascontiguousarray(embs[k][None]) if kind in ("deepocsort", "strongsort") else None
    t0 = time.perf_counter()
    out = b.step(d, cnt, e)
    lat.append(1e3 * (time.perf_counter() - t0))
    c = b.counters()
    fl.append(c["flushes"] - prev); prev = c["flushes"]
    try: rows.append(int(b._cnt[0]))
    except Exception: rows.append(-1)
lat = np.array(lat)
print(kind, "median %.3f ms p90 %.3f p99 %.3f max %.3f mean %.3f" % (np.median(lat[20:]), np.percentile(lat[20:], 90), np.percentile(lat[20:], 99), lat[20:].max(), lat[20:].mean()))
for f in range(20, F):
    if lat[f] > 2 * np.median(lat[20:]): print("  slow frame", f, "%.3f ms" % lat[f], "flushes", fl[f], "rows", rows[f])
print("flushes per frame", np.mean(fl[20:]), "rows", rows[20:40])
st = L.Context(0).lap_behind_stats()
n = max(1, st["problems"])
print("behind the fast path:", st["problems"], "problems; per problem:", {k: round(v / n) for k, v in st["sum"].items() if v})
print("slowest:", {k: v for k, v in st["slowest"].items() if v}, st["slowest_cycles"])
