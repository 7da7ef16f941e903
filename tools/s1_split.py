"""Where one north-star problem spends its cycles when it has the GPU to itself (S streams, default 1): the fast path's cycle counters
(mot_lap_fast_stats) over the steady-state frames of a device ByteTrack batch. Usage: python tools/s1_split.py [S]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P, M = 1000, 500
dev = L.DeviceByteTrack(S, 2048, 512)
st = [SynthStream(P, M, 1234 + s) for s in range(S)]
for f in range(80):
    dets = np.zeros((S, 512, 6), np.float32)
    cnt = np.zeros(S, np.int32)
    for s in range(S):
        d, _ = st[s].next_frame()
        cnt[s] = len(d); dets[s, :len(d)] = d
    if f == 40:
        dev.ctx.lap_fast_stats(reset=True)
    dev.step(dets, cnt)
h = dev.ctx.lap_fast_stats()
n = max(h["fast"], 1)
print({k: (round(v / n, 1) if k.startswith("cycles") or k in ("searches", "column_scans", "lane0_candidates", "lane0_pairs_evaluated") else v) for k, v in h.items() if v})
