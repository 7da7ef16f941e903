"""Distribution of single-problem solve cycles over many different north-star-shaped problems (frames of many seeded
streams, first association of ByteTrack): a launch over thousands of problems lasts as long as its SLOWEST problem, so
the tail matters as much as the mean. Needs a gfx950 GPU: `gpurun -- python tools/lap_latency_distribution.py`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream

ctx = L.Context(0)
P, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 500)
tot, phases = [], []
for seed in range(120):
    s = SynthStream(P, M, 1234 + seed)
    for fr in range(3):
        d, _ = s.next_frame()
    tb = np.stack([s.c[:, 0] - s.w / 2, s.c[:, 1] - s.h / 2, s.c[:, 0] + s.w / 2, s.c[:, 1] + s.h / 2], 1).astype(np.float32)
    tb = tb + np.random.RandomState(seed).randn(*tb.shape).astype(np.float32) * 2
    hi = d[d[:, 4] > 0.45]
    ctx.lap_geom(tb, hi[:, :4], 0.8, L.COST_IOU_DIST_FUSE, hi[:, 4], prof=True)
    p = np.array(ctx._prof, np.int64)
    tot.append(p[:4].sum())
    phases.append(p[:8].copy())
tot = np.array(tot) / 2.4e6
ph = np.array(phases)
print("problems", len(tot), "ms: mean %.2f median %.2f p90 %.2f max %.2f min %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), tot.min()))
worst = int(tot.argmax())
print("worst problem phases (cycles colmin, transfer, carr, aug | n_uniq, serial carr, serial aug, n):", ph[worst].tolist())
print("mean phases:", ph.mean(0).round(0).tolist())
print("serial aug paths: mean %.1f max %d; serial carr rounds: mean %.0f max %d" % (ph[:, 6].mean(), ph[:, 6].max(), ph[:, 5].mean(), ph[:, 5].max()))
