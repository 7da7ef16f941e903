import sys
sys.path.insert(0, ".")
import numpy as np
from motcpp_amd import _lib as L
from tests import orclib
from tests.test_gpu_primitives import boxes
orc = orclib.load()
ctx = L.Context(0)
for (n, m) in ((1000, 500), (700, 1500), (300, 200)):
    r = np.random.default_rng(3 * n + m)
    a = boxes(r, n, (1920, 1080))
    k = min(n, m)
    b = boxes(r, m, (1920, 1080))
    b[:k] = a[r.permutation(n)[:k]] + r.normal(0, 2, (k, 4)).astype(np.float32)
    conf = r.uniform(0.3, 1, m).astype(np.float32)
    dist = orc.iou_distance(a, b)
    for mode, cost, th in ((L.COST_IOU_DIST, dist, 0.7), (L.COST_IOU_DIST_FUSE, orc.fuse_score(dist, conf), 0.8), (L.COST_NEG_IOU, -orc.iou_batch(a, b), -0.3)):
        ctx.lap_fast_stats(reset=True)
        xo, yo = orc.linear_assignment(cost, th)
        xg, yg, xv, info = ctx.lap_geom(a, b, th, mode, conf)
        st = ctx.lap_fast_stats()
        print(n, m, mode, "equal", np.array_equal(xg, xo), {k: v for k, v in st.items() if v and (len(sys.argv) > 1 or (not k.startswith("cycles") and not k.startswith("lane0")))})
