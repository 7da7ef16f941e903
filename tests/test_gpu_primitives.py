"""GPU parity tests of the C-ABI primitives (through the _host entry points of include/motcpp_amd.h) against
the CPU oracle on the same seeded inputs. Integer results (assignment indices) must be identical; floats are
required to be bit-identical where the kernel follows the oracle's operation order (IoU family, Kalman,
cosine via the fp32 MFMA fmaf chain) and within 1e-4 relative otherwise (OC-SORT's acos term)."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from tests import orclib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def boxes(r, n, world=(1920, 1080)):
    cx, cy = r.uniform(0, world[0], n), r.uniform(0, world[1], n)
    w = r.uniform(30, 90, n)
    h = w * r.uniform(1.8, 2.6, n)
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)


@pytest.mark.parametrize("n,m", [(1, 1), (3, 130), (64, 64), (65, 63), (256, 128), (1000, 500), (777, 1231)])
def test_iou_family_bit_exact(ctx, orc, n, m):
    r = np.random.default_rng(n * 1000 + m)
    a, b = boxes(r, n, (600, 400)), boxes(r, m, (600, 400))
    b[: min(n, m) // 2] = a[: min(n, m) // 2] + r.normal(0, 2, (min(n, m) // 2, 4)).astype(np.float32)
    conf = r.uniform(0.1, 1, m).astype(np.float32)
    iou = orc.iou_batch(a, b)
    assert (iou > 0).sum() > 0 or min(n, m) < 2
    assert np.array_equal(ctx.iou_cost(a, b, L.COST_IOU), iou)
    dist = orc.iou_distance(a, b)
    assert np.array_equal(ctx.iou_cost(a, b, L.COST_IOU_DIST), dist)
    assert np.array_equal(ctx.iou_cost(a, b, L.COST_IOU_DIST_FUSE, conf), orc.fuse_score(dist, conf))
    assert np.array_equal(ctx.iou_cost(a, b, L.COST_NEG_IOU), -iou)


def test_iou_degenerate_boxes(ctx, orc):
    a = np.array([[0, 0, 0, 0], [10, 10, 5, 5], [0, 0, 100, 100], [np.nan, 0, 1, 1]], np.float32)
    b = np.array([[0, 0, 100, 100], [0, 0, 0, 0], [50, 50, 60, 60]], np.float32)
    assert np.array_equal(ctx.iou_cost(a, b, L.COST_IOU), orc.iou_batch(a, b), equal_nan=True)


@pytest.mark.parametrize("n,m,d", [(5, 7, 16), (64, 64, 256), (100, 33, 255), (1024, 512, 256)])
def test_cosine_mfma_matches_fmaf_chain(ctx, orc, n, m, d):
    r = np.random.default_rng(d + n)
    a = r.standard_normal((n, d)).astype(np.float32)
    b = r.standard_normal((m, d)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    k = min(n, m) // 2
    b[:k] = a[:k] + 0.05 * r.standard_normal((k, d)).astype(np.float32)
    ref = orc.cosine_distance(a, b)
    got = ctx.cosine_cost(a, b)
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-6)
    assert np.array_equal(got, ref), f"max abs diff {np.abs(got - ref).max()}"


@pytest.mark.parametrize("nd,nt", [(9, 5), (64, 64), (200, 333), (512, 1024), (2048, 4096)])  # the last: BASELINE configs[3]
def test_ocsort_cost(ctx, orc, nd, nt):
    r = np.random.default_rng(nd + nt)
    world = (800, 600) if nd * nt < 1e6 else (1920, 1080)
    trks = boxes(r, nt, world)
    dets = np.concatenate([boxes(r, nd, world), r.uniform(0.3, 1, (nd, 1)).astype(np.float32)], 1)
    k = min(nd, nt) // 2
    dets[:k, :4] = trks[:k] + r.normal(0, 3, (k, 4)).astype(np.float32)
    vel = r.standard_normal((nt, 2)).astype(np.float32)
    vel /= np.linalg.norm(vel, axis=1, keepdims=True) + 1e-6
    prev = np.concatenate([trks + r.normal(0, 5, (nt, 4)).astype(np.float32), r.uniform(0.3, 1, (nt, 1)).astype(np.float32)], 1)
    prev[::7] = -1
    cost_o, iou_o = orc.ocsort_cost(dets, np.concatenate([trks, np.zeros((nt, 1), np.float32)], 1), vel, prev, 0.2)
    cost_g, iou_g = ctx.ocsort_cost(dets, trks, vel, prev, 0.2)
    assert np.array_equal(iou_g, iou_o)
    assert np.allclose(cost_g, cost_o, rtol=1e-4, atol=1e-7)
    frac_exact = (cost_g == cost_o).mean()
    assert frac_exact > 0.999, frac_exact


def lap_case(r, kind, n, m):
    if kind == "dense":
        return r.uniform(0, 1, (n, m)).astype(np.float32), 0.8
    if kind == "dense_forced":
        return r.uniform(0, 1, (n, m)).astype(np.float32), 10.0
    if kind == "neg":
        return (-r.uniform(0, 1, (n, m))).astype(np.float32), -0.3
    if kind == "quant":
        return (r.integers(0, 6, (n, m)) / 5.0).astype(np.float32), 0.7
    if kind == "quant_forced":
        return (r.integers(0, 4, (n, m)) / 3.0).astype(np.float32), 5.0
    c = np.ones((n, m), np.float32)
    for i in range(n):
        if r.uniform() < 0.8:
            c[i, r.integers(m)] = r.uniform(0.05, 0.6)
    e = r.uniform(0, 1, (n, m)) < 0.03
    c[e] = r.uniform(0.2, 0.95, e.sum()).astype(np.float32)
    return c, 0.8


@pytest.mark.parametrize("kind", ["sparse", "dense", "dense_forced", "neg", "quant", "quant_forced"])
def test_lap_indices_identical(ctx, orc, kind):
    r = np.random.default_rng(abs(hash(kind)) % 997)
    for n, m in [(1, 1), (2, 3), (17, 9), (9, 17), (64, 64), (100, 130), (256, 128), (300, 300)]:
        c, th = lap_case(r, kind, n, m)
        xo, yo = orc.linear_assignment(c, th)
        xg, yg, info = ctx.lap(c, th)
        assert info == 0
        assert np.array_equal(xg, xo) and np.array_equal(yg, yo), (kind, n, m)


@pytest.mark.parametrize("n,m", [(9, 150), (30, 400), (150, 90), (64, 1000), (12, 70), (700, 2600), (2600, 700)])
def test_lap_contested_rows_over_long_tie_runs(ctx, orc, n, m):
    # OC-SORT-like costs: almost every entry exactly 0, a few negative ones, several rows wanting the same column. The
    # augmenting-path searches then rebuild tied sets of tens to thousands of columns (lap_core.hpp: the register replay
    # of _find_dense and the closed-form tie runs of the wide variants). The replay's chunk-boundary case — 64 tie records
    # with no new minimum among them — is what tests/test_gpu_trackers.py::test_ocsort_association_measures[1] runs into.
    r = np.random.default_rng(n * 31 + m)
    for trial in range(6 if n * m < 100000 else 2):
        c = np.zeros((n, m), np.float32)
        hot = r.permutation(m)[: max(2, min(n, m) // 3)]
        for i in range(n):
            for j in r.choice(hot, size=min(len(hot), 1 + trial % 3), replace=False):
                c[i, j] = -np.float32(r.integers(1, 6)) / 5 if trial % 2 else -r.uniform(0.05, 0.9)
        th = -0.1
        xo, yo = orc.linear_assignment(c, th)
        xg, yg, info = ctx.lap(c, th)
        assert info == 0 and np.array_equal(xg, xo) and np.array_equal(yg, yo), (n, m, trial)


def test_lap_north_star_and_global_workspace(ctx, orc):
    r = np.random.default_rng(5)
    for n, m in [(1000, 500), (2600, 1400)]:  # second one exceeds the LDS limit -> global-scratch variant
        c, th = lap_case(r, "sparse", n, m)
        xo, yo = orc.linear_assignment(c, th)
        xg, yg, _ = ctx.lap(c, th)
        assert np.array_equal(xg, xo) and np.array_equal(yg, yo), (n, m)


@pytest.mark.parametrize("n,m", [(1, 1), (40, 64), (256, 128), (1000, 500), (2600, 1400)])
def test_lap_on_the_fly_geometry(ctx, orc, n, m):
    # mot_lap_task.geom: the solver recomputes the IoU-family cost from boxes; decisions must equal the oracle's on the matrix
    r = np.random.default_rng(n + m)
    a = boxes(r, n, (1920, 1080))
    k = min(n, m)
    b = boxes(r, m, (1920, 1080))
    b[:k] = a[r.permutation(n)[:k]] + r.normal(0, 2, (k, 4)).astype(np.float32)
    if k > 8:
        b[1] = b[0]  # duplicate detection: exact ties
    conf = r.uniform(0.3, 1, m).astype(np.float32)
    dist = orc.iou_distance(a, b)
    for mode, cost, th in ((L.COST_IOU_DIST, dist, 0.7), (L.COST_IOU_DIST_FUSE, orc.fuse_score(dist, conf), 0.8),
                           (L.COST_NEG_IOU, -orc.iou_batch(a, b), -0.3)):
        xo, yo = orc.linear_assignment(cost, th)
        xg, yg, xv, info = ctx.lap_geom(a, b, th, mode, conf)
        assert info == 0 and np.array_equal(xg, xo) and np.array_equal(yg, yo), (n, m, mode)
        hit = xg >= 0
        assert np.array_equal(xv[hit], cost[np.arange(n)[hit], xg[hit]])
    # gate (OC-SORT rematch): nothing overlaps enough -> untouched
    far = boxes(r, m, (1920, 1080)) + 5000
    xg, yg, _, info = ctx.lap_geom(a, far, -0.3, L.COST_NEG_IOU, None, L.LAP_GATE_MIN, -0.3)
    assert info == 2 and (xg == -1).all() and (yg == -1).all()


def test_lap_ocsort_modes(ctx, orc):
    r = np.random.default_rng(11)
    # trivial one-to-one case (ocsort.cpp:684-696)
    iou = np.zeros((6, 5), np.float32)
    iou[0, 1], iou[2, 0], iou[5, 4] = 0.7, 0.5, 0.9
    x, y, info = ctx.lap(-iou, -0.3, L.LAP_OCSORT, iou=iou, gate=0.3)
    assert info == 1 and list(x) == [1, -1, 0, -1, -1, 4] and list(y) == [2, 0, -1, -1, 5]
    # two hits in one row -> falls through to lapjv
    iou[0, 2] = 0.6
    x, y, info = ctx.lap(-iou, -0.3, L.LAP_OCSORT, iou=iou, gate=0.3)
    xo, yo = orc.linear_assignment(-iou, -0.3)
    assert info == 0 and np.array_equal(x, xo) and np.array_equal(y, yo)
    # gate: nothing above the threshold -> untouched (ocsort.cpp:499)
    low = r.uniform(0, 0.25, (7, 9)).astype(np.float32)
    x, y, info = ctx.lap(-low, -0.3, L.LAP_GATE_MIN, gate=-0.3)
    assert info == 2 and (x == -1).all() and (y == -1).all()
    low[3, 3] = 0.8
    x, y, info = ctx.lap(-low, -0.3, L.LAP_GATE_MIN, gate=-0.3)
    xo, yo = orc.linear_assignment(-low, -0.3)
    assert info == 0 and np.array_equal(x, xo) and np.array_equal(y, yo)


@pytest.mark.parametrize("kind", [L.KF_XYSR, L.KF_XYAH, L.KF_XYWH])
def test_kalman_bit_exact(ctx, orc, kind):
    r = np.random.default_rng(kind)
    n = 700
    b = boxes(r, n)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    cx, cy = b[:, 0] + w / 2, b[:, 1] + h / 2
    if kind == L.KF_XYSR:
        z = np.stack([cx, cy, w * h, w / h], 1)
    elif kind == L.KF_XYAH:
        z = np.stack([cx, cy, w / h, h], 1)
    else:
        z = np.stack([cx, cy, w, h], 1)
    z = z.astype(np.float32)
    q = np.array([1e-4, 1e-4, 1e-8], np.float32) if kind == L.KF_XYSR else None
    mo, co = orc.kf_initiate(kind, z)
    mg, cg = ctx.kf_apply(kind, 0, None, None, meas=z)
    assert np.array_equal(mg, mo) and np.array_equal(cg, co)
    scale = np.array([2, 2, 60 if kind == L.KF_XYSR else (0.01 if kind == L.KF_XYAH else 2), 0.01 if kind == L.KF_XYSR else 2],
                     np.float32)
    for step in range(6):
        mo, co = orc.kf_predict(kind, mo, co, q)
        mg, cg, bx = ctx.kf_apply(kind, 1, mg, cg, q=q, want_boxes=True)
        assert np.array_equal(mg, mo) and np.array_equal(cg, co), f"predict step {step}"
        zz = (mo[:, :4] + r.normal(0, 1, (n, 4)).astype(np.float32) * scale).astype(np.float32)
        mo, co = orc.kf_update(kind, mo, co, zz, q)
        mg, cg = ctx.kf_apply(kind, 2, mg, cg, meas=zz, q=q)
        assert np.allclose(mg, mo, rtol=1e-4, atol=1e-5) and np.allclose(cg, co, rtol=1e-4, atol=1e-4), f"update step {step}"
        assert np.array_equal(mg, mo) and np.array_equal(cg, co), f"update step {step}: max diff {np.abs(cg - co).max()}"
    # state -> box conversion
    if kind == L.KF_XYSR:
        ref = orc.box_convert(1, mo[:, :4])
    elif kind == L.KF_XYAH:
        ref = orc.box_convert(3, orc.box_convert(6, mo[:, :4]))
    else:
        ref = orc.box_convert(3, mo[:, :4])
    _, _, bx = ctx.kf_apply(kind, 1, mg, cg, q=q, want_boxes=True)
    mo2, _ = orc.kf_predict(kind, mo, co, q)
    if kind == L.KF_XYSR:
        ref = orc.box_convert(1, mo2[:, :4])
    elif kind == L.KF_XYAH:
        ref = orc.box_convert(3, orc.box_convert(6, mo2[:, :4]))
    else:
        ref = orc.box_convert(3, mo2[:, :4])
    assert np.array_equal(bx, ref)


def test_kalman_predict_flags(ctx, orc):
    r = np.random.default_rng(3)
    z = np.stack([r.uniform(0, 1000, 50), r.uniform(0, 1000, 50), r.uniform(0.3, 0.6, 50), r.uniform(50, 200, 50)], 1).astype(np.float32)
    m, c = orc.kf_initiate(L.KF_XYAH, z)
    m[:, 4:] = r.normal(0, 1, (50, 4)).astype(np.float32)
    flags = (np.arange(50) % 2).astype(np.uint8)  # MOT_KF_ZERO_V7 on odd items
    mref = m.copy()
    mref[flags == 1, 7] = 0
    mo, co = orc.kf_predict(L.KF_XYAH, mref, c)
    mg, cg = ctx.kf_apply(L.KF_XYAH, 1, m, c, flags=flags)
    assert np.array_equal(mg, mo) and np.array_equal(cg, co)


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5])  # hmiou, giou, ciou, diou, centroid (include/motcpp/utils/iou.hpp:122-414)
def test_association_measures(ctx, orc, kind):
    for n, m in [(1, 1), (3, 130), (65, 63), (256, 128)]:
        r = np.random.default_rng(kind * 100 + n)
        a, b = boxes(r, n, (600, 400)), boxes(r, m, (600, 400))
        k = min(n, m) // 2
        b[:k] = a[:k] + r.normal(0, 3, (k, 4)).astype(np.float32)
        ref = orc.asso_batch(kind, a, b, (640, 480))
        got = ctx.assoc_cost(a, b, kind, (640, 480))
        if kind == 3:  # ciou: atan through fp64 on both sides; last-bit differences only where two fp64 libms disagree
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-6)
            assert (got == ref).mean() >= 0.999
        else:
            assert np.array_equal(got, ref), np.abs(got - ref).max()
        neg = ctx.assoc_cost(a, b, kind, (640, 480), mode=L.COST_NEG_IOU)
        assert np.array_equal(neg, -got)


def _warps(r):
    # identity, pure translation, a small similarity (what ECC returns between consecutive frames), a full affine, and a
    # projective one (multi_gmc divides by the third coordinate; the XYSR rule ignores the last row)
    th = 0.02
    sim = [[1.01 * np.cos(th), -1.01 * np.sin(th), 3.5], [1.01 * np.sin(th), 1.01 * np.cos(th), -2.25], [0, 0, 1]]
    return [np.eye(3), [[1, 0, 12.5], [0, 1, -7.25], [0, 0, 1]], sim,
            [[0.97, 0.04, 5], [-0.03, 1.02, 1], [0, 0, 1]], [[1, 0.01, 2], [0.02, 1, 3], [1e-5, -2e-5, 1.001]]]


@pytest.mark.parametrize("kind", [L.KF_XYSR, L.KF_XYWH])
def test_camera_motion_warp_bit_exact(ctx, orc, kind):
    # mot_kf_warp against BotSTrack::multi_gmc (botsort.cpp:60-91) / KalmanFilterXYSR::apply_affine_correction
    # (xysr_kf.cpp:114-141) as restated in oracle/; standalone and fused behind a predict
    r = np.random.default_rng(40 + kind)
    n = 300
    b = boxes(r, n)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    cx, cy = b[:, 0] + w / 2, b[:, 1] + h / 2
    z = (np.stack([cx, cy, w * h, w / h], 1) if kind == L.KF_XYSR else np.stack([cx, cy, w, h], 1)).astype(np.float32)
    m0, c0 = orc.kf_initiate(kind, z)
    m0[:, 4:] = r.normal(0, 2, m0[:, 4:].shape).astype(np.float32)
    m0, c0 = orc.kf_predict(kind, m0, c0)  # off-diagonal covariance terms
    for W in _warps(r):
        W = np.asarray(W, np.float32)
        mo, co = orc.kf_warp(kind, m0, c0, W)
        mg, cg, bx = ctx.kf_warp(kind, m0, c0, W, want_boxes=True)
        assert np.array_equal(mg, mo) and np.array_equal(cg, co)
        if kind == L.KF_XYWH:
            ref = np.stack([mo[:, 0] - mo[:, 2] / 2, mo[:, 1] - mo[:, 3] / 2, mo[:, 0] + mo[:, 2] / 2, mo[:, 1] + mo[:, 3] / 2], 1)
            assert np.array_equal(bx, ref.astype(np.float32))
            assert np.array_equal(cg, c0) and np.array_equal(mg[:, 4:], m0[:, 4:])  # only cx,cy,w,h move
        mp, cp = orc.kf_predict(kind, m0, c0)
        mo, co = orc.kf_warp(kind, mp, cp, W)
        mg, cg = ctx.kf_warp(kind, m0, c0, W, predict_first=True)
        assert np.array_equal(mg, mo) and np.array_equal(cg, co)


def test_camera_motion_warp_rejects_xyah(ctx):
    m, c = np.zeros((2, 8), np.float32), np.zeros((2, 8, 8), np.float32)
    with pytest.raises(L.MotError):
        ctx.kf_warp(L.KF_XYAH, m, c, np.eye(3, dtype=np.float32))


@pytest.mark.parametrize("n,m,world", [(33, 7, 300), (200, 150, 400), (300, 100, 600), (400, 50, 1920), (256, 128, 250), (128, 256, 500)])
def test_lap_sparse_first_phase_shapes(ctx, orc, n, m, world):
    # the plain-cost kernel variants with LDS-resident rows take phase 1's column minima from x-sorted candidate rows
    # (lap_core.hpp::sparse_column_minima): every rank-counting width (<= 4, 8, 16 rows per lane), crowded scenes (long
    # candidate lists), a coarse coordinate grid (equal x1, touching boxes, exact cost ties), NaN rows and a NaN column
    r = np.random.default_rng(n * 7 + m)
    a = boxes(r, n, (world, world // 2))
    k = min(n, m)
    b = boxes(r, m, (world, world // 2))
    b[:k] = a[r.permutation(n)[:k]] + r.normal(0, 2, (k, 4)).astype(np.float32)
    b[::4] = np.round(b[::4] / 8) * 8
    a[::3] = np.round(a[::3] / 8) * 8
    b[1] = b[0]
    a[5, 2] = np.nan
    a[min(n - 1, 40), 0] = np.nan
    if m > 20:
        b[17, 3] = np.nan
    conf = r.uniform(0.3, 1, m).astype(np.float32)
    iou = orc.iou_batch(a, b)
    dist = orc.iou_distance(a, b)
    for mode, cost, th in ((L.COST_IOU, iou, 0.3), (L.COST_IOU_DIST, dist, 0.7), (L.COST_IOU_DIST_FUSE, orc.fuse_score(dist, conf), 0.8),
                           (L.COST_NEG_IOU, -iou, -0.3)):
        xo, yo = orc.linear_assignment(cost, th)
        xg, yg, xv, info = ctx.lap_geom(a, b, th, mode, conf)
        assert info == 0 and np.array_equal(xg, xo) and np.array_equal(yg, yo), (n, m, mode)


@pytest.mark.parametrize("n,m", [(64, 40), (300, 200), (1000, 500), (700, 1500)])
def test_lap_fast_path_and_exact_path_agree_with_the_oracle(ctx, orc, n, m):
    """mot_lap_solve = lap_sparse_kernel (certifies the unique optimum over the viable pairs) + lap_kernel (the exact lapjv
    emulation for everything else). Clean tracking problems must be finished by the fast path; the same problems forced down
    the exact path (prof requested) and problems with ties (which the fast path must decline) give the oracle's answer too."""
    r = np.random.default_rng(3 * n + m)
    a = boxes(r, n, (1920, 1080))
    k = min(n, m)
    b = boxes(r, m, (1920, 1080))
    b[:k] = a[r.permutation(n)[:k]] + r.normal(0, 2, (k, 4)).astype(np.float32)
    conf = r.uniform(0.3, 1, m).astype(np.float32)
    dist = orc.iou_distance(a, b)
    cases = ((L.COST_IOU_DIST, dist, 0.7), (L.COST_IOU_DIST_FUSE, orc.fuse_score(dist, conf), 0.8), (L.COST_NEG_IOU, -orc.iou_batch(a, b), -0.3))
    ctx.lap_fast_stats(reset=True)
    for mode, cost, th in cases:
        xo, yo = orc.linear_assignment(cost, th)
        for prof in (False, True):  # False: fast path first; True: straight to the exact emulation
            xg, yg, xv, info = ctx.lap_geom(a, b, th, mode, conf, prof=prof)
            assert info == 0 and np.array_equal(xg, xo) and np.array_equal(yg, yo), (n, m, mode, prof)
            hit = xg >= 0
            assert np.array_equal(xv[hit], cost[np.arange(n)[hit], xg[hit]])
        xm, ym, _ = ctx.lap(cost, th)  # materialised-matrix source of the fast path
        assert np.array_equal(xm, xo) and np.array_equal(ym, yo)
    st = ctx.lap_fast_stats()
    assert st["fast"] == 2 * len(cases) and st["not_attempted"] == len(cases), st
    # ties: a duplicated detection -> declined by the certificate, solved by the exact path, same answer
    b2 = b.copy()
    b2[1] = b2[0]
    cost = orc.fuse_score(orc.iou_distance(a, b2), conf)
    xo, yo = orc.linear_assignment(cost, 0.8)
    ctx.lap_fast_stats(reset=True)
    xg, yg, _, _ = ctx.lap_geom(a, b2, 0.8, L.COST_IOU_DIST_FUSE, conf)
    assert np.array_equal(xg, xo) and np.array_equal(yg, yo)
    # ... and a case where the duplicate certainly decides the matching: one track, the same detection twice
    a1 = np.array([[0, 0, 50, 100], [900, 500, 950, 600]], np.float32)
    b1 = np.array([[2, 1, 52, 101], [2, 1, 52, 101], [1500, 800, 1550, 900]], np.float32)
    c1 = orc.iou_distance(a1, b1)
    xo, yo = orc.linear_assignment(c1, 0.7)
    ctx.lap_fast_stats(reset=True)
    xg, yg, _, _ = ctx.lap_geom(a1, b1, 0.7, L.COST_IOU_DIST, None)
    assert np.array_equal(xg, xo) and np.array_equal(yg, yo)
    st = ctx.lap_fast_stats()
    assert st["fast"] == 0 and st["not_unique"] == 1, st


@pytest.mark.parametrize("n,m", [(1000, 500), (600, 256), (256, 128)])
def test_lap_tie_heavy_problems_behind_the_fast_path(ctx, orc, n, m):
    """Problems the certificate must decline (duplicated tracks with detections exactly on them: non-unique optima) go to the exact
    emulation launched BEHIND the sparse solver — full / all-LDS state (lap_kernel.hip: lds_mode 2 / 5), void real rows in the
    shortest-path search (lap_core.hpp) — and come out index for index like the oracle's lapjv; mot_lap_behind_stats counts them."""
    r = np.random.default_rng(7 * n + m)
    for trial in range(3):
        a = boxes(r, n, (1920, 1080))
        ndup = n // 8
        a[r.permutation(n)[:ndup]] = a[r.integers(0, n, ndup)]  # duplicated tracks
        src = r.integers(0, n, m)
        b = (a[src] + r.normal(0, 2.0, (m, 4))).astype(np.float32)
        b[::2] = a[src[::2]]  # detections exactly on (possibly duplicated) tracks
        far = r.random(m) < 0.25
        b[far] += np.float32(5000.0)  # detections with no track near them
        conf = r.uniform(0.3, 1, m).astype(np.float32)
        for mode, th in ((L.COST_IOU_DIST, 0.7), (L.COST_IOU_DIST_FUSE, 0.8)):
            dist = orc.iou_distance(a, b)
            cost = dist if mode == L.COST_IOU_DIST else orc.fuse_score(dist, conf)
            xo, yo = orc.linear_assignment(cost, th)
            ctx.lap_fast_stats(reset=True)
            ctx.lap_behind_stats(reset=True)
            xg, yg, _, info = ctx.lap_geom(a, b, th, mode, conf)
            assert info == 0 and np.array_equal(xg, xo) and np.array_equal(yg, yo), (n, m, trial, mode)
            st, bh = ctx.lap_fast_stats(), ctx.lap_behind_stats()
            assert st["fast"] == 0 and bh["problems"] == 1 and bh["slowest_cycles"] > 0, (st, bh)


@pytest.mark.parametrize("n,d", [(1, 8), (77, 64), (512, 256), (45, 256), (130, 100)])
def test_appearance_post_processing_bit_exact(ctx, orc, n, d):
    """mot_feat_update against the oracle's restatement of BotSTrack's feature handling (botsort.cpp:38-46 set +
    normalise, :158-169 EMA 0.9 / 0.1 + renormalise) and of ReIDBackend::normalize_features (reid_backend.cpp:72-88, rows of
    norm <= 1e-6 left alone). Same k-ordered inner product on both sides: bit-identical."""
    r = np.random.default_rng(n * 7 + d)
    src = r.standard_normal((n, d)).astype(np.float32)
    if n > 3:
        src[1] = 0.0            # a zero row is not normalised in any mode
        src[2] *= 1e-9          # norm below the ReID rule's 1e-6 but > 0
    old = r.standard_normal((n, d)).astype(np.float32)
    old /= np.maximum(np.linalg.norm(old, axis=1, keepdims=True), 1e-12).astype(np.float32)
    for mode in (0, 1, 2, 3):
        g = ctx.feat_update(mode, old, src)
        o = orc.feat_update(mode, old, src)
        assert np.array_equal(g, o), (mode, np.abs(g - o).max())
    # DeepOC-SORT's update_emb: EMA with a weight per detection (dets_alpha), normalised where the norm exceeds 1e-6
    ai = r.uniform(0.5, 1.0, n).astype(np.float32)
    g, o = ctx.feat_update(3, old, src, alpha_i=ai), orc.feat_update(3, old, src, alpha_i=ai)
    assert np.array_equal(g, o), np.abs(g - o).max()
    if n > 3:
        assert np.array_equal(ctx.feat_update(2, old, src)[2], src[2])  # untouched by the ReID rule
        assert abs(np.linalg.norm(ctx.feat_update(0, old, src)[2]) - 1.0) < 1e-5  # but normalised by BotSTrack's
    # two EMA steps in a row stay unit-norm and follow the oracle
    g1, o1 = ctx.feat_update(1, old, src), orc.feat_update(1, old, src)
    src2 = r.standard_normal((n, d)).astype(np.float32)
    assert np.array_equal(ctx.feat_update(1, g1, src2), orc.feat_update(1, o1, src2))


@pytest.mark.parametrize("n,m,d", [(3, 5, 7), (64, 64, 32), (200, 130, 100), (1024, 512, 256)])
def test_embedding_metrics(ctx, orc, n, m, d):
    """utils::embedding_distance: cosine (matching.cpp:79-92), euclidean (:93-101) and the raw dot product DeepOC-SORT uses
    (deepocsort.cpp:404), all k-ordered fp32 chains -> bit-identical to the restatement; plus the float64 value within 1e-4."""
    r = np.random.default_rng(n + m + d)
    a = r.standard_normal((n, d)).astype(np.float32)
    b = r.standard_normal((m, d)).astype(np.float32)
    for metric in (0, 1, 2):
        g, o = ctx.embedding_cost(metric, a, b), orc.embedding_distance(metric, a, b)
        assert np.array_equal(g, o), (metric, np.abs(g - o).max())
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    eu = np.sqrt(((a64[:, None, :] - b64[None, :, :]) ** 2).sum(-1)) if n * m * d < 5e7 else None
    if eu is not None:
        assert np.allclose(ctx.embedding_cost(2, a, b), eu, rtol=1e-4)
    assert np.allclose(ctx.embedding_cost(1, a, b), a64 @ b64.T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,m,d,case", [(1, 1, 8, "plain"), (300, 130, 100, "plain"), (861, 450, 256, "plain"), (1024, 512, 256, "plain"),
                                        (130, 70, 64, "coincident"), (200, 90, 32, "nan"), (40, 30, 16, "prox_one"), (40, 30, 16, "other_mode")])
def test_gated_appearance_distances(ctx, orc, n, m, d, case):
    """BoT-SORT's appearance term (botsort.cpp:433-466): the reference masks every pair with iou_distance > proximity_thresh to 1 (:439-447),
    so mot_cosine_cost_gated evaluates the other pairs only — their entries equal the oracle's embedding_distance bit for bit, every other
    entry of the output is left as the caller had it. Edge cases: strips whose pair list overflows (thousands of coincident boxes), NaN boxes
    (their IoU is +0: masked), a threshold that masks nothing, a cost mode that reads the whole matrix."""
    r = np.random.default_rng(n * 31 + m * 7 + d)
    a = r.standard_normal((n, d)).astype(np.float32)
    b = r.standard_normal((m, d)).astype(np.float32)
    ta, tb = boxes(r, n, (900, 600)), boxes(r, m, (900, 600))
    k = min(n, m)
    tb[:k] = ta[r.permutation(n)[:k]] + r.normal(0, 3, (k, 4)).astype(np.float32)
    prox, mode = 0.5, L.COST_BOTSORT
    if case == "coincident":
        ta[:] = ta[0]
        tb[:] = ta[0]
    elif case == "nan":
        ta[3, 0] = np.nan
        tb[5, 2] = np.nan
        tb[6] = ta[3]
    elif case == "prox_one":
        prox = 1.0
    elif case == "other_mode":
        mode = L.COST_IOU_DIST
    full = orc.embedding_distance(0, a, b)
    dist = orc.iou_distance(ta, tb)
    passes = ~(dist > prox) if case not in ("prox_one", "other_mode") else np.ones((n, m), bool)
    sentinel = np.full((n, m), -7.0, np.float32)
    g = ctx.cosine_cost_gated(a, b, ta, tb, prox, sentinel, cost_mode=mode)
    assert np.array_equal(g[passes], full[passes]), np.abs(g[passes] - full[passes]).max()
    assert np.all(g[~passes] == -7.0)
    assert passes.sum() >= (1 if case != "nan" else 0)
    if case == "coincident":
        assert passes.all()
    if case == "plain" and n > 1:
        assert 0 < passes.sum() < n * m / 8


def _gating_states(orc, kind, n, seed):
    """n Kalman states a few frames old (initiate, then predict/update rounds with jittered measurements) + their measurements"""
    r = np.random.default_rng(seed)
    cx, cy = r.uniform(100, 1800, n), r.uniform(100, 900, n)
    h = r.uniform(30, 300, n)
    third = r.uniform(0.3, 0.6, n) if kind == L.KF_XYAH else r.uniform(0.3, 0.6, n) * h  # aspect ratio / width
    z = np.stack([cx, cy, third, h], 1).astype(np.float32)
    mean, cov = orc.kf_initiate(kind, z)
    for _ in range(3):
        mean, cov = orc.kf_predict(kind, mean, cov)
        z = (z + r.normal(0, 1, z.shape) * np.array([2.0, 2.0, 0.01 if kind == L.KF_XYAH else 1.0, 2.0])).astype(np.float32)
        mean, cov = orc.kf_update(kind, mean, cov, z)
    mean, cov = orc.kf_predict(kind, mean, cov)
    return mean, cov, z


@pytest.mark.parametrize("kind", [L.KF_XYAH, L.KF_XYWH])
@pytest.mark.parametrize("n,m", [(1, 1), (7, 300), (130, 257), (600, 513)])
def test_gating_distance_and_blends(ctx, orc, kind, n, m):
    """gating_distance (kalman_filter.cpp:148-176 / xywh_kf.hpp:140-176), utils::fuse_motion (matching.hpp:60-94) and StrongSORT's
    gate_cost_matrix (strongsort.cpp:449-492): bit-identical to the restatement; the distances also against float64 linear algebra
    of the same formula within 1e-4 relative."""
    mean, cov, z = _gating_states(orc, kind, n, 100 * n + m + kind)
    r = np.random.default_rng(n * m + kind)
    # measurements: every track's own (jittered) measurement is in the set -> both sides of every threshold occur
    meas = np.concatenate([z + r.normal(0, 1, z.shape).astype(np.float32) * np.array([3, 3, 0.02, 3], np.float32),
                           z[r.integers(0, n, m)] + r.normal(0, 40, (m, 4)).astype(np.float32) * np.array([1, 1, 0.001, 0.2], np.float32)])[:m]
    meas = meas.astype(np.float32)
    cost = r.uniform(0, 1, (n, meas.shape[0])).astype(np.float32)
    for pos in (False, True):
        for metric in ((0, 1) if kind == L.KF_XYAH else (0,)):
            g = ctx.gate_cost(kind, 0, mean, cov, meas, only_position=pos, metric=metric)
            o = orc.gate_cost(kind, 0, mean, cov, meas, only_position=pos, metric=metric)
            assert np.array_equal(g, o), (pos, metric, np.abs(g - o).max())
        dim = 2 if pos else 4
        S = cov.astype(np.float64)[:, :4, :4].copy()
        hh = mean[:, 3].astype(np.float64)
        for i in range(n):
            sd = np.full(4, hh[i] / 20.0)
            if kind == L.KF_XYAH:
                sd[2] = 0.1
            S[i] += np.diag(sd ** 2)
        d = meas.astype(np.float64)[None, :, :dim] - mean.astype(np.float64)[:, None, :dim]
        if kind == L.KF_XYAH:  # |S_sub^-1 d|^2 (the full LLT solve, kalman_filter.cpp:169-170)
            zz = np.einsum("nab,nmb->nma", np.linalg.inv(S[:, :dim, :dim]), d)
            ref = (zz ** 2).sum(-1)
        else:  # d^T (S^-1)[:dim,:dim] d
            ref = np.einsum("nma,nab,nmb->nm", d, np.linalg.inv(S)[:, :dim, :dim], d)
        g = ctx.gate_cost(kind, 0, mean, cov, meas, only_position=pos)
        assert np.allclose(g, ref, rtol=1e-3, atol=1e-6), np.abs(g / np.maximum(ref, 1e-30) - 1).max()
        thr = 5.9915 if pos else 9.4877
        if n * m > 100:
            assert (g > thr).any() and (g <= thr).any()
        for mode, lam, gc in ((1, 0.98, 0.0), (2, 0.995, 1e5)):
            a = ctx.gate_cost(kind, mode, mean, cov, meas, cost, only_position=pos, lam=lam, gated_cost=gc)
            b = orc.gate_cost(kind, mode, mean, cov, meas, cost, only_position=pos, lam=lam, gated_cost=gc)
            assert np.array_equal(a, b), (mode, pos)
            if mode == 1:
                assert np.array_equal(np.isinf(a), g > np.float32(thr))


def test_gating_failed_factorisation_falls_back(ctx, orc):
    """a covariance that is not positive definite: the XYAH distance becomes the plain squared norm (kalman_filter.cpp:161-167)"""
    mean = np.array([[10, 20, 0.5, 100, 0, 0, 0, 0]], np.float32)
    cov = np.zeros((1, 8, 8), np.float32)
    cov[0, 0, 0] = -1000.0
    meas = np.array([[13, 24, 0.5, 100], [10, 20, 0.5, 100]], np.float32)
    g = ctx.gate_cost(L.KF_XYAH, 0, mean, cov, meas)
    assert np.array_equal(g, orc.gate_cost(L.KF_XYAH, 0, mean, cov, meas))
    assert g[0, 0] == 25.0 and g[0, 1] == 0.0


@pytest.mark.parametrize("n,m", [(1, 1), (33, 65), (300, 500)])
def test_fuse_iou(ctx, orc, n, m):
    """utils::fuse_iou (matching.cpp:109-128): 1 - (1 - reid) * (1 + iou) / 2 in the reference's operation order"""
    r = np.random.default_rng(n + m)
    a, b = boxes(r, n), boxes(r, m)
    b[: min(n, m)] = a[: min(n, m)] + r.normal(0, 3, (min(n, m), 4)).astype(np.float32)
    reid = r.uniform(0, 1, (n, m)).astype(np.float32)
    g, o = ctx.fuse_iou(reid, a, b), orc.fuse_iou(reid, a, b)
    assert np.array_equal(g, o)
    iou = orc.iou_batch(a, b).astype(np.float64)
    assert np.allclose(g, 1 - (1 - reid.astype(np.float64)) * (1 + iou) / 2, atol=1e-6)


@pytest.mark.parametrize("n", [1, 64, 333, 5000])
def test_nsa_kalman_update(ctx, orc, n):
    """BaseKalmanFilter::update(mean, cov, z, confidence) (kalman_filter.cpp:60-112): the measurement noise is scaled by
    (1 - confidence) per detection — StrongSORT's Track::update (strongsort.cpp:153). Bit-identical to the restatement, and
    confidence 0 is the plain update."""
    mean, cov, z = _gating_states(orc, L.KF_XYAH, n, 7 * n)
    r = np.random.default_rng(n)
    conf = r.uniform(0.0, 1.0, n).astype(np.float32)
    conf[::7] = 0.0
    gm, gc = ctx.kf_update_conf(mean, cov, z, conf)
    om, oc = orc.kf_update_conf(mean, cov, z, conf)
    assert np.array_equal(gm, om) and np.array_equal(gc, oc)
    pm, pc = orc.kf_update(L.KF_XYAH, mean, cov, z)
    assert np.array_equal(gm[::7], pm[::7]) and np.array_equal(gc[::7], pc[::7])
    assert n < 8 or not np.array_equal(gm, pm)


def _blocks_of(cov):
    """8 x 8 covariances -> block form [n, 4, 4] = {P(c,c), P(c,c+4), P(c+4,c), P(c+4,c+4)} (mot_kf_task.cov_blocks)"""
    n = cov.shape[0]
    b = np.zeros((n, 4, 4), np.float32)
    for c in range(4):
        b[:, c, 0], b[:, c, 1], b[:, c, 2], b[:, c, 3] = cov[:, c, c], cov[:, c, c + 4], cov[:, c + 4, c], cov[:, c + 4, c + 4]
    return b


def _dense_of(blocks):
    n = blocks.shape[0]
    cov = np.zeros((n, 8, 8), np.float32)
    for c in range(4):
        cov[:, c, c], cov[:, c, c + 4], cov[:, c + 4, c], cov[:, c + 4, c + 4] = blocks[:, c, 0], blocks[:, c, 1], blocks[:, c, 2], blocks[:, c, 3]
    return cov


def test_kalman_block_form_is_the_dense_filter(ctx, orc):
    """Round 6: ByteTrack's lifecycle keeps covariances as four 2 x 2 blocks (kf_update_blocks_kernel: 96 bytes per track instead of 576). A track that
    is initiated and then only predicted / updated has exact zeros everywhere else, so the block arithmetic IS the dense arithmetic term by term:
    700 tracks through eight predict + update rounds against the oracle's dense KalmanFilterXYAH (kalman_filter.cpp:44-112), every entry bit for bit,
    with and without the predict folded into the update (MOT_KF_PREDICT_FIRST, MOT_KF_ZERO_V7) — and the structural zeros of the oracle's dense
    covariance checked to BE zeros, which is what the block form rests on."""
    r = np.random.default_rng(11)
    n = 700
    b = boxes(r, n)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    z = np.stack([b[:, 0] + w / 2, b[:, 1] + h / 2, w / h, h], 1).astype(np.float32)
    mo, co = orc.kf_initiate(L.KF_XYAH, z)
    mg, bg = mo.copy(), _blocks_of(co)
    mask = np.ones((8, 8), bool)
    for c in range(4):
        mask[c, c] = mask[c, c + 4] = mask[c + 4, c] = mask[c + 4, c + 4] = False
    scale = np.array([2, 2, 0.01, 2], np.float32)
    for step in range(8):
        zero_v7 = (r.uniform(0, 1, n) < 0.3)
        flags = (8 | np.where(zero_v7, 1, 0)).astype(np.uint8)  # MOT_KF_PREDICT_FIRST | MOT_KF_ZERO_V7 for a third of the tracks
        mp = mo.copy()
        mp[zero_v7, 7] = 0.0  # bytetrack.cpp:108-110: the height's velocity of a non-tracked state is zeroed before the prediction
        mp, cp = orc.kf_predict(L.KF_XYAH, mp, co)
        zz = (mp[:, :4] + r.normal(0, 1, (n, 4)).astype(np.float32) * scale).astype(np.float32)
        mo, co = orc.kf_update(L.KF_XYAH, mp, cp, zz)
        assert not np.any(co[:, mask] != 0.0), "the oracle's covariance has a non-zero where the block form keeps a structural zero"
        mg, bg, dense, _cd = ctx.kf_update_blocks(mg, bg, zz, flags)
        assert not dense.any(), (step, int(dense.sum()))
        assert np.array_equal(mg, mo), (step, np.abs(mg - mo).max())
        assert np.array_equal(_dense_of(bg), co), (step, np.abs(_dense_of(bg) - co).max())
    # without the folded predict (ByteTrack's unconfirmed tracks are updated from their stored state)
    zz = (mo[:, :4] + r.normal(0, 1, (n, 4)).astype(np.float32) * scale).astype(np.float32)
    mo2, co2 = orc.kf_update(L.KF_XYAH, mo, co, zz)
    mg2, bg2, dense, _cd = ctx.kf_update_blocks(mg, bg, zz, np.zeros(n, np.uint8))
    assert not dense.any() and np.array_equal(mg2, mo2) and np.array_equal(_dense_of(bg2), co2)


def test_kalman_block_form_hands_degenerate_tracks_to_the_dense_filter(ctx, orc):
    """What the block form cannot represent leaves it untouched: a height of zero (innovation variance 0: the reference's Cholesky fails and its
    pivoted-LU inverse spreads inf / NaN over the whole state), a NaN or infinite entry, a measurement of 1e38 (the innovation overflows: 0 * inf in the
    dense filter). Those tracks come back flagged, with the 8 x 8 covariance the dense kernel computed — equal to the oracle's with NaN == NaN — while
    the ordinary tracks in between are updated in block form, bit for bit."""
    r = np.random.default_rng(5)
    n = 64
    b = boxes(r, n)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    z = np.stack([b[:, 0] + w / 2, b[:, 1] + h / 2, w / h, h], 1).astype(np.float32)
    z[3, 3] = 0.0     # height 0 at birth: every variance that scales with it is 0
    mo, co = orc.kf_initiate(L.KF_XYAH, z)
    mo, co = orc.kf_predict(L.KF_XYAH, mo, co)
    mo, co = orc.kf_update(L.KF_XYAH, mo, co, z)  # (one ordinary round so that the off-diagonal block entries are populated; track 3 is already degenerate)
    finite_before = np.isfinite(co).all(axis=(1, 2)) & np.isfinite(mo).all(axis=1)
    mg, bg = mo.copy(), _blocks_of(np.nan_to_num(co, nan=0.0, posinf=0.0, neginf=0.0))
    weird = [3, 10, 20, 30]
    mg[10, 5] = np.inf; mo[10, 5] = np.inf            # an infinite velocity
    bg[20, 1, 2] = np.nan; co[20, 5, 1] = np.nan      # a NaN in a block
    zz = (mo[:, :4] + r.normal(0, 1, (n, 4)).astype(np.float32)).astype(np.float32)
    zz[30, 0] = 1.0e38                                 # the innovation is finite, its product with the gain is not
    mg[3], bg[3] = np.nan_to_num(mo[3]), 0.0; mo[3], co[3] = mg[3], 0.0   # track 3: a clean all-zero covariance with height 0 -> S = 0
    mg[3, 3] = 0.0; mo[3, 3] = 0.0
    flags = np.full(n, 8, np.uint8)
    mp, cp = orc.kf_predict(L.KF_XYAH, mo, co)
    mo2, co2 = orc.kf_update(L.KF_XYAH, mp, cp, zz)
    mg2, bg2, dense, cd = ctx.kf_update_blocks(mg, bg, zz, flags)
    assert set(np.nonzero(dense)[0]) >= {3, 10, 20}, np.nonzero(dense)[0]
    ok = dense == 0
    assert ok.sum() >= n - 6
    assert np.array_equal(mg2[ok], mo2[ok]) and np.array_equal(_dense_of(bg2)[ok], co2[ok])
    # the tracks handed on: exactly what the dense kernel (kf_update8_kernel through mot_kf_predict / mot_kf_update) makes of the same states ...
    md, cdn = ctx.kf_apply(L.KF_XYAH, 1, mo, co)
    md, cdn = ctx.kf_apply(L.KF_XYAH, 2, md, cdn, meas=zz)
    bad = dense != 0
    assert np.array_equal(mg2[bad], md[bad], equal_nan=True) and np.array_equal(cd[bad], cdn[bad], equal_nan=True)
    # ... which is the oracle's state for the zero innovation variance and for the NaN entry. (An INFINITE mean entry is outside what any of the
    # kernels reproduces: they apply F = I + shift and H = [I 0] structurally, the reference multiplies by their zeros — 0 * inf = NaN; that has been so
    # since round 1 and is why "finite" is part of the block form's test.)
    for t in (3, 20):
        assert np.array_equal(mg2[t], mo2[t], equal_nan=True) and np.array_equal(cd[t], co2[t], equal_nan=True), t
