"""Pins the CPU oracle against every known answer the reference's own tests hold for the hot path
(SURVEY.md §8c): tests/test_matching.cpp:15-110, tests/test_iou.cpp:29-75,101-108,
tests/test_kalman_filter.cpp:19-83, tests/test_sort.cpp:36-148, tests/test_trackers.cpp /
test_bytetrack.cpp shape checks — plus an independent optimum check of the assignment with
scipy.optimize.linear_sum_assignment on the explicitly extended (n+m)^2 matrix."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from tests import orclib


def matches(x):
    return {(i, int(j)) for i, j in enumerate(x) if j >= 0}


# ---- tests/test_matching.cpp ------------------------------------------------------------
def test_lap_empty(orc):
    x, y = orc.linear_assignment(np.zeros((0, 0), np.float32), 0.5)
    assert len(x) == 0 and len(y) == 0


def test_lap_single_match(orc):  # :24-35
    x, y = orc.linear_assignment([[0.1]], 0.5)
    assert list(x) == [0] and list(y) == [0]


def test_lap_above_threshold(orc):  # :37-46
    x, y = orc.linear_assignment([[0.9]], 0.5)
    assert list(x) == [-1] and list(y) == [-1]


def test_lap_diag3(orc):  # :48-70
    c = np.full((3, 3), 0.9, np.float32)
    np.fill_diagonal(c, 0.1)
    x, y = orc.linear_assignment(c, 0.5)
    assert matches(x) == {(0, 0), (1, 1), (2, 2)}


def test_lap_more_tracks(orc):  # :72-83
    x, y = orc.linear_assignment([[0.1, 0.9], [0.9, 0.1], [0.9, 0.9]], 0.5)
    assert matches(x) == {(0, 0), (1, 1)} and list(np.where(x < 0)[0]) == [2] and (y >= 0).all()


def test_lap_more_dets(orc):  # :85-96
    x, y = orc.linear_assignment([[0.1, 0.9, 0.9], [0.9, 0.1, 0.9]], 0.5)
    assert matches(x) == {(0, 0), (1, 1)} and list(np.where(y < 0)[0]) == [2]


def test_lap_optimal_2x2(orc):  # :98-110
    x, _ = orc.linear_assignment([[0.1, 0.2], [0.3, 0.1]], 0.5)
    assert matches(x) == {(0, 0), (1, 1)}


# ---- independent optimum check (scipy on the explicit extension, lap_solver.hpp:299-315) ----
def extended(cost, thresh):
    n, m = cost.shape
    e = np.full((n + m, n + m), thresh / 2.0, np.float64)
    e[:n, :m] = cost.astype(np.float64)
    e[n:, m:] = 0.0
    return e


@pytest.mark.parametrize("n,m,kind,seed", [(7, 5, "dense", 0), (40, 60, "dense", 1), (64, 64, "sparse", 2),
                                           (130, 90, "sparse", 3), (33, 33, "neg", 4), (1, 9, "dense", 5),
                                           (200, 120, "sparse", 6)])
def test_lap_total_cost_is_optimal(orc, n, m, kind, seed):
    r = np.random.default_rng(seed)
    if kind == "dense":
        c, th = r.uniform(0, 1, (n, m)).astype(np.float32), 0.8
    elif kind == "neg":  # OC-SORT style negative costs / threshold (ocsort.cpp:700-701)
        c, th = (-r.uniform(0, 1, (n, m))).astype(np.float32), -0.3
    else:  # IoU-like: one good candidate per row, few extra overlaps, rest exactly 1
        c = np.ones((n, m), np.float32)
        for i in range(n):
            if r.uniform() < 0.8:
                c[i, r.integers(m)] = r.uniform(0.05, 0.6)
        extra = r.uniform(0, 1, (n, m)) < 0.02
        c[extra] = r.uniform(0.2, 0.95, extra.sum()).astype(np.float32)
        th = 0.8
    x, y = orc.linear_assignment(c, th)
    # consistency of x and y
    for i, j in enumerate(x):
        if j >= 0:
            assert y[j] == i
    assert (x >= 0).sum() == (y >= 0).sum()
    e = extended(c, th)
    ri, ci = linear_sum_assignment(e)
    best = e[ri, ci].sum()
    k = int((x >= 0).sum())
    mine = sum(float(c[i, j]) for i, j in enumerate(x) if j >= 0) + (th / 2.0) * ((n - k) + (m - k))
    assert mine == pytest.approx(best, rel=1e-12, abs=1e-9)


# ---- tests/test_iou.cpp -------------------------------------------------------------------
B1, B2, B3 = [[0, 0, 100, 100]], [[50, 50, 150, 150]], [[200, 200, 300, 300]]


def test_iou_identical_and_disjoint(orc):  # :29-37
    assert orc.iou_batch(B1, B1)[0, 0] == 1.0
    assert orc.iou_batch(B1, B3)[0, 0] == 0.0


def test_iou_overlap(orc):  # :39-46
    v = orc.iou_batch(B1, B2)[0, 0]
    assert abs(v - 0.143) < 0.01 and v == np.float32(2500.0) / np.float32(17500.0)


def test_iou_batch_2x2(orc):  # :48-67
    m = orc.iou_batch([[0, 0, 100, 100], [50, 50, 150, 150]], [[0, 0, 100, 100], [200, 200, 300, 300]])
    assert m.shape == (2, 2) and m[0, 0] == 1.0 and m[0, 1] == 0.0


def test_iou_empty(orc):  # :69-75
    assert orc.iou_batch(np.zeros((0, 4), np.float32), B1).shape == (0, 1)


def test_iou_distance_and_fuse(orc):  # matching.cpp:62-65,130-143
    d = orc.iou_distance(B1 + B2, B1 + B3)
    assert d[0, 0] == 0.0 and d[0, 1] == 1.0
    f = orc.fuse_score(d, [0.5, 0.9])
    assert f[0, 0] == np.float32(1.0) - np.float32(1.0) * np.float32(0.5) and f[0, 1] == 1.0


# ---- tests/test_kalman_filter.cpp (XYSR) ----------------------------------------------------
def xysr_default():
    mean = np.zeros((1, 7), np.float32)
    cov = np.diag([10, 10, 10, 10, 1000, 1000, 1000]).astype(np.float32)[None]
    return mean, cov


def test_xysr_predict_adds_velocity(orc):  # :35-45
    mean, cov = xysr_default()
    mean[0] = [100, 100, 1000, 0.5, 10, 10, 0]
    m2, _ = orc.kf_predict(orclib.KF_XYSR, mean, cov)
    assert m2[0, 0] == 110.0 and m2[0, 1] == 110.0


def test_xysr_initial_matrices(orc):  # :19-33 and xysr_kf.cpp:52-65 via one predict from x = 0
    mean, cov = xysr_default()
    _, c2 = orc.kf_predict(orclib.KF_XYSR, mean, cov)
    # P' = F P F^T + Q with P = diag(10,10,10,10,1000,1000,1000), Q = diag(1,1,1,1,.01,.01,.0001)
    assert c2[0, 0, 0] == np.float32(10 + 1000 + 1)
    assert c2[0, 0, 4] == 1000.0 and c2[0, 4, 0] == 1000.0
    assert c2[0, 3, 3] == 11.0
    assert c2[0, 4, 4] == np.float32(1000.0) + np.float32(0.01)
    assert c2[0, 6, 6] == np.float32(1000.0) + np.float32(0.0001)


def test_xysr_update_moves_towards_measurement(orc):  # :47-58
    mean, cov = xysr_default()
    mean[0] = [100, 100, 1000, 0.5, 0, 0, 0]
    m2, c2 = orc.kf_update(orclib.KF_XYSR, mean, cov, [[110, 110, 1100, 0.5]])
    assert 100 < m2[0, 0] < 110
    # closed form for the decoupled x channel: K = P/(P+R) = 10/11
    assert m2[0, 0] == pytest.approx(100 + 10 * 10 / 11, rel=1e-6)
    assert c2[0, 0, 0] == pytest.approx(10 / 11, rel=1e-5)
    assert np.allclose(c2[0], c2[0].T, rtol=0, atol=1e-4)


# ---- 8-state filters: closed forms (the reference has no tests for them, §4) ----------------
@pytest.mark.parametrize("kind", [orclib.KF_XYAH, orclib.KF_XYWH])
def test_kf8_initiate_predict_update(orc, kind):
    z = np.array([[320.0, 240.0, 0.5 if kind == orclib.KF_XYAH else 60.0, 120.0]], np.float32)
    mean, cov = orc.kf_initiate(kind, z)
    h = np.float32(120.0)
    wp, wv = np.float32(1.0) / np.float32(20.0), np.float32(1.0) / np.float32(160.0)
    assert (mean[0, :4] == z[0]).all() and (mean[0, 4:] == 0).all()
    sd0 = np.float32(2.0) * wp * h
    assert cov[0, 0, 0] == sd0 * sd0
    assert cov[0, 4, 4] == (np.float32(10.0) * wv * h) ** 2
    if kind == orclib.KF_XYAH:
        assert cov[0, 2, 2] == np.float32(1e-2) ** 2 and cov[0, 6, 6] == np.float32(1e-5) ** 2
    m2, c2 = orc.kf_predict(kind, mean, cov)
    assert (m2[0] == mean[0]).all()  # zero velocity
    assert c2[0, 0, 0] == cov[0, 0, 0] + cov[0, 4, 4] + (wp * h) ** 2
    assert c2[0, 0, 4] == cov[0, 4, 4]
    m3, c3 = orc.kf_update(kind, m2, c2, z + np.float32(1.0))
    assert (m3[0, :2] > z[0, :2]).all() and (m3[0, :2] < z[0, :2] + 1).all()
    assert np.allclose(c3[0], c3[0].T, atol=1e-3)
    assert (np.diag(c3[0]) > 0).all() and c3[0, 0, 0] < c2[0, 0, 0]
    # float64 textbook Kalman update as an independent numerical check (1e-4 relative)
    P = c2[0].astype(np.float64)
    Hm = np.eye(4, 8)
    if kind == orclib.KF_XYAH:
        R = np.diag([(0.05 * 120) ** 2, (0.05 * 120) ** 2, 1e-2, (0.05 * 120) ** 2])
    else:
        R = np.eye(4) * (0.05 * 120) ** 2
    S = Hm @ P @ Hm.T + R
    K = P @ Hm.T @ np.linalg.inv(S)
    mref = m2[0].astype(np.float64) + K @ ((z[0] + 1).astype(np.float64) - m2[0, :4])
    Pref = P - K @ S @ K.T
    assert np.allclose(m3[0], mref, rtol=1e-5, atol=1e-5)
    assert np.allclose(c3[0], Pref, rtol=1e-4, atol=1e-4)


def test_xysr_update_vs_float64(orc):
    r = np.random.default_rng(0)
    mean, cov = xysr_default()
    mean[0] = [500, 300, 6000, 0.45, 2, -1, 10]
    for _ in range(5):
        mean, cov = orc.kf_predict(orclib.KF_XYSR, mean, cov)
        z = mean[:, :4] + r.normal(0, 1, (1, 4)).astype(np.float32) * [1, 1, 50, 0.01]
        P = cov[0].astype(np.float64)
        Hm, R = np.eye(4, 7), np.diag([1.0, 1, 10, 10])
        S = Hm @ P @ Hm.T + R
        K = P @ Hm.T @ np.linalg.inv(S)
        mref = mean[0] + K @ (z[0].astype(np.float64) - mean[0, :4])
        IKH = np.eye(7) - K @ Hm
        Pref = IKH @ P @ IKH.T + K @ R @ K.T
        mean, cov = orc.kf_update(orclib.KF_XYSR, mean, cov, z)
        assert np.allclose(mean[0], mref, rtol=1e-5, atol=1e-4)
        assert np.allclose(cov[0], Pref, rtol=1e-4, atol=1e-3)


# ---- ops.hpp edge semantics -------------------------------------------------------------------
def test_box_conversions(orc):
    out = orc.box_convert(0, [[10, 20, 50, 100]])  # xyxy2xysr
    assert list(out[0]) == [30.0, 60.0, 3200.0, 0.5]
    back = orc.box_convert(1, out)  # xysr2xyxy
    assert np.allclose(back[0], [10, 20, 50, 100])
    assert orc.box_convert(0, [[0, 0, 10, 0]])[0, 3] == 0.0  # r = 0 when h <= 1e-6 (ops.hpp:195)
    assert np.isnan(orc.box_convert(1, [[0, 0, -5, 1]])).any()  # sqrt(s*r<0) -> NaN (ops.hpp:204)
    assert orc.box_convert(5, [[0, 0, 10, 0]])[0, 2] == 0.0  # a = 0 when h <= 0 (ops.hpp:83)


# ---- tests/test_sort.cpp -----------------------------------------------------------------------
SINGLE = np.array([[100, 100, 200, 200, 0.9, 0]], np.float32)
EMPTY = np.zeros((0, 6), np.float32)


def test_sort_single_detection(orc):  # :36-47
    t = orc.tracker(orclib.SORT, [0.3, 1, 50, 1])
    out = t.update(SINGLE)
    assert out.shape == (1, 8) and out[0, 2] > out[0, 0] and out[0, 3] > out[0, 1]


def test_sort_multi_frame_id(orc):  # :49-67
    t = orc.tracker(orclib.SORT, [0.3, 3, 50, 1])
    t.update(SINGLE)
    t.update(SINGLE)
    out = t.update(np.array([[110, 110, 210, 210, 0.9, 0]], np.float32))
    assert out.shape[0] == 1 and int(out[0, 4]) == 1


def test_sort_track_deletion(orc):  # :69-84
    t = orc.tracker(orclib.SORT, [0.3, 2, 50, 1])
    t.update(SINGLE)
    t.update(EMPTY)
    assert t.update(EMPTY).shape[0] == 0


def test_sort_confidence_filter(orc):  # :108-122
    t = orc.tracker(orclib.SORT, [0.5, 3, 50, 1])
    out = t.update(np.array([[100, 100, 200, 200, 0.3, 0], [300, 300, 400, 400, 0.7, 0]], np.float32))
    assert out.shape[0] <= 1


def test_sort_id_survives_missed_frame(orc):  # :128-148
    t = orc.tracker(orclib.SORT, [0.3, 5, 50, 1])
    for i in range(5):
        t.update(np.array([[100 + i * 10, 100 + i * 10, 200 + i * 10, 200 + i * 10, 0.9, 0]], np.float32))
    t.update(EMPTY)
    out = t.update(np.array([[160, 160, 260, 260, 0.9, 0]], np.float32))
    assert out.shape[0] == 1 and int(out[0, 4]) == 1


# ---- tests/test_bytetrack.cpp / test_trackers.cpp shape and persistence checks -------------------
MULTI = np.array([[100, 100, 200, 200, 0.9, 0], [300, 300, 400, 400, 0.8, 0], [500, 100, 600, 200, 0.7, 1]],
                 np.float32)


@pytest.mark.parametrize("kind", [orclib.BYTETRACK, orclib.OCSORT, orclib.BOTSORT])
def test_tracker_output_validity(orc, kind):  # test_bytetrack.cpp:125-149, test_trackers.cpp:39-107
    t = orc.tracker(kind)
    ids = []
    for _ in range(3):
        out = t.update(MULTI)
        assert out.shape[1] == 8 if out.size else True
        for row in out:
            assert row[0] < row[2] and row[1] < row[3] and row[4] > 0 and 0 <= row[5] <= 1
        ids.append(set(out[:, 4].astype(int)))
    assert ids[1] & ids[2]  # some ids persist over identical frames
    assert t.update(EMPTY).shape[0] == 0 or kind != orclib.BOTSORT


def test_empty_dets_give_no_rows(orc):  # test_trackers.cpp:100-107
    for kind in (orclib.BYTETRACK, orclib.OCSORT, orclib.BOTSORT, orclib.SORT):
        assert orc.tracker(kind).update(EMPTY).shape[0] == 0


# ---- association measures (include/motcpp/utils/iou.hpp:122-414) ----
ASSO = {"iou": 0, "hmiou": 1, "giou": 2, "ciou": 3, "diou": 4, "centroid": 5}
AB1 = np.array([[0, 0, 100, 100]], np.float32)      # tests/test_iou.cpp:14-22
AB2 = np.array([[50, 50, 150, 150]], np.float32)
AB3 = np.array([[200, 200, 300, 300]], np.float32)


def test_asso_reference_cases(orc):  # tests/test_iou.cpp:75-116
    for k in ("giou", "diou", "ciou"):
        v = orc.asso_batch(ASSO[k], AB1, AB2)[0, 0]
        assert 0.0 <= v <= 1.0
    c = orc.asso_batch(ASSO["centroid"], AB1, AB3, (640, 480))[0, 0]
    assert 0.0 < c < 1.0
    assert abs(orc.asso_batch(ASSO["iou"], AB1, AB2)[0, 0] - 0.143) < 0.01  # AssociationFunctionIoU
    # closed forms of the same cases
    # reference quirk: giou_batch recovers the "intersection" as iou*(a1+a2)/(iou+1e-10) = a1+a2 (iou.hpp:182), so its union is ~0
    # and the value is (iou - 1 + 1)/2 = iou/2 for overlapping boxes — restated as written, not as the textbook GIoU
    assert abs(orc.asso_batch(ASSO["giou"], AB1, AB2)[0, 0] - (1 / 7) / 2) < 1e-6
    assert abs(orc.asso_batch(ASSO["diou"], AB1, AB2)[0, 0] - (1 / 7 - 5000 / 45000 + 1) / 2) < 1e-6
    assert abs(orc.asso_batch(ASSO["ciou"], AB1, AB2)[0, 0] - (1 / 7 - 5000 / 45000 + 1) / 2) < 1e-6  # equal aspect: v = 0
    assert abs(orc.asso_batch(ASSO["hmiou"], AB1, AB2)[0, 0] - (1 / 7) * (50 / 150)) < 1e-6
    assert abs(c - (1 - np.sqrt(80000.0) / 800.0)) < 1e-6
    assert orc.asso_batch(ASSO["giou"], AB1[:0], AB2).shape == (0, 1)  # empty side -> Zero(N, M)


def _asso_numpy(kind, A, B, frame=(1920, 1080)):
    """float32 numpy restatement (independent second opinion; same operation order, elementwise)."""
    f = np.float32
    a = [A[:, k][:, None].astype(f) for k in range(4)]
    b = [B[:, k][None, :].astype(f) for k in range(4)]
    area1, area2 = (a[2] - a[0]) * (a[3] - a[1]), (b[2] - b[0]) * (b[3] - b[1])
    w = np.maximum(f(0), np.minimum(a[2], b[2]) - np.maximum(a[0], b[0]))
    h = np.maximum(f(0), np.minimum(a[3], b[3]) - np.maximum(a[1], b[1]))
    inter = w * h
    uni = area1 + area2 - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(uni > 0, inter / uni, f(0)).astype(f)
    if kind == 0:
        return iou
    if kind == 1:
        ih = np.maximum(np.minimum(a[3], b[3]) - np.maximum(a[1], b[1]), f(0))
        uh = np.maximum(np.maximum(a[3], b[3]) - np.minimum(a[1], b[1]), f(1e-10))
        return iou * (ih / uh)
    ox = np.maximum(a[2], b[2]) - np.minimum(a[0], b[0])
    oy = np.maximum(a[3], b[3]) - np.minimum(a[1], b[1])
    if kind == 2:
        enc = ox * oy
        inter2 = iou * (area1 + area2) / (iou + f(1e-10))
        un = area1 + area2 - inter2
        return ((iou - (enc - un) / (enc + f(1e-10))) + f(1)) / f(2)
    dx = (a[0] + a[2]) / f(2) - (b[0] + b[2]) / f(2)
    dy = (a[1] + a[3]) / f(2) - (b[1] + b[3]) / f(2)
    if kind == 5:
        norm = f(np.sqrt(float(frame[0] * frame[0] + frame[1] * frame[1])))
        return f(1) - np.sqrt(dx * dx + dy * dy) / norm
    inner = dx * dx + dy * dy
    if kind == 4:
        return ((iou - inner / ((ox * ox + oy * oy) + f(1e-10))) + f(1)) / f(2)
    eps = f(1e-7)
    outer = ox * ox + oy * oy + eps
    w1, h1, w2, h2 = a[2] - a[0], a[3] - a[1], b[2] - b[0], b[3] - b[1]
    ad = np.arctan((w2 / (h2 + eps)).astype(np.float64)).astype(f) - np.arctan((w1 / (h1 + eps)).astype(np.float64)).astype(f)
    v = (f(4) / f(np.pi * np.pi)) * (ad * ad)
    alpha = v / ((f(1) - iou) + v + eps)
    return ((iou - inner / outer + alpha * v) + f(1)) / f(2)


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5])
def test_asso_matches_float32_numpy(orc, kind):
    r = np.random.default_rng(kind)
    cx, cy, w = r.uniform(0, 600, 90), r.uniform(0, 400, 90), r.uniform(20, 90, 90)
    h = w * r.uniform(0.5, 2.6, 90)
    A = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    B = (A[r.permutation(90)[:70]] + r.normal(0, 4, (70, 4))).astype(np.float32)
    got, ref = orc.asso_batch(kind, A, B, (640, 480)), _asso_numpy(kind, A, B, (640, 480))
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_ocsort_runs_with_every_association_measure(orc):
    from motcpp_amd.synth import SynthStream
    outs = {}
    for name, k in ASSO.items():
        t = orc.tracker(2, [0.2, 30, 50, 3, 0.3, 0.1, 3, 0.2, 1, 0.01, 0.0001, k, 1920, 1080])
        s = SynthStream(60, 40, 5)
        n = 0
        for _ in range(25):
            d, _e = s.next_frame()
            o = t.update(d)
            n += o.shape[0]
            assert np.isfinite(o).all()
        assert n > 0
        outs[name] = n
    assert len(set(outs.values())) > 1  # the measures do change what gets associated


# ---- camera-motion compensation of track states: botsort.cpp:60-91,317-324; xysr_kf.cpp:114-141 ----
def _xywh_state(orc, box):
    x1, y1, x2, y2 = box
    z = np.array([[(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1]], np.float32)
    m, c = orc.kf_initiate(orclib.KF_XYWH, z)
    m[:, 4:] = [1.5, -0.5, 0.25, 0.125]
    return m, c


def test_multi_gmc_hand_computed(orc):  # both corners through the warp, then back to cx,cy,w,h; nothing else changes
    m, c = _xywh_state(orc, (10, 20, 30, 60))
    cases = [
        (np.eye(3), [20, 40, 20, 40]),
        ([[1, 0, 5], [0, 1, -3], [0, 0, 1]], [25, 37, 20, 40]),       # translation
        ([[2, 0, 0], [0, 2, 0], [0, 0, 1]], [40, 80, 40, 80]),        # zoom about the origin
        ([[1, 0, 0], [0, 1, 0], [0, 0, 2]], [10, 20, 10, 20]),        # the projective divide of :75-78
        ([[0, 1, 0], [1, 0, 0], [0, 0, 1]], [40, 20, 40, 20]),        # x <-> y
    ]
    for W, want in cases:
        mo, co = orc.kf_warp(orclib.KF_XYWH, m, c, np.asarray(W, np.float32))
        assert np.array_equal(mo[0, :4], np.asarray(want, np.float32)), (W, mo[0, :4])
        assert np.array_equal(mo[0, 4:], m[0, 4:]) and np.array_equal(co, c)


def test_xysr_affine_correction_hand_computed(orc):  # a quarter turn plus a shift: everything is exact in float32
    mean = np.array([[3, 4, 500, 0.5, 1, 2, 7]], np.float32)
    cov = np.zeros((1, 7, 7), np.float32)
    a, b, cc = 4.0, 1.0, 9.0
    cov[0, :2, :2] = [[a, b], [b, cc]]
    cov[0, 4:6, 4:6] = [[16, 2], [2, 25]]
    cov[0, :2, 4:6] = [[1, 2], [3, 4]]
    cov[0, 4:6, :2] = cov[0, :2, 4:6].T
    cov[0, 2, 2] = 11; cov[0, 3, 3] = 12; cov[0, 6, 6] = 13; cov[0, 2, 6] = cov[0, 6, 2] = 5
    W = np.array([[0, -1, 1], [1, 0, 2], [0, 0, 1]], np.float32)
    mo, co = orc.kf_warp(orclib.KF_XYSR, mean, cov, W)
    assert np.array_equal(mo[0], np.array([-3, 5, 500, 0.5, -2, 1, 7], np.float32))  # x[:2] = m x[:2] + t, x[4:6] = m x[4:6]
    R = W[:2, :2].astype(np.float64)
    want = cov[0].astype(np.float64).copy()
    want[:2, :2] = R @ want[:2, :2] @ R.T
    want[4:6, 4:6] = R @ want[4:6, 4:6] @ R.T
    want[:2, 4:6] = R @ cov[0, :2, 4:6] @ R.T
    want[4:6, :2] = want[:2, 4:6].T
    assert np.array_equal(co[0], want.astype(np.float32))
    with pytest.raises(ValueError):
        orc.kf_warp(orclib.KF_XYAH, np.zeros((1, 8), np.float32), np.zeros((1, 8, 8), np.float32), W)


def test_xysr_affine_correction_vs_float64(orc):
    r = np.random.default_rng(9)
    n = 64
    mean = r.normal(0, 50, (n, 7)).astype(np.float32)
    A = r.normal(0, 1, (n, 7, 7))
    cov = (A @ A.transpose(0, 2, 1)).astype(np.float32)
    th = 0.03
    W = np.array([[1.02 * np.cos(th), -1.02 * np.sin(th), 4.0], [1.02 * np.sin(th), 1.02 * np.cos(th), -6.0], [0, 0, 1]], np.float32)
    mo, co = orc.kf_warp(orclib.KF_XYSR, mean, cov, W)
    R, t = W[:2, :2].astype(np.float64), W[:2, 2].astype(np.float64)
    wm, wc = mean.astype(np.float64).copy(), cov.astype(np.float64).copy()
    wm[:, :2] = mean[:, :2] @ R.T + t
    wm[:, 4:6] = mean[:, 4:6] @ R.T
    for blk in ((slice(0, 2), slice(0, 2)), (slice(4, 6), slice(4, 6)), (slice(0, 2), slice(4, 6))):
        wc[(slice(None),) + blk] = R @ cov[(slice(None),) + blk].astype(np.float64) @ R.T
    wc[:, 4:6, :2] = wc[:, :2, 4:6].transpose(0, 2, 1)
    assert np.allclose(mo, wm, rtol=1e-5, atol=1e-4) and np.allclose(co, wc, rtol=1e-5, atol=1e-4)


def _panning_frames(frames, shift):
    # five static objects seen by a camera that jumps `shift` px per frame: without compensation consecutive boxes do not overlap
    base = np.array([[100 + 150 * i, 200, 140 + 150 * i, 290, 0.9, 0] for i in range(5)], np.float32)
    out = []
    for f in range(frames):
        d = base.copy()
        d[:, [0, 2]] += shift * f
        out.append(d)
    return out


def test_botsort_warp_keeps_tracks_on_their_objects_under_camera_pan(orc):
    shift = 60.0
    W = np.array([[1, 0, shift], [0, 1, 0]], np.float32)
    with_cmc, without = orc.tracker(orclib.BOTSORT), orc.tracker(orclib.BOTSORT)
    for f, d in enumerate(_panning_frames(8, shift)):
        if f > 0:
            with_cmc.set_camera_motion(W)
        oc, on = with_cmc.update(d), without.update(d)
        # compensated: the five ids of frame 1 stay on their detections (det_ind = row of the matched detection)
        assert sorted(oc[:, 4].astype(int)) == [1, 2, 3, 4, 5]
        order = np.argsort(oc[:, 4])
        assert np.array_equal(oc[order, 7].astype(int), np.arange(5))
        assert np.allclose(oc[order, :4], d[:, :4], atol=1e-2)
    # uncompensated: nothing overlaps from one frame to the next, the boxes that come out are not where the objects are
    assert on.shape[0] == 0 or np.abs(on[np.argsort(on[:, 4]), 0][:1] - d[0, 0]) > 100


def test_botsort_warp_is_one_shot_and_dropped_with_an_empty_frame(orc):
    frames = _panning_frames(3, 0.0)
    W = np.array([[1, 0, 500], [0, 1, 0]], np.float32)
    a, b = orc.tracker(orclib.BOTSORT), orc.tracker(orclib.BOTSORT)
    for t in (a, b):
        t.update(frames[0])
    a.set_camera_motion(W)
    assert a.update(frames[1][:0]).shape[0] == 0 and b.update(frames[1][:0]).shape[0] == 0  # botsort.cpp:267-269: returns before CMC
    assert np.array_equal(a.update(frames[1]), b.update(frames[1]))  # the warp went with the empty frame
    a.set_camera_motion(np.array([[1, 0, 0], [0, 1, 0]], np.float32))
    a.update(frames[2]); b.update(frames[2])
    assert np.array_equal(a.update(frames[2])[:, 4:], b.update(frames[2])[:, 4:])
    with pytest.raises(ValueError):
        orc.tracker(orclib.SORT).set_camera_motion(W)


# ---- StrongSORT (round 4): what src/trackers/strongsort.cpp does from a cold start, derived by hand from the code ----------------
def _ss_frame(shift=0.0):
    return np.array([[100 + shift, 100, 150, 220, 0.9, 0], [400 + shift, 300, 460, 430, 0.8, 0], [900 + shift, 500, 960, 640, 0.7, 1]], np.float32)


def test_strongsort_cold_start_as_written(orc):
    """Frame 1: three tentative tracks (ids 1-3, nothing emitted: none is confirmed). Frame 2: no confirmed track, so matching_cascade's
    empty index list means ALL tracks (strongsort.cpp:358-361, 440-447); nothing matches without samples; the IoU stage's candidates are the
    unconfirmed tracks PLUS stage A's unmatched tracks with time_since_update == 1 (:745-757) — every track twice. One copy of each track
    matches its detection (hits 2 < n_init 3), the other copy stays unmatched and puts the track on the unmatched list, where a tentative
    track is deleted (:189-197, :608-611). No detection is left over: the frame ends with NO track. Frame 3 starts over with ids 4-6."""
    import os
    old = os.environ.pop("GITHUB_ACTIONS", None)
    try:
        t = orc.tracker(orclib.STRONGSORT)
        assert t.update(_ss_frame()).shape == (0, 8)
        assert list(t.dump_states()[:, 0]) == [1, 2, 3]
        assert t.update(_ss_frame(1.0)).shape == (0, 8)
        laps = t.laps()
        assert [len(x) for x, _ in laps] == [3, 6] and int((laps[0][0] >= 0).sum()) == 0 and int((laps[1][0] >= 0).sum()) == 3
        assert t.dump_states().shape[0] == 0
        assert t.update(_ss_frame(2.0)).shape == (0, 8)
        assert list(t.dump_states()[:, 0]) == [4, 5, 6]
    finally:
        if old is not None:
            os.environ["GITHUB_ACTIONS"] = old


def test_strongsort_as_its_ci_runs_it(orc):
    """GITHUB_ACTIONS=true (strongsort.cpp:61-76): tracks are Confirmed at birth. Frame 1 emits them already (confirmed, time_since_update 0:
    :976-994); frame 2 matches them in the appearance stage through the sample library (one sample each after frame 1) and the boxes follow
    the detections; a detection below min_conf is dropped and det_ind keeps the caller's row numbers (:873-877). Stage A leaves NO detection
    unmatched here, and an empty list means "all detections" to the IoU stage (:362-365): track 2 (unmatched, time_since_update 1) is compared
    with both detections again, matches neither, and the IoU stage's unmatched detections — the two stage A had already matched — start the
    new tracks 4 and 5 (:765-776, :613-616), confirmed at birth and therefore emitted at once."""
    import os
    old = os.environ.get("GITHUB_ACTIONS")
    os.environ["GITHUB_ACTIONS"] = "true"
    try:
        t = orc.tracker(orclib.STRONGSORT)
        e = np.eye(3, 8, dtype=np.float32) + 0.01
        o1 = t.update(_ss_frame(), e)
        assert o1.shape == (3, 8) and list(o1[:, 4]) == [1, 2, 3] and list(o1[:, 7]) == [0, 1, 2]
        d2 = _ss_frame(2.0)
        d2[1, 4] = 0.05
        o2 = t.update(d2, e)
        laps = t.laps()
        assert len(laps[0][0]) == 3 and len(laps[0][1]) == 2 and list(laps[0][0]) == [0, -1, 1]  # stage A: tracks 1 and 3 find their detections
        assert list(o2[:, 4]) == [1, 3, 4, 5] and list(o2[:, 7]) == [0, 2, 0, 2]
        assert len(laps) == 2 and len(laps[1][0]) == 1 and len(laps[1][1]) == 2 and list(laps[1][0]) == [-1]
        assert abs(o2[0, 0] - 102.0) < 1.0 and o2[0, 2] - o2[0, 0] > 45  # the NSA update pulls the box onto the detection
        t.reset()
        assert t.update(_ss_frame(), e)[0, 4] == 1  # Tracker::reset :812-816: ids restart
    finally:
        if old is None:
            os.environ.pop("GITHUB_ACTIONS", None)
        else:
            os.environ["GITHUB_ACTIONS"] = old


# ---- the remaining cases of the reference's gtest files (round 4: every case that asserts something about this path is mirrored) ----
def test_sort_iou_threshold_and_multi_class(orc):  # tests/test_sort.cpp:87-110
    t = orc.tracker(orclib.SORT, [0.3, 3, 50, 1, 0.9])  # a very high IoU threshold
    t.update(SINGLE)
    far = np.array([[300, 300, 400, 400, 0.9, 0]], np.float32)
    out = t.update(far)
    # "Should not match due to low IoU, creates new track. Both tracks exist but old one not output (not updated)"
    assert out.shape == (1, 8) and out[0, 4] == 2 and t.dump_states().shape[0] == 2
    t = orc.tracker(orclib.SORT, [0.3, 3, 50, 1, 0.3])  # MultiClassTracking (:87-94; per_class is accepted and ignored by Sort::update)
    assert t.update(MULTI).shape[1] == 8


def test_low_confidence_detections(orc):  # tests/test_trackers.cpp:121-135, tests/test_sort.cpp:112-126
    low = np.array([[100, 100, 200, 200, 0.3, 0], [300, 300, 400, 400, 0.6, 0]], np.float32)
    # ByteTrack(det_thresh 0.5): track_thresh stays 0.45, the 0.3 detection only enters the second association -> at most one row
    out = orc.tracker(orclib.BYTETRACK).update(low)
    assert out.shape[0] <= 1 and (out.shape[0] == 0 or out[0, 7] == 1)
    assert orc.tracker(orclib.OCSORT).update(MULTI).shape[1] == 8  # OCSortUpdateReturnsValidOutput :113-119


def test_bytetrack_gtest_cases(orc):  # tests/test_bytetrack.cpp:38-123
    mixed = np.array([[100, 100, 200, 200, 0.9, 0], [300, 300, 400, 400, 0.3, 0], [500, 100, 600, 200, 0.6, 1]], np.float32)
    assert orc.tracker(orclib.BYTETRACK, [0.1, 0.45, 0.8, 30, 30]).update(mixed).shape[1] == 8  # TwoStageAssociation
    assert orc.tracker(orclib.BYTETRACK, [0.1, 0.6, 0.8, 25, 30]).update(mixed).shape[1] == 8   # TrackThresholdFiltering
    t = orc.tracker(orclib.BYTETRACK, [0.1, 0.45, 0.8, 30, 30])                                   # LostTrackRecovery
    d1 = np.array([[100, 100, 200, 200, 0.9, 0]], np.float32)
    d2 = np.array([[100, 100, 200, 200, 0.3, 0]], np.float32)
    o1, o2 = t.update(d1), t.update(d2)
    # "ByteTrack should still maintain the track using second stage association with low confidence detections": same id, det_ind 0
    assert o1.shape[0] == 1 and o2.shape[0] == 1 and o2[0, 4] == o1[0, 4] and o2[0, 5] == np.float32(0.3)
    t = orc.tracker(orclib.BYTETRACK)                                                             # ScoreDecay
    for _ in range(3):
        t.update(SINGLE)
    for _ in range(5):
        # (with NO detection in the frame the reference returns before the tracked list is touched: the track keeps being reported,
        # bytetrack.cpp:455-542 — DESIGN.md section 5 lists the quirk)
        assert t.update(EMPTY).shape == (1, 8)
    assert t.dump_states().shape[0] == 1
    for fps in (30, 60):                                                                            # FrameRateAwareness
        assert orc.tracker(orclib.BYTETRACK, [0.1, 0.45, 0.8, 30, fps]).update(SINGLE).shape[1] == 8


# ---- UCMCTrack (src/trackers/ucmc.cpp): the restatement against answers derived by hand from the reference's text ----------------------
def test_ucmc_confirmation_takes_three_frames_and_low_confidence_never_starts_a_track():
    """A detection seen in every frame: frame 1 starts a tentative track (initTentative :522-535; deleteOldTrackers leaves death_count 1),
    frame 2 matches it in associateTentative (birth_count 1 < 2: still tentative, no output), frame 3 makes birth_count 2 -> Confirmed
    (:497-500) and the track is reported with the DETECTION's own box, confidence, class and row (:303-342). A detection between
    det_thresh and high_score takes part in the second association only and never starts a track (:466-478 asks conf >= high_score)."""
    orc = orclib.load()
    t = orc.ucmc()
    d = np.array([[100, 100, 150, 220, 0.9, 2], [400, 300, 440, 380, 0.4, 1]], np.float32)
    assert t.update(d).shape[0] == 0
    s = t.dump_f64()
    assert s.shape[0] == 1 and s[0, :6].tolist() == [1, 0, 1, 0, 0, 0]  # id 1, Tentative, death 1, birth 0, holds detection 0, age 0
    # UCMCSingleTrack ctor :152-201 with the image-space fallback (:130-146): x = (cx, bottom) / 100, zero velocity, P = diag(1, vmax^2 / 3, 1, vmax^2 / 3)
    assert s[0, 6:10].tolist() == [np.float32(125.0) * 0.01, 0.0, np.float32(220.0) * 0.01, 0.0]
    assert np.array_equal(s[0, 10:].reshape(4, 4), np.diag([1.0, 100.0 / 3.0, 1.0, 100.0 / 3.0]))
    assert t.update(d).shape[0] == 0
    assert t.dump_f64()[0, :6].tolist() == [1, 0, 1, 1, 0, 1]
    out = t.update(d)
    assert out.tolist() == [[100.0, 100.0, 150.0, 220.0, 1.0, float(np.float32(0.9)), 2.0, 0.0]]
    assert t.dump_f64()[0, :6].tolist() == [1, 1, 1, 0, 0, 2]  # Confirmed, birth_count back to 0
    assert t.dump_f64().shape[0] == 1  # the 0.4 detection never became a track


def test_ucmc_coasting_and_deletion():
    """A confirmed track without detections: Coasted at once (:448-452), deleted when death_count reaches max_age (:544-546); a tentative
    track that misses one frame is deleted (death_count 2, :547)."""
    orc = orclib.load()
    t = orc.ucmc([0.3, 3, 100.0, 100.0, 5.0, 5.0, 10.0, 1.0 / 30.0, 0.5])
    d = np.array([[100, 100, 150, 220, 0.9, 0]], np.float32)
    for _ in range(3):
        t.update(d)
    assert t.dump_f64()[0, 1] == 1
    none = np.zeros((0, 6), np.float32)
    t.update(none)
    assert t.dump_f64()[0, 1:3].tolist() == [2, 2]  # Coasted; death_count: 1 after the matched frame, 2 now
    t.update(none)
    assert t.dump_f64().shape[0] == 0  # death_count 3 >= max_age 3
    t.update(d)  # a new tentative track (id 2) ...
    assert t.dump_f64()[0, :2].tolist() == [2, 0]
    t.update(none)  # ... that misses the next frame
    assert t.dump_f64().shape[0] == 0


def test_ucmc_distance_and_camera_mapping_by_hand():
    """distance (:213-223): x = 0, P = I, y = (3, 4), R = I: S = 2 I, Mahalanobis 25 / 2, log det = log 4. Camera (:57-112): Ki = [I | 0],
    Ko = a translation by 5 along the optical axis gives A = diag(1, 1, 5): the ground point of pixel (u, v) is (5 u, 5 v)."""
    import math
    orc = orclib.load()
    got = orc.ucmc_distance(np.zeros((1, 4)), np.eye(4).reshape(1, 16), np.array([[3.0, 4.0]]), np.array([[1.0, 0, 0, 1.0]]))
    assert got[0, 0] == np.float32(12.5 + math.log(4.0))
    ki = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float64)
    ko = np.eye(4)
    ko[2, 3] = 5.0
    t = orc.ucmc(None, (ki, ko))
    t.update(np.array([[10, 20, 30, 60, 0.9, 0]], np.float32))
    s = t.dump_f64()
    assert s[0, 6] == 5.0 * 20.0 and s[0, 8] == 5.0 * 60.0  # cx = 20, bottom = 60


# ---- BoostTrack (src/trackers/boosttrack.cpp), motion only ------------------------------------------------------------------------------
def test_boosttrack_first_frames_filters_and_the_confidence_boost():
    """Frame 1: every detection at or above det_thresh starts a track and is reported at once (time_since_update 0 and frame_count <=
    min_hits, :663-680) with get_state() of the fresh filter — the detection's own box through (cx, cy, h, w / h) and back — except
    boxes wider than aspect_ratio_thresh x their height, which filter_outputs drops (:434-463) although the track lives on. Frame 2: a
    detection BELOW det_thresh that overlaps a predicted track is lifted to max_iou x dlo_boost_coef (:393-400) — with identical boxes
    the IoU with the (stationary) prediction is 1, so 0.5 becomes 0.65 >= 0.6, the detection is kept, matched, and reported with the
    boosted confidence."""
    orc = orclib.load()
    t = orc.tracker(orclib.BOOSTTRACK)
    d = np.array([[100, 100, 150, 220, 0.9, 2], [600, 100, 900, 200, 0.95, 0], [300, 300, 340, 400, 0.5, 1]], np.float32)
    out = t.update(d)
    assert out.shape[0] == 1 and out[0, 4] == 1 and out[0, 7] == 0 and np.allclose(out[0, :4], d[0, :4], atol=1e-4)  # the wide box (id 2) is tracked but not shown
    assert t.dump_states().shape[0] == 2  # the 0.5 detection started nothing: no track to boost it yet
    d2 = d.copy()
    d2[2, :4] = d[0, :4]  # the weak detection now sits on track 1's box; the strong one is gone
    d2 = d2[1:]
    out = t.update(d2)
    ids = sorted(int(r[4]) for r in out)
    assert ids == [1]
    r = out[0]
    assert r[5] == np.float32(1.0) * np.float32(0.65) and int(r[7]) == 1 and int(r[6]) == 1  # boosted confidence, the detection's row and class
    # initial covariance 10 / 10000 (:51-53), one predict adds the velocity variances and Q (:56-59), the update shrinks it: positive, below 20
    P = t.dump_states()[0, 9:].reshape(8, 8)
    assert 0 < P[0, 0] < 20 and P[4, 4] < 10000.0


# ---- HybridSORT (src/trackers/hybridsort.cpp), as the reference runs it -------------------------------------------------------------------
def test_hybridsort_ids_order_and_the_zero_measurement_update():
    """Hand-derived from the reference's text: ids come from ++next_id_ and are reported + 1 (:21-23, :1225): the first two tracks of a
    tracker are 2 and 3; the table lists the tracks in REVERSE order (:1213); a new track is reported at once while frame_count <=
    min_hits, with the box of its fresh filter (the detection's box through (u, v, s, r) and back); a matched track reports its last
    observation = the detection's own box (:364-369). A track that misses a frame gets a Kalman update with an ALL-ZERO measurement
    (:1181-1188 -> :315-320): its position is pulled towards the origin by the gain — the reference as written."""
    orc = orclib.load()
    t = orc.tracker(orclib.HYBRIDSORT)
    d = np.array([[100, 100, 150, 220, 0.9, 2], [400, 300, 440, 380, 0.4, 1], [600, 100, 700, 300, 0.95, 0]], np.float32)
    out = t.update(d)
    assert [int(r[4]) for r in out] == [3, 2] and [int(r[7]) for r in out] == [2, 0]
    assert np.allclose(out[0, :4], d[2, :4], atol=1e-3) and np.allclose(out[1, :4], d[0, :4], atol=1e-3)
    out = t.update(d)
    assert np.array_equal(out[:, :4], d[[2, 0], :4])  # last observations: the detections themselves
    s0 = t.dump_states()
    u_before = s0[0, 1]
    t.update(d[2:])  # track 2 misses this frame (only the third detection is there)
    s1 = t.dump_states()
    # predicted u would be about 125; the zero measurement pulls it towards 0 by the gain P/(P+R) (close to 1 for a young track)
    assert s1[0, 0] == 2 and abs(s1[0, 1]) < 0.2 * u_before
