"""CPU checks of the restatements added in round 2 (no reference test covers them: "parity unpinned"), each against an independent
float64 numpy formulation of the same reference formula — gating distances and the two blends built on them
(kalman_filter.cpp:148-176, xywh_kf.hpp:140-176, matching.hpp:60-94, strongsort.cpp:449-492), fuse_iou (matching.cpp:109-128),
the NSA Kalman update (kalman_filter.cpp:60-112), the euclidean / dot embedding distances (matching.cpp:93-101,
deepocsort.cpp:404) and the feature normalise / EMA rules (botsort.cpp:38-46,158-169, reid_backend.cpp:72-88, deepocsort.cpp:132-150)."""
import numpy as np
import pytest

from tests import orclib

KF_XYAH, KF_XYWH = 1, 2


@pytest.fixture(scope="module")
def orc():
    return orclib.load()


def states(orc, kind, n, seed):
    r = np.random.default_rng(seed)
    h = r.uniform(30, 300, n)
    third = r.uniform(0.3, 0.6, n) if kind == KF_XYAH else r.uniform(0.3, 0.6, n) * h
    z = np.stack([r.uniform(100, 1800, n), r.uniform(100, 900, n), third, h], 1).astype(np.float32)
    mean, cov = orc.kf_initiate(kind, z)
    for _ in range(3):
        mean, cov = orc.kf_predict(kind, mean, cov)
        z = (z + r.normal(0, 1, z.shape) * np.array([2.0, 2.0, 0.01 if kind == KF_XYAH else 1.0, 2.0])).astype(np.float32)
        mean, cov = orc.kf_update(kind, mean, cov, z)
    return orc.kf_predict(kind, mean, cov) + (z,)


def innovation_cov(mean, cov, kind):
    S = cov.astype(np.float64)[:, :4, :4].copy()
    for i in range(len(S)):
        sd = np.full(4, float(mean[i, 3]) / 20.0)
        if kind == KF_XYAH:
            sd[2] = 0.1
        S[i] += np.diag(sd ** 2)
    return S


@pytest.mark.parametrize("kind", [KF_XYAH, KF_XYWH])
@pytest.mark.parametrize("pos", [False, True])
def test_gating_distance_against_float64(orc, kind, pos):
    mean, cov, z = states(orc, kind, 60, 3 + kind)
    r = np.random.default_rng(11)
    meas = (z[r.integers(0, 60, 90)] + r.normal(0, 15, (90, 4)) * np.array([1, 1, 0.002, 0.3])).astype(np.float32)
    g = orc.gate_cost(kind, 0, mean, cov, meas, only_position=pos)
    dim = 2 if pos else 4
    S = innovation_cov(mean, cov, kind)
    d = meas.astype(np.float64)[None, :, :dim] - mean.astype(np.float64)[:, None, :dim]
    if kind == KF_XYAH:  # z = S_sub^-1 d (the full LLT solve), |z|^2
        ref = (np.einsum("nab,nmb->nma", np.linalg.inv(S[:, :dim, :dim]), d) ** 2).sum(-1)
    else:  # d^T (S^-1)[:dim,:dim] d
        ref = np.einsum("nma,nab,nmb->nm", d, np.linalg.inv(S)[:, :dim, :dim], d)
    assert np.allclose(g, ref, rtol=1e-4, atol=1e-7)
    if kind == KF_XYAH:
        gg = orc.gate_cost(kind, 0, mean, cov, meas, only_position=pos, metric=1)
        assert np.allclose(gg, (d ** 2).sum(-1), rtol=1e-5)


def test_fuse_motion_and_strongsort_gate(orc):
    mean, cov, z = states(orc, KF_XYAH, 40, 9)
    meas = np.concatenate([z[:25], z[:25] + np.float32(400)]).astype(np.float32)
    cost = np.random.default_rng(2).uniform(0, 1, (40, 50)).astype(np.float32)
    g = orc.gate_cost(KF_XYAH, 0, mean, cov, meas)
    f = orc.gate_cost(KF_XYAH, 1, mean, cov, meas, cost, lam=0.98)
    assert np.array_equal(np.isinf(f), g > np.float32(9.4877)) and np.isinf(f).any() and np.isfinite(f).any()
    ok = np.isfinite(f)
    assert np.allclose(f[ok], 0.98 * cost[ok].astype(np.float64) + 0.02 * g[ok], rtol=1e-5)
    s = orc.gate_cost(KF_XYAH, 2, mean, cov, meas, cost, lam=0.995, gated_cost=1e5)
    c2 = np.where(g > np.float32(9.4877), 1e5, cost.astype(np.float64))
    assert np.allclose(s, 0.995 * c2 + 0.005 * g, rtol=1e-5)
    fp = orc.gate_cost(KF_XYAH, 1, mean, cov, meas, cost, only_position=True)
    gp = orc.gate_cost(KF_XYAH, 0, mean, cov, meas, only_position=True)
    assert np.array_equal(np.isinf(fp), gp > np.float32(5.9915))


def test_fuse_iou(orc):
    r = np.random.default_rng(4)
    a = np.array([[0, 0, 10, 10], [20, 20, 40, 40], [5, 5, 15, 15]], np.float32)
    b = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [5, 5, 15, 15], [22, 21, 41, 39]], np.float32)
    reid = r.uniform(0, 1, (3, 4)).astype(np.float32)
    out = orc.fuse_iou(reid, a, b)
    iou = orc.iou_batch(a, b).astype(np.float64)
    assert np.allclose(out, 1 - (1 - reid.astype(np.float64)) * (1 + iou) / 2, atol=1e-6)
    assert out[0, 0] == reid[0, 0] and np.isclose(out[0, 1], 1 - (1 - reid[0, 1]) / 2)


def test_nsa_kalman_update_against_float64(orc):
    mean, cov, z = states(orc, KF_XYAH, 30, 21)
    conf = np.linspace(0, 0.95, 30).astype(np.float32)
    m1, c1 = orc.kf_update_conf(mean, cov, z, conf)
    H = np.eye(4, 8)
    for i in range(30):
        x, P = mean[i].astype(np.float64), cov[i].astype(np.float64)
        sd = np.array([x[3] / 20, x[3] / 20, 0.1, x[3] / 20]) * (1.0 - float(conf[i]))
        S = H @ P @ H.T + np.diag(sd ** 2)
        K = P @ H.T @ np.linalg.inv(S)
        xn = x + K @ (z[i].astype(np.float64) - H @ x)
        Pn = P - K @ S @ K.T
        assert np.allclose(m1[i], xn, rtol=1e-4, atol=1e-4) and np.allclose(c1[i], Pn, rtol=1e-3, atol=1e-4)
    m0, c0 = orc.kf_update(KF_XYAH, mean, cov, z)
    assert np.array_equal(m1[0], m0[0]) and np.array_equal(c1[0], c0[0])  # confidence 0 = the plain update


def test_embedding_metrics_against_float64(orc):
    r = np.random.default_rng(8)
    a, b = r.standard_normal((17, 48)).astype(np.float32), r.standard_normal((23, 48)).astype(np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    assert np.allclose(orc.embedding_distance(1, a, b), a64 @ b64.T, rtol=1e-5, atol=1e-5)
    assert np.allclose(orc.embedding_distance(2, a, b), np.sqrt(((a64[:, None] - b64[None]) ** 2).sum(-1)), rtol=1e-5)
    cos = 1 - (a64 @ b64.T) / (np.linalg.norm(a64, axis=1)[:, None] * np.linalg.norm(b64, axis=1)[None])
    assert np.allclose(orc.embedding_distance(0, a, b), np.maximum(cos, 0), atol=1e-5)


def test_feature_rules(orc):
    r = np.random.default_rng(5)
    src = r.standard_normal((9, 32)).astype(np.float32)
    src[3] = 0
    src[4] *= 1e-9
    old = r.standard_normal((9, 32)).astype(np.float32)
    old /= np.linalg.norm(old, axis=1, keepdims=True)
    n = np.linalg.norm(src.astype(np.float64), axis=1, keepdims=True)
    n[3] = 1.0  # (the zero row is checked separately)
    set0 = orc.feat_update(0, old, src)
    assert np.allclose(set0[[0, 1, 2, 5]], (src / n)[[0, 1, 2, 5]], atol=1e-6) and np.array_equal(set0[3], src[3])
    norm2 = orc.feat_update(2, old, src)
    assert np.array_equal(norm2[4], src[4])  # |src| <= 1e-6: left as it is (reid_backend.cpp:80-84)
    ema = orc.feat_update(1, old, src, alpha=0.9)
    e = 0.9 * old.astype(np.float64) + 0.1 * src
    assert np.allclose(ema, e / np.linalg.norm(e, axis=1, keepdims=True), atol=1e-6)
