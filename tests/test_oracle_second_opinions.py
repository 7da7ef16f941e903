"""The oracle's restatements of the SURVEY 8 f3 trackers' filters against an INDEPENDENT second opinion: the fixture
tests/golden/f3_second_opinions.npz holds what numpy restatements written from the reference's sources give in float32 and in float64
(tests/golden/make_f3_second_opinions.py; generated in the build container, the script is committed). The reference has no test vector for
these filters, so the rule here is: the oracle (float32 state, its own summation orders) must sit as close to the float64 value of the formula as
the independent float32 evaluation does — differences between the two float32 evaluations are summation order, anything larger is a wrong
formula. CPU only: nothing here touches the product libraries."""
import os

import numpy as np
import pytest

from tests import orclib

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f3_second_opinions.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(FIX)


@pytest.fixture(scope="module")
def orc():
    return orclib.load()


def close_as_an_independent_f32(got, f32, f64, what, slack=8.0):
    """|got - f64| <= slack * |f32 - f64| elementwise-in-norm, plus an absolute floor of a few float32 ulps of the largest entry"""
    got, f32, f64 = (np.asarray(a, np.float64) for a in (got, f32, f64))
    scale = np.abs(f64).max() + 1e-30
    own = np.abs(f32 - f64).max()
    err = np.abs(got - f64).max()
    assert err <= slack * own + 64 * np.finfo(np.float32).eps * scale, (what, err, own, scale)


def test_hybridsort_nine_state_filter(orc, fx):
    boxes, present = fx["boxes"], fx["present"]
    trk = orc.tracker(orclib.HYBRIDSORT)
    for f in range(boxes.shape[0]):
        d = np.zeros((int(present[f].sum()), 6), np.float32)
        d[:, :5] = boxes[f][present[f]]
        trk.update(d)
        st = trk.dump_states()
        assert st.shape == (4, 91), (f, st.shape)  # id, x(9), P(81): the undetected track stays (zero-measurement updates, hybridsort.cpp:315-320)
        st = st[np.argsort(st[:, 0])]            # ids count up in birth order = detection order of frame 0
        for k in range(4):
            close_as_an_independent_f32(st[k, 1:10], fx["hyb_x_f32"][f, k], fx["hyb_x_f64"][f, k], ("x", f, k))
            close_as_an_independent_f32(st[k, 10:], fx["hyb_P_f32"][f, k], fx["hyb_P_f64"][f, k], ("P", f, k))
    # the zero-measurement updates really happened: the undetected track's state has collapsed towards the origin
    assert abs(st[3, 1]) < 0.2 * abs(fx["hyb_x_f64"][4, 3, 0])


def test_ucmc_ground_plane_filter(orc, fx):
    boxes = fx["boxes"]
    trk = orc.ucmc()
    for f in range(boxes.shape[0]):
        d = np.zeros((3, 6), np.float32)
        d[:, :5] = boxes[f][:3]
        trk.update(d)
        st = trk.dump_f64()
        assert st.shape[0] == 3, (f, st.shape)
        st = st[np.argsort(st[:, 0])]
        for k in range(3):
            assert np.allclose(st[k, 6:10], fx["ucmc_x"][f, k], rtol=1e-10, atol=1e-12), (f, k, st[k, 6:10], fx["ucmc_x"][f, k])
            assert np.allclose(st[k, 10:26], fx["ucmc_P"][f, k], rtol=1e-9, atol=1e-12), (f, k)


def test_ucmc_association_cost(orc, fx):
    X, P, Y, R = fx["ucmc_d_x"], fx["ucmc_d_P"], fx["ucmc_d_y"], fx["ucmc_d_R"]
    want = fx["ucmc_d"]
    for j in range(Y.shape[0]):  # (the oracle's entry point takes one R per column)
        got = orc.ucmc_distance(X, P, Y[j:j + 1], R[j:j + 1])[:, 0]
        assert np.allclose(got, want[:, j].astype(np.float32), rtol=2e-6, atol=1e-6), (j, got, want[:, j])


def test_xyah_gate_and_strongsort_blend(orc, fx):
    mean, cov, meas, cost = fx["gate_mean"], fx["gate_cov"], fx["gate_meas"], fx["gate_cost_in"]
    g = orc.gate_cost(orclib.KF_XYAH, 0, mean, cov, meas)
    close_as_an_independent_f32(g, fx["gate_g_f32"], fx["gate_g_f64"], "gating distances")
    c = orc.gate_cost(orclib.KF_XYAH, 2, mean, cov, meas, cost=cost, lam=0.98, gated_cost=1e5)
    # the same entries are gated off in all three evaluations (no distance sits on the threshold in this fixture), and the blend agrees
    assert np.array_equal(fx["gate_g_f64"] > 9.4877, g > 9.4877)
    close_as_an_independent_f32(c, fx["gate_c_f32"], fx["gate_c_f64"], "gated and blended costs")
    assert (fx["gate_g_f64"] > 9.4877).any() and (fx["gate_g_f64"] < 9.4877).any()
