"""BoostTrack (motion only) on the GPU (csrc/host/boosttrack.cpp + csrc/boost_kernels.hip) against the CPU oracle's restatement of
src/trackers/boosttrack.cpp: output tables (the boosted confidences included), the assignment of every frame and every track's Kalman
state, bit for bit, on seeded streams with ragged and empty frames — default DLO boost, the BoostTrack++ switches (soft-BIoU, visual
tracking) and no boost at all."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu


def run(P, M, frames, params=None, empty_every=9, seed=31, check_states_every=4):
    orc = orclib.load()
    trk = L.Tracker("boosttrack", params)
    ref = orc.tracker(orclib.BOOSTTRACK, params)
    st = SynthStream(P, M, seed)
    rows = laps = 0
    for f in range(frames):
        d, _ = st.next_frame()
        if empty_every and f % empty_every == empty_every - 2:
            d = d[:0]
        if f % 3 == 1:
            d = d.copy()
            d[::2, 4] *= 0.6  # half of the detections fall below det_thresh: the boost decides which of them come back
        want = ref.update(d)
        got = trk.update(d)
        assert got.shape == want.shape and np.array_equal(got, want), f
        lg, lo = trk.laps(), ref.laps()
        assert len(lg) == len(lo), f
        for (xg, yg), (xo, yo) in zip(lg, lo):
            assert np.array_equal(xg, xo) and np.array_equal(yg, yo), f
        laps += len(lo)
        if f % check_states_every == check_states_every - 1:
            sg, so = trk.dump_states(), ref.dump_states()
            assert sg.shape == so.shape and np.array_equal(sg, so), (f, np.abs(sg - so).max() if sg.shape == so.shape else None)
        rows += want.shape[0]
    trk.close()
    return rows, laps


def test_default_boost():
    rows, laps = run(30, 20, 60)
    assert rows > 150 and laps > 30


def test_boosttrack_plus_plus_switches():
    # use_sb, use_vt (motcpp_eval.cpp:247-278), a lower threshold and a short memory
    rows, laps = run(40, 30, 60, params=[0.5, 5, 2, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 1, 1], seed=33)
    assert rows > 150 and laps > 30
    run(40, 30, 30, params=[0.6, 60, 3, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 0, 1], seed=34)
    run(40, 30, 30, params=[0.6, 60, 3, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 1, 0], seed=35)


def test_no_boost_and_other_weights():
    run(25, 25, 40, params=[0.4, 30, 3, 0.4, 200, 1.2, 0.5, 0.6, 0.25, 0, 0, 0.65, 0, 0], seed=36)


def test_crowded():
    rows, laps = run(200, 150, 30, empty_every=0, seed=37, check_states_every=10)
    assert rows > 1000


def test_reset_restarts_ids():
    orc = orclib.load()
    trk, ref = L.Tracker("boosttrack"), orc.tracker(orclib.BOOSTTRACK)
    st = SynthStream(15, 10, 3)
    for rep in range(2):
        for f in range(10):
            d, _ = st.next_frame()
            assert np.array_equal(trk.update(d), ref.update(d)), (rep, f)
        trk.reset()
        ref.reset()
    trk.close()


def run_reid(P, M, frames, D, params, seed, empty_every=9, skip_embs_every=0):
    """with_reid = true: embeddings with update(); tables, assignments, Kalman states and the stored embeddings against the oracle"""
    orc = orclib.load()
    trk = L.Tracker("boosttrack", params)
    ref = orc.tracker(orclib.BOOSTTRACK, params)
    st = SynthStream(P, M, seed, D)
    rows = 0
    for f in range(frames):
        d, e = st.next_frame()
        if empty_every and f % empty_every == empty_every - 2:
            d, e = d[:0], e[:0]
        if f % 3 == 1:
            d = d.copy()
            d[::2, 4] *= 0.6
        if f % 5 == 2 and len(e):
            e = e.copy()
            e[0] = 0.0  # a zero embedding: stays as it is (its norm is not positive, boosttrack.cpp:150-152, :187-189)
        ee = None if (skip_embs_every and f % skip_embs_every == skip_embs_every - 1) else e
        want = ref.update(d, ee)
        got = trk.update(d, ee)
        assert got.shape == want.shape and np.array_equal(got, want), f
        for (xg, yg), (xo, yo) in zip(trk.laps(), ref.laps()):
            assert np.array_equal(xg, xo) and np.array_equal(yg, yo), f
        if f % 3 == 2:
            sg, so = trk.dump_states(), ref.dump_states()
            assert sg.shape == so.shape and np.array_equal(sg, so), f
            fg, fo = trk.dump_features(), ref.dump_features()
            if fo.shape[1]:
                assert fg.shape == fo.shape and np.array_equal(fg, fo), (f, np.abs(fg - fo).max() if fg.shape == fo.shape else (fg.shape, fo.shape))
        rows += want.shape[0]
    trk.close()
    return rows


def test_with_reid_embeddings():
    assert run_reid(30, 20, 50, 16, [0.6, 60, 3, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 0, 0, 1], 51) > 150
    run_reid(40, 30, 40, 128, [0.5, 10, 2, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 1, 1, 1], 52)


def test_with_reid_frames_without_embeddings_are_motion_only():
    run_reid(25, 20, 40, 8, [0.6, 60, 3, 0.3, 10, 1.6, 0.5, 0.25, 0.25, 1, 1, 0.65, 0, 0, 1], 53, skip_embs_every=4)
