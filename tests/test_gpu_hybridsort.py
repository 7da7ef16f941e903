"""HybridSORT on the GPU (csrc/host/hybridsort.cpp + csrc/hybrid_kernels.hip) against the CPU oracle's restatement of
src/trackers/hybridsort.cpp as the reference runs it (its "simplified" association; zero-measurement updates of unmatched tracks): output
tables, every assignment of every frame and every track's nine-state Kalman filter, bit for bit, on seeded streams with ragged and empty
frames — hmiou and IoU, with and without the BYTE step and its score term, and the zero-feature behaviour of with_reid = true."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu


def run(P, M, frames, params=None, empty_every=9, seed=41, check_states_every=3):
    orc = orclib.load()
    trk = L.Tracker("hybridsort", params)
    ref = orc.tracker(orclib.HYBRIDSORT, params)
    st = SynthStream(P, M, seed)
    rows = laps = 0
    for f in range(frames):
        d, _ = st.next_frame()
        if empty_every and f % empty_every == empty_every - 2:
            d = d[:0]
        if f % 3 == 1:
            d = d.copy()
            d[::2, 4] *= 0.55  # half of the detections drop into the BYTE band (or below it)
        want = ref.update(d)
        got = trk.update(d)
        assert got.shape == want.shape and np.array_equal(got, want), f
        lg, lo = trk.laps(), ref.laps()
        assert len(lg) == len(lo), (f, len(lg), len(lo))
        for (xg, yg), (xo, yo) in zip(lg, lo):
            assert np.array_equal(xg, xo) and np.array_equal(yg, yo), f
        laps += len(lo)
        if f % check_states_every == check_states_every - 1:
            sg, so = trk.dump_states(), ref.dump_states()
            assert sg.shape == so.shape and np.array_equal(sg, so), (f, np.abs(sg - so).max() if sg.shape == so.shape else (sg.shape, so.shape))
        rows += want.shape[0]
    trk.close()
    return rows, laps


def test_defaults():
    rows, laps = run(30, 20, 60)
    assert rows > 150 and laps > 40


def test_eval_preset_and_iou():
    # hybridsort.yaml through the tool (motcpp_eval.cpp:279-316): det_thresh 0.5, iou_threshold 0.3, hmiou
    rows, laps = run(40, 30, 60, params=[0.5, 30, 3, 0.3, 1, 0.1, 1, 0.5, 4.6, 1.3, 1, 1, 1.0, 0], seed=43)
    assert rows > 300 and laps > 60
    run(40, 30, 40, params=[0.5, 5, 2, 0.25, 0, 0.1, 1, 0.5, 4.6, 1.3, 1, 1, 0.5, 0], seed=44)


def test_without_byte_without_score_term_without_first_step():
    run(30, 30, 40, params=[0.6, 30, 3, 0.2, 1, 0.1, 0, 0.5, 4.6, 1.3, 1, 1, 1.0, 0], seed=45)
    run(30, 30, 40, params=[0.6, 30, 3, 0.2, 1, 0.1, 1, 0.5, 4.6, 1.3, 1, 0, 1.0, 0], seed=46)
    run(30, 30, 30, params=[0.6, 30, 3, 0.2, 1, 0.1, 1, 0.5, 4.6, 1.3, 0, 1, 1.0, 0], seed=47)


def test_zero_feature_behaviour_of_with_reid():
    # with_reid = true and no embeddings: + EG_weight_high_score on the first association's costs and threshold, BYTE pairs never survive
    rows, laps = run(30, 20, 50, params=[0.6, 30, 3, 0.2, 1, 0.1, 1, 0.5, 4.6, 1.3, 1, 1, 1.0, 1], seed=48)
    assert rows > 60


def test_crowded():
    rows, laps = run(200, 150, 25, empty_every=0, seed=49, check_states_every=8)
    assert rows > 1000


def test_reset_and_embeddings_refused():
    orc = orclib.load()
    trk, ref = L.Tracker("hybridsort"), orc.tracker(orclib.HYBRIDSORT)
    st = SynthStream(15, 10, 3)
    for rep in range(2):
        for f in range(10):
            d, _ = st.next_frame()
            assert np.array_equal(trk.update(d), ref.update(d)), (rep, f)
        trk.reset()
        ref.reset()
    d, _ = st.next_frame()
    with pytest.raises(L.MotError):
        trk.update(d, np.ones((d.shape[0], 8), np.float32))
    trk.close()
