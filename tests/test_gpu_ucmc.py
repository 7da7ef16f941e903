"""UCMCTrack on the GPU (csrc/host/ucmc.cpp + csrc/ucmc_kernels.hip) against the CPU oracle's restatement of src/trackers/ucmc.cpp: output
tables, the three assignments of every frame, and every track's bookkeeping and double-precision filter state — x and P bit for bit
(IEEE double operations in the reference's order on both sides; the only value that is not correctly rounded, log det S, only enters
the cost matrices, which are cast to float before the assignment)."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu

# a camera 5 m above the ground looking down at 30 degrees, 1000 px focal length: Ki 3 x 4, Ko 4 x 4 (row-major)
KI = np.array([[1000, 0, 960, 0], [0, 1000, 540, 0], [0, 0, 1, 0]], np.float64)
_c, _s = np.cos(np.deg2rad(120.0)), np.sin(np.deg2rad(120.0))
KO = np.array([[1, 0, 0, 0], [0, _c, -_s, 0], [0, _s, _c, 5.0], [0, 0, 0, 1]], np.float64)


def run(P, M, frames, fps=30.0, params=None, camera=None, empty_every=11, seed=5):
    orc = orclib.load()
    p = list(params if params is not None else [0.3, 30, 100.0, 100.0, 5.0, 5.0, 10.0, fps, 0.5])
    trk = L.Tracker("ucmc", p, camera=camera)
    po = list(p)
    po[7] = 1.0 / np.float64(np.float32(p[7]))  # the oracle takes dt itself, the handle the frame rate
    ref = orc.ucmc(po, camera)
    st = SynthStream(P, M, seed)
    rows = laps = 0
    for f in range(frames):
        d, _ = st.next_frame()
        if empty_every and f % empty_every == empty_every - 3:
            d = d[:0]
        if f % 4 == 1:
            d = d.copy()
            d[::3, 4] *= 0.55  # a third of the detections drop into the low-confidence band (and some below det_thresh)
        want = ref.update(d)
        got = trk.update(d)
        assert got.shape == want.shape and np.array_equal(got, want), f
        lg, lo = trk.laps(), ref.laps()
        assert len(lg) == len(lo), f
        for (xg, yg), (xo, yo) in zip(lg, lo):
            assert np.array_equal(xg, xo) and np.array_equal(yg, yo), f
        laps += len(lo)
        sg, so = trk.dump_f64(), ref.dump_f64()
        assert sg.shape == so.shape, f
        assert np.array_equal(sg[:, :6], so[:, :6]), f  # id, state, death, birth, det_idx, age
        assert np.array_equal(sg[:, 6:], so[:, 6:]), (f, np.abs(sg[:, 6:] - so[:, 6:]).max())  # x, P
        rows += want.shape[0]
    trk.close()
    return rows, laps


def test_image_space_fallback_small():
    rows, laps = run(20, 12, 60)
    assert rows > 300 and laps > 60


def test_crowded():
    rows, laps = run(200, 150, 40, empty_every=0)
    assert rows > 3000 and laps > 80


def test_calibrated_camera():
    rows, laps = run(60, 40, 50, camera=(KI, KO), seed=9)
    assert rows > 500 and laps > 50


def test_short_memory_and_other_rates():
    run(40, 25, 70, fps=25.0, params=[0.2, 4, 60.0, 40.0, 3.0, 7.0, 8.0, 25.0, 0.6], seed=11)
    run(40, 25, 40, fps=14.0, params=[0.3, 30, 100.0, 100.0, 5.0, 5.0, 10.0, 14.0, 0.5], seed=12, empty_every=5)


def test_reset_restarts_ids():
    orc = orclib.load()
    trk = L.Tracker("ucmc")
    ref = orc.ucmc()
    st = SynthStream(15, 10, 3)
    for rep in range(2):
        for f in range(12):
            d, _ = st.next_frame()
            assert np.array_equal(trk.update(d), ref.update(d)), (rep, f)
        trk.reset()
        ref.reset()
    trk.close()


def test_public_class_through_the_c_handle_defaults():
    # the defaults of kind 6 are the constructor defaults of motcpp::trackers::UCMCTrack (dt = 1 / 30)
    orc = orclib.load()
    trk, ref = L.Tracker("ucmc"), orc.ucmc()
    st = SynthStream(10, 8, 21)
    n = 0
    for f in range(20):
        d, _ = st.next_frame()
        got, want = trk.update(d), ref.update(d)
        assert np.array_equal(got, want), f
        n += want.shape[0]
    assert n > 50
    trk.close()
