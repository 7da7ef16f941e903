"""GPU parity tests of the four trackers (host lifecycle in C++ over the HIP kernels) against the CPU oracle on
identical seeded detection streams: output tables (boxes, ids, conf, cls, det_ind), every assignment solved during
the frame (index-for-index) and the Kalman state of every live track. Integer results must be identical; floats
are asserted within the 1e-4 relative contract and additionally reported/required bit-identical where the kernels
follow the oracle's operation order."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu


def same_laps(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for (xa, ya), (xb, yb) in zip(a, b):
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)


def check_frame(f, out_g, out_o, trk_g, trk_o, exact=True):
    assert out_g.shape == out_o.shape, (f, out_g.shape, out_o.shape)
    same_laps(trk_g.laps(), trk_o.laps())
    assert np.array_equal(out_g[:, 4:], out_o[:, 4:]), f  # id, conf, cls, det_ind
    assert np.allclose(out_g[:, :4], out_o[:, :4], rtol=1e-4, atol=1e-3), f
    sg, so = trk_g.dump_states(), trk_o.dump_states()
    assert sg.shape == so.shape, (f, sg.shape, so.shape)
    if sg.size:
        assert np.array_equal(sg[:, 0], so[:, 0]), f  # ids in list order
        assert np.allclose(sg, so, rtol=1e-4, atol=1e-4), f
    if exact:
        assert np.array_equal(out_g, out_o), f
        assert np.array_equal(sg, so), (f, np.abs(sg - so).max())
    if hasattr(trk_o, "dump_features") and trk_o.kind in (orclib.BOTSORT, orclib.DEEPOCSORT) and f % 5 == 4:
        # the stored appearance state (botsort.hpp smooth_feat_: normalised at birth, EMA 0.9/0.1 + renormalise per update)
        fg, fo = trk_g.dump_features(), trk_o.dump_features()
        assert fg.shape[0] == fo.shape[0], f
        if fo.shape[1] and fg.shape[1]:  # (DeepOC-SORT with embedding_off keeps a dummy 1-d embedding in the reference: nothing to compare)
            assert fg.shape == fo.shape and np.array_equal(fg, fo), (f, np.abs(fg - fo).max())


def run_stream(kind_g, kind_o, P, M, frames, emb_dim=0, params=None, seed=1234, exact=True):
    orc = orclib.load()
    tg, to = L.Tracker(kind_g, params), orc.tracker(kind_o, params)
    s = SynthStream(P, M, seed, emb_dim)
    n_out = n_exact = 0
    for f in range(frames):
        d, e = s.next_frame()
        if f % 17 == 13:
            d = d[:0]  # an empty frame now and then
            e = e[:0] if e is not None else None
        og, oo = tg.update(d, e), to.update(d, e)
        check_frame(f, og, oo, tg, to, exact)
        n_out += og.shape[0]
        n_exact += int(np.sum(np.all(og == oo, axis=1))) if og.shape == oo.shape else 0
    assert n_out > 0
    # trackers whose cost goes through acos are compared at 1e-4, but at least 99.9 % of their output
    # rows must still be bit-identical to the oracle's at tracker level (in practice all of them are)
    assert n_exact >= 0.999 * n_out, (n_exact, n_out)
    tg.close()


def test_sort_stream():
    run_stream("sort", orclib.SORT, 120, 70, 60, params=[0.3, 3, 50, 3, 0.3])


def test_sort_reference_known_answers():
    # tests/test_sort.cpp:36-84,128-148 through the GPU path
    single = np.array([[100, 100, 200, 200, 0.9, 0]], np.float32)
    empty = np.zeros((0, 6), np.float32)
    t = L.Tracker("sort", [0.3, 1, 50, 1])
    out = t.update(single)
    assert out.shape == (1, 8) and out[0, 2] > out[0, 0] and out[0, 3] > out[0, 1]
    t = L.Tracker("sort", [0.3, 3, 50, 1])
    t.update(single); t.update(single)
    out = t.update(np.array([[110, 110, 210, 210, 0.9, 0]], np.float32))
    assert out.shape[0] == 1 and int(out[0, 4]) == 1
    t = L.Tracker("sort", [0.3, 2, 50, 1])
    t.update(single); t.update(empty)
    assert t.update(empty).shape[0] == 0
    t = L.Tracker("sort", [0.3, 5, 50, 1])
    for i in range(5):
        t.update(np.array([[100 + i * 10, 100 + i * 10, 200 + i * 10, 200 + i * 10, 0.9, 0]], np.float32))
    t.update(empty)
    out = t.update(np.array([[160, 160, 260, 260, 0.9, 0]], np.float32))
    assert out.shape[0] == 1 and int(out[0, 4]) == 1


def test_sort_nan_track_is_dropped():
    orc = orclib.load()
    tg, to = L.Tracker("sort", [0.3, 5, 50, 1]), orc.tracker(orclib.SORT, [0.3, 5, 50, 1])
    # a degenerate detection (x2 < x1 and y2 > y1 -> negative area*ratio) makes xysr2xyxy produce NaN after predict
    seq = [np.array([[100, 100, 200, 200, 0.9, 0], [300, 300, 290, 340, 0.9, 0]], np.float32),
           np.array([[102, 101, 202, 201, 0.9, 0]], np.float32),
           np.array([[104, 102, 204, 202, 0.9, 0], [500, 500, 560, 640, 0.8, 0]], np.float32)]
    for f, d in enumerate(seq):
        check_frame(f, tg.update(d), to.update(d), tg, to)


def test_bytetrack_c2_stream():
    run_stream("bytetrack", orclib.BYTETRACK, 256, 128, 80)


def test_bytetrack_small_and_ragged():
    orc = orclib.load()
    tg, to = L.Tracker("bytetrack"), orc.tracker(orclib.BYTETRACK)
    r = np.random.default_rng(3)
    s = SynthStream(40, 30, 7)
    for f in range(50):
        d, _ = s.next_frame()
        d = d[: r.integers(0, 31)]
        check_frame(f, tg.update(d), to.update(d), tg, to)


def test_bytetrack_north_star_shape():
    run_stream("bytetrack", orclib.BYTETRACK, 1000, 500, 12)


def test_ocsort_stream():
    run_stream("ocsort", orclib.OCSORT, 300, 150, 60, exact=False)


def test_ocsort_c4_shape():
    # BASELINE configs[3]: 4096 objects, 2048 detections per frame. From the fourth frame on the reference's OC-SORT carries
    # duplicated tracks (Q4: a match rejected by the IoU post-filter is pushed to the unmatched lists twice, ocsort.cpp:709-734),
    # i.e. exact ties: those associations are the exact lapjv emulation's, the earlier ones the fast path's.
    run_stream("ocsort", orclib.OCSORT, 4096, 2048, 6, exact=False)


def test_bytetrack_c5_shape():
    # the per-GPU shape of BASELINE configs[4]: 1000 objects x 512 detections
    run_stream("bytetrack", orclib.BYTETRACK, 1000, 512, 12)


def test_ocsort_use_byte():
    run_stream("ocsort", orclib.OCSORT, 120, 80, 40, params=[0.5, 30, 50, 3, 0.3, 0.1, 3, 0.2, 1], exact=False)


@pytest.mark.parametrize("asso", [1, 2, 3, 4, 5])  # hmiou, giou, ciou, diou, centroid
def test_ocsort_association_measures(asso):
    # asso_func reaches all three association stages (ocsort.cpp:413,438,494); use_byte on so that all of them run
    run_stream("ocsort", orclib.OCSORT, 150, 90, 40, params=[0.5, 30, 50, 3, 0.3, 0.1, 3, 0.2, 1, 0.01, 0.0001, asso], exact=False)


def test_botsort_with_embeddings():
    run_stream("botsort", orclib.BOTSORT, 256, 128, 40, emb_dim=64)


def test_botsort_c3_shape():
    run_stream("botsort", orclib.BOTSORT, 1024, 512, 6, emb_dim=256)


def test_botsort_without_embeddings():
    run_stream("botsort", orclib.BOTSORT, 120, 70, 40)


def test_botsort_camera_motion():
    # a caller-supplied warp per frame (what cmc_->apply returns in botsort.cpp:317-324): a panning, slightly zooming camera;
    # every third frame has none, and a warp handed over before an empty frame is dropped with it (:267-269)
    orc = orclib.load()
    tg, to = L.Tracker("botsort"), orc.tracker(orclib.BOTSORT)
    s = SynthStream(200, 110, 77, 32)
    r = np.random.default_rng(5)
    pan = np.zeros(2, np.float32)
    n_out = 0
    for f in range(50):
        d, e = s.next_frame()
        step = r.uniform(-6, 6, 2).astype(np.float32)
        pan += step
        d = d.copy()
        d[:, [0, 2]] += pan[0]
        d[:, [1, 3]] += pan[1]
        if f % 11 == 7:
            d, e = d[:0], e[:0]
        if f % 3 != 2:
            k = np.float32(1.0 + r.uniform(-0.004, 0.004))
            W = np.array([[k, 0.001, step[0]], [-0.001, k, step[1]]], np.float32)
            tg.set_camera_motion(W)
            to.set_camera_motion(W)
        og, oo = tg.update(d, e), to.update(d, e)
        check_frame(f, og, oo, tg, to)
        n_out += og.shape[0]
    assert n_out > 1000
    tg.close()


def test_camera_motion_only_for_botsort():
    t = L.Tracker("bytetrack")
    with pytest.raises(L.MotError):
        t.set_camera_motion(np.eye(3, dtype=np.float32)[:2])
    t.close()


def test_batch_matches_single_streams():
    S, P, M = 5, 120, 60
    b = L.Batch("bytetrack", S)
    singles = [L.Tracker("bytetrack") for _ in range(S)]
    streams = [SynthStream(P, M, 100 + s) for s in range(S)]
    batch_flushes = 0
    for f in range(25):
        frames = [st.next_frame()[0] for st in streams]
        before = b.counters()["flushes"]
        out, cnt = b.step(np.stack(frames))
        batch_flushes += b.counters()["flushes"] - before
        for s in range(S):
            ref = singles[s].update(frames[s])
            assert cnt[s] == ref.shape[0] and np.array_equal(out[s, :cnt[s]], ref)
    assert b.counters()["frames"] == 25 * S
    assert batch_flushes <= 25 * 4  # lockstep: at most 4 flushes per frame however many streams are in the batch


# ---- DeepOC-SORT: OC-SORT's lifecycle + embedding similarity in the cost, per-detection EMA, CMC on
# observations. params: det_thresh, max_age, max_obs, min_hits, iou_thr, delta_t, inertia, w_emb, alpha_fixed, aw_param,
# emb_off, cmc_off, aw_off, q_xy, q_s, asso
def test_deepocsort_stream():
    run_stream("deepocsort", orclib.DEEPOCSORT, 200, 110, 50, emb_dim=32, exact=False)


def test_deepocsort_fixed_weight_and_larger_embedding():
    run_stream("deepocsort", orclib.DEEPOCSORT, 150, 90, 35, emb_dim=128, exact=False,
               params=[0.3, 30, 50, 3, 0.3, 3, 0.2, 0.5, 0.95, 0.5, 0, 0, 1])  # aw_off


def test_deepocsort_without_embeddings_is_not_ocsort():
    # embedding_off: the cost is OC-SORT's, but the duplicated unmatched lists (two tracks per unmatched detection) remain
    run_stream("deepocsort", orclib.DEEPOCSORT, 120, 70, 35, exact=False, params=[0.3, 30, 50, 3, 0.3, 3, 0.2, 0.5, 0.95, 0.5, 1, 0, 0])


def test_deepocsort_camera_motion():
    orc = orclib.load()
    tg, to = L.Tracker("deepocsort"), orc.tracker(orclib.DEEPOCSORT)
    s = SynthStream(150, 90, 91, 24)
    r = np.random.default_rng(8)
    pan = np.zeros(2, np.float32)
    n_out = 0
    for f in range(40):
        d, e = s.next_frame()
        step = r.uniform(-5, 5, 2).astype(np.float32)
        pan += step
        d = d.copy()
        d[:, [0, 2]] += pan[0]
        d[:, [1, 3]] += pan[1]
        if f % 3 != 2:  # a warp for two frames out of three (what cmc_->apply would have returned)
            k = np.float32(1.0 + r.uniform(-0.003, 0.003))
            W = np.array([[k, 0.001, step[0]], [-0.001, k, step[1]]], np.float32)
            tg.set_camera_motion(W)
            to.set_camera_motion(W)
        og, oo = tg.update(d, e), to.update(d, e)
        check_frame(f, og, oo, tg, to, exact=False)
        fg, fo = tg.dump_features(), to.dump_features()
        assert fg.shape == fo.shape and np.allclose(fg, fo, rtol=1e-5, atol=1e-6), f
        n_out += len(oo)
    assert n_out > 500


# ---- the trackers of round 4 in a StreamBatch: one launch per kernel family and stage however many streams ---------------------------------
@pytest.mark.parametrize("kind,max_flushes", [("strongsort", 3), ("ucmc", 3), ("boosttrack", 3), ("hybridsort", 5)])
def test_round4_trackers_batch_matches_single_streams(kind, max_flushes):
    S, P, M = 6, 60, 40
    b = L.Batch(kind, S)
    singles = [L.Tracker(kind) for _ in range(S)]
    streams = [SynthStream(P, M, 300 + s) for s in range(S)]
    flushes = rows = 0
    for f in range(20):
        frames = [st.next_frame()[0] for st in streams]
        if f == 7:
            frames[2] = frames[2][:0]  # one stream has an empty frame
        n = max(len(x) for x in frames)
        dets = np.zeros((S, max(n, 1), 6), np.float32)
        cnt = np.zeros(S, np.int32)
        for s in range(S):
            dets[s, :len(frames[s])] = frames[s]
            cnt[s] = len(frames[s])
        before = b.counters()["flushes"]
        out, oc = b.step(dets, cnt)
        flushes += b.counters()["flushes"] - before
        for s in range(S):
            ref = singles[s].update(frames[s])
            assert oc[s] == ref.shape[0] and np.array_equal(out[s, :oc[s]], ref), (f, s)
            rows += ref.shape[0]
    assert rows > 100
    assert flushes <= 20 * max_flushes  # lockstep: the streams' stages share their launches
