"""BoT-SORT with the lifecycle on the device (mot_bot_*, motcpp_amd/csrc/bot_device.hip) against the CPU oracle: output tables,
track ids, the Kalman state and the smooth feature of every live track, bit for bit, on seeded streams with ragged and empty
frames, with and without embeddings, with per-stream camera-motion warps."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu


def run(shapes, frames, cap, maxd, emb_dim=0, params=None, warps=False, check_states_every=5, empty_every=13):
    orc = orclib.load()
    S = len(shapes)
    dev = L.DeviceBotSort(S, cap, maxd, emb_dim, params)
    streams = [SynthStream(P, M, 4321 + i, emb_dim) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(orclib.BOTSORT, (list(params) + [30, 50]) if params else None) for _ in range(S)]
    r = np.random.default_rng(9)
    pan = np.zeros((S, 2), np.float32)
    rows = 0
    for f in range(frames):
        dets = np.zeros((S, maxd, 6), np.float32)
        embs = np.zeros((S, maxd, emb_dim), np.float32) if emb_dim else None
        cnt = np.zeros(S, np.int32)
        W = np.zeros((S, 6), np.float32)
        hw = np.zeros(S, np.uint8)
        per = []
        for s, st in enumerate(streams):
            d, e = st.next_frame()
            if warps:  # a panning camera: detections move with it, the warp says by how much
                step = r.uniform(-5, 5, 2).astype(np.float32)
                pan[s] += step
                d = d.copy()
                d[:, [0, 2]] += pan[s, 0]
                d[:, [1, 3]] += pan[s, 1]
                if (f + s) % 3 != 2:
                    k = np.float32(1.0 + r.uniform(-0.004, 0.004))
                    W[s] = [k, 0.001, step[0], -0.001, k, step[1]]
                    hw[s] = 1
            if empty_every and (f + s) % empty_every == empty_every - 2:
                d = d[:0]
                e = e[:0] if e is not None else None
            per.append((d, e))
            cnt[s] = len(d)
            dets[s, :len(d)] = d
            if emb_dim:
                embs[s, :len(d)] = e
        tables = dev.step(dets, cnt, embs, W if warps else None, hw if warps else None)
        for s in range(S):
            d, e = per[s]
            if warps and hw[s]:
                oracles[s].set_camera_motion(W[s].reshape(2, 3))
            oo = oracles[s].update(d, e if emb_dim else None)
            assert tables[s].shape == oo.shape, (f, s, tables[s].shape, oo.shape)
            assert np.array_equal(tables[s], oo), (f, s)
            rows += oo.shape[0]
            if f % check_states_every == check_states_every - 1:
                ids, mean, cov, feats, has = dev.dump(s)
                so = oracles[s].dump_states()
                assert len(ids) == so.shape[0], (f, s)
                if len(ids):
                    assert np.array_equal(ids, so[:, 0].astype(np.int32)), (f, s)
                    assert np.array_equal(mean, so[:, 1:9]), (f, s)
                    assert np.array_equal(cov.reshape(len(ids), -1), so[:, 9:73]), (f, s)
                if emb_dim:
                    fo = oracles[s].dump_features()
                    assert fo.shape[0] == len(ids)
                    if fo.shape[1]:
                        assert np.array_equal(feats[has != 0], fo[has != 0]), (f, s)
    assert rows > 0
    dev.close()


def test_small_streams_with_embeddings():
    run([(40, 30), (256, 128), (8, 8), (90, 64)], 45, 768, 128, emb_dim=64)


def test_without_embeddings():
    run([(120, 70), (30, 20), (200, 100)], 40, 512, 128)


def test_reid_on_but_no_features_in_the_frames():
    run([(60, 40), (100, 50)], 30, 512, 64, emb_dim=0, params=[0.5, 0.1, 0.6, 30, 0.8, 0.5, 0.25, 30, 0, 1])


def test_camera_motion_per_stream():
    run([(200, 110), (80, 50), (150, 90)], 50, 768, 128, emb_dim=32, warps=True, empty_every=11)


def test_custom_thresholds_and_score_fusion():
    run([(150, 80), (60, 60)], 40, 512, 128, emb_dim=16, params=[0.55, 0.15, 0.65, 20, 0.75, 0.6, 0.3, 25, 1, 1])


def test_c3_shape():
    run([(1024, 512)], 14, 2048, 512, emb_dim=256, empty_every=0)  # BASELINE configs[2]; the oracle needs ~60 ms per frame here


def test_long_run_recycles_slots():
    run([(50, 30), (25, 20)], 220, 256, 64, emb_dim=8, params=[0.5, 0.1, 0.6, 5, 0.8, 0.5, 0.25, 30, 0, 1], check_states_every=20)


def test_two_frames_in_flight_give_the_same_tables():
    """mot_bot_enqueue_packed / mot_bot_collect_packed with embeddings and per-stream warps: frame f + 1 is queued before frame f
    is fetched; the packed rows of every frame equal the oracle's tables"""
    import torch
    orc = orclib.load()
    shapes = [(60, 40), (200, 120), (30, 30)]
    S, maxd, E, F = len(shapes), 128, 16, 30
    dev = L.DeviceBotSort(S, 512, maxd, E)
    streams = [SynthStream(P, M, 999 + i, E) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(orclib.BOTSORT) for _ in range(S)]
    r = np.random.default_rng(3)
    soa = np.zeros((F, S, 6, maxd), np.float32)
    emb = np.zeros((F, S, maxd, E), np.float32)
    counts = np.zeros((F, S), np.int32)
    W = np.zeros((F, S, 6), np.float32)
    HW = np.zeros((F, S), np.uint8)
    want = []
    for f in range(F):
        per = []
        for s, st in enumerate(streams):
            d, e = st.next_frame()
            if (f + s) % 7 == 5:
                d, e = d[:0], e[:0]
            if (f + s) % 4 == 1:
                W[f, s] = [1.0, 0.0, r.uniform(-3, 3), 0.0, 1.0, r.uniform(-3, 3)]
                HW[f, s] = 1
                oracles[s].set_camera_motion(W[f, s].reshape(2, 3))
            counts[f, s] = len(d)
            soa[f, s, :, :len(d)] = d.T
            emb[f, s, :len(d)] = e
            per.append(oracles[s].update(d, e))
        want.append(per)
    ddets, dembs = torch.from_numpy(soa).cuda(), torch.from_numpy(emb).cuda()
    rows = L.pinned_array(dev.ctx, (S * 512, 8), np.float32)
    cnt = L.pinned_array(dev.ctx, (S,), np.int32)

    def enq(f):
        dev.enqueue_packed(ddets.data_ptr() + f * S * 6 * maxd * 4, counts[f].copy(), rows.shape[0],
                           embs_ptr=dembs.data_ptr() + f * S * maxd * E * 4, warps=W[f].copy(), has_warp=HW[f].copy())

    def check(f):
        total = dev.collect_packed(rows, cnt)
        assert total == sum(w.shape[0] for w in want[f]), f
        off = np.concatenate([[0], np.cumsum(cnt)])
        for s in range(S):
            assert np.array_equal(rows[off[s]:off[s + 1]], want[f][s]), (f, s)

    enq(0)
    for f in range(1, F):
        enq(f)
        check(f - 1)
    check(F - 1)
    dev.close()


def test_capacity_error_is_reported():
    dev = L.DeviceBotSort(1, 16, 64, 8)
    st = SynthStream(64, 40, 3, 8)
    with pytest.raises(L.MotError):
        for _ in range(5):
            d, e = st.next_frame()
            dets = np.zeros((1, 64, 6), np.float32)
            embs = np.zeros((1, 64, 8), np.float32)
            dets[0, :len(d)] = d
            embs[0, :len(d)] = e
            dev.step(dets, np.array([len(d)], np.int32), embs)
    dev.close()


def test_reset_restarts_ids():
    orc = orclib.load()
    dev = L.DeviceBotSort(2, 128, 32, 4)
    for rep in range(2):
        streams = [SynthStream(20, 12, 77 + i, 4) for i in range(2)]
        oracles = [orc.tracker(orclib.BOTSORT) for _ in range(2)]
        for f in range(12):
            dets = np.zeros((2, 32, 6), np.float32)
            embs = np.zeros((2, 32, 4), np.float32)
            cnt = np.zeros(2, np.int32)
            per = []
            for s, st in enumerate(streams):
                d, e = st.next_frame()
                per.append((d, e)); cnt[s] = len(d); dets[s, :len(d)] = d; embs[s, :len(d)] = e
            tables = dev.step(dets, cnt, embs)
            for s in range(2):
                assert np.array_equal(tables[s], oracles[s].update(*per[s])), (rep, f, s)
        dev.reset()
    dev.close()
