"""ByteTrack with the lifecycle on the device (mot_bt_*, motcpp_amd/csrc/bt_device.hip) against the CPU oracle: output
tables, track ids and the Kalman state of every live track, bit for bit, on seeded streams with ragged and empty frames."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu


def run(shapes, frames, cap, maxd, params=None, check_states_every=7):
    orc = orclib.load()
    S = len(shapes)
    dev = L.DeviceByteTrack(S, cap, maxd, params)
    streams = [SynthStream(P, M, 1234 + i) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(orclib.BYTETRACK, params + [30, 50] if params else None) for _ in range(S)]
    rows = 0
    for f in range(frames):
        dets = np.zeros((S, maxd, 6), np.float32)
        cnt = np.zeros(S, np.int32)
        per = []
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            if (f + s) % 13 == 11:
                d = d[:0]  # an empty frame now and then
            per.append(d)
            cnt[s] = len(d)
            dets[s, :len(d)] = d
        out, oc = dev.step(dets, cnt)
        for s in range(S):
            oo = oracles[s].update(per[s])
            assert oc[s] == oo.shape[0], (f, s, oc[s], oo.shape)
            assert np.array_equal(out[s, :oc[s]], oo), (f, s)
            rows += oo.shape[0]
            if f % check_states_every == check_states_every - 1:
                ids, mean, cov = dev.dump(s)
                so = oracles[s].dump_states()
                assert len(ids) == so.shape[0], (f, s)
                if len(ids):
                    assert np.array_equal(ids, so[:, 0].astype(np.int32)), (f, s)
                    assert np.array_equal(mean, so[:, 1:9]), (f, s)
                    assert np.array_equal(cov.reshape(len(ids), -1), so[:, 9:73]), (f, s)
    assert rows > 0
    dev.close()


def test_small_streams():
    run([(24, 12), (40, 30), (8, 8), (64, 40), (1, 1), (90, 64)], 60, cap=256, maxd=64)


def test_c2_shape():
    run([(256, 128), (256, 128), (200, 100)], 45, cap=512, maxd=128)


def test_custom_thresholds():
    run([(64, 40), (64, 48)], 40, cap=256, maxd=64, params=[0.2, 0.6, 0.7, 10, 30])


def test_capacity_error_is_reported():
    dev = L.DeviceByteTrack(1, 16, 64)
    st = SynthStream(64, 40, 3)
    with pytest.raises(L.MotError):
        for _ in range(5):
            d, _ = st.next_frame()
            dets = np.zeros((1, 64, 6), np.float32)
            dets[0, :len(d)] = d
            dev.step(dets, np.array([len(d)], np.int32))
    dev.close()


def test_north_star_shape():
    run([(1000, 500), (1000, 500)], 24, cap=2048, maxd=512, check_states_every=8)


def test_c5_per_gpu_shape():
    run([(1000, 512), (1000, 512)], 16, cap=2048, maxd=512, check_states_every=8)


def test_reset_drops_the_tracks_and_keeps_counting_ids():
    # ByteTrack::reset (bytetrack.cpp: lists cleared, frame counters zeroed; clear_count() is empty, bytetrack.hpp:38-40): round 4 — the
    # batch-level reset follows it (before, it restarted the ids)
    orc = orclib.load()
    dev = L.DeviceByteTrack(2, 128, 32)
    oracles = [orc.tracker(orclib.BYTETRACK) for _ in range(2)]
    for rep in range(2):
        streams = [SynthStream(20, 12, 77 + i) for i in range(2)]
        if rep:
            for o in oracles:
                o.reset()
        for f in range(12):
            dets = np.zeros((2, 32, 6), np.float32)
            cnt = np.zeros(2, np.int32)
            per = []
            for s, st in enumerate(streams):
                d, _ = st.next_frame()
                per.append(d); cnt[s] = len(d); dets[s, :len(d)] = d
            out, oc = dev.step(dets, cnt)
            for s in range(2):
                assert np.array_equal(out[s, :oc[s]], oracles[s].update(per[s])), (rep, f, s)
        dev.reset()
    dev.close()


def test_long_run_recycles_slots_and_ages_out_lost_tracks():
    # 350 frames: lost tracks age out after track_buffer frames, their slots are reused by later births, ids keep growing
    run([(30, 18), (12, 12), (50, 20)], 350, cap=128, maxd=32, check_states_every=50)


def test_all_streams_empty_then_busy():
    orc = orclib.load()
    dev = L.DeviceByteTrack(3, 64, 16)
    oracles = [orc.tracker(orclib.BYTETRACK) for _ in range(3)]
    st = [SynthStream(10, 8, 500 + i) for i in range(3)]
    for f in range(20):
        dets = np.zeros((3, 16, 6), np.float32)
        cnt = np.zeros(3, np.int32)
        per = []
        for s in range(3):
            d, _ = st[s].next_frame()
            if f < 5 or s == 1:  # the first frames have no detections at all; stream 1 never has any
                d = d[:0]
            per.append(d); cnt[s] = len(d); dets[s, :len(d)] = d
        out, oc = dev.step(dets, cnt)
        for s in range(3):
            assert np.array_equal(out[s, :oc[s]], oracles[s].update(per[s])), (f, s)
    dev.close()


# ---- SORT with the lifecycle on the device (mot_sort_*) ----
def run_sort(shapes, frames, cap, maxd, params=None, check_states_every=9, nan_at=None):
    orc = orclib.load()
    S = len(shapes)
    dev = L.DeviceSort(S, cap, maxd, params)
    streams = [SynthStream(P, M, 4321 + i) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(orclib.SORT, params) for _ in range(S)]
    rows = 0
    for f in range(frames):
        dets = np.zeros((S, maxd, 6), np.float32)
        cnt = np.zeros(S, np.int32)
        per = []
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            if (f + 2 * s) % 11 == 7:
                d = d[:0]
            per.append(d); cnt[s] = len(d); dets[s, :len(d)] = d
        out, oc = dev.step(dets, cnt)
        for s in range(S):
            oo = oracles[s].update(per[s])
            assert oc[s] == oo.shape[0], (f, s, oc[s], oo.shape)
            assert np.array_equal(out[s, :oc[s]], oo), (f, s)
            rows += oo.shape[0]
            if f % check_states_every == check_states_every - 1:
                ids, mean, cov = dev.dump(s)
                so = oracles[s].dump_states()
                assert len(ids) == so.shape[0], (f, s)
                if len(ids):
                    assert np.array_equal(ids, so[:, 0].astype(np.int32)), (f, s)
                    assert np.array_equal(mean, so[:, 1:8]), (f, s)
                    assert np.array_equal(cov, so[:, 8:57]), (f, s)
    assert rows > 0
    dev.close()


def test_sort_device_streams():
    run_sort([(24, 12), (40, 30), (8, 8), (64, 40), (1, 1)], 70, cap=256, maxd=64)


def test_sort_device_c2_shape_and_max_age():
    run_sort([(256, 128), (200, 100)], 40, cap=512, maxd=128)
    run_sort([(30, 20), (30, 12)], 60, cap=128, maxd=32, params=[0.3, 5, 50, 2, 0.2])  # tracks survive 5 missed frames


def test_sort_device_mot17_mini():
    from tests import mot17
    orc = orclib.load()
    for seq in mot17.SEQS:
        dev = L.DeviceSort(1, 256, 64, [0.3, 1, 50, 3, 0.3])
        to = orc.tracker(orclib.SORT, [0.3, 1, 50, 3, 0.3])
        for f, d in enumerate(mot17.load(seq)):
            dets = np.zeros((1, 64, 6), np.float32)
            dets[0, :len(d)] = d
            out, oc = dev.step(dets, np.array([len(d)], np.int32))
            assert np.array_equal(out[0, :oc[0]], to.update(d)), (seq, f)
        dev.close()


def test_packed_output_equals_the_padded_tables():
    """mot_bt_step_packed: the emitted rows of all streams back to back — the same rows, stream after stream, as the oracle's
    tables, with no per-stream row limit (a stream of the C2 shape has been seen emitting 183 rows for 128 detections) —
    and the device-resident copy (mot_bt_device_output) that the multi-GPU gather reads holds the same bytes."""
    import torch

    from motcpp_amd import dist as mdist
    orc = orclib.load()
    shapes = [(40, 30), (256, 128), (8, 8), (90, 64), (256, 128)]
    S, maxd = len(shapes), 128
    dev = L.DeviceByteTrack(S, 512, maxd)
    streams = [SynthStream(P, M, 4321 + i) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(orclib.BYTETRACK) for _ in range(S)]
    rows = L.pinned_array(dev.ctx, (S * maxd * 2, 8), np.float32)
    cnt = L.pinned_array(dev.ctx, (S,), np.int32)
    ddets = torch.zeros((S, 6, maxd), dtype=torch.float32, device="cuda:0")
    comm = None
    for f in range(40):
        counts = np.zeros(S, np.int32)
        soa = np.zeros((S, 6, maxd), np.float32)
        per = []
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            if (f + s) % 11 == 7:
                d = d[:0]
            per.append(d)
            counts[s] = len(d)
            soa[s, :, :len(d)] = d.T
        ddets.copy_(torch.from_numpy(soa))
        torch.cuda.synchronize()
        total = dev.step_packed(ddets.data_ptr(), counts, rows, cnt)
        want = [oracles[s].update(per[s]) for s in range(S)]
        assert total == sum(w.shape[0] for w in want)
        assert np.array_equal(cnt, np.array([w.shape[0] for w in want], np.int32)), f
        off = np.concatenate([[0], np.cumsum(cnt)])
        for s in range(S):
            assert np.array_equal(rows[off[s]:off[s + 1]], want[s]), (f, s)
        if f % 9 == 8 and total:
            r_ptr, o_ptr, c_ptr = dev.device_output()
            dv = torch.device("cuda", 0)
            dr = mdist.device_view(r_ptr, (total, 8), torch.float32, dv)
            do = mdist.device_view(o_ptr, (S + 1,), torch.int32, dv)
            gt, gc = mdist.gather_packed([dr], [mdist.device_view(c_ptr, (S,), torch.int32, dv)], S * maxd * 2)  # world 1: pack only
            assert np.array_equal(do.cpu().numpy(), off.astype(np.int32))
            got = mdist.unpack_packed(gt, gc)
            assert all(np.array_equal(got[s], want[s]) for s in range(S))
            # the native gather (mot_comm_*: RCCL from the library on the context's stream), one-rank communicator
            if comm is None:
                comm = mdist.NativeComm(dev.ctx, world=1, rank=0)
                all_rows = torch.zeros((S * maxd * 2, 8), dtype=torch.float32, device=dv)
            all_rows.zero_()
            torch.cuda.synchronize()
            counts_all, per_rank = comm.gather_tables(r_ptr, c_ptr, S, all_rows.data_ptr(), all_rows.shape[0])
            dev.ctx._chk(dev.ctx.lib.mot_ctx_sync(dev.ctx.h))
            assert per_rank.tolist() == [total] and np.array_equal(counts_all[0], cnt)
            assert np.array_equal(all_rows[:total].cpu().numpy(), rows[:total])
    assert comm is not None
    comm.close()
    dev.close()


def test_two_frames_in_flight_give_the_same_tables():
    """mot_bt_enqueue_packed / mot_bt_collect_packed: one host thread keeps two frames in flight (frame f + 1 is queued before
    frame f is fetched); every frame's packed rows still equal the oracle's tables, in order, and a third enqueue is refused."""
    import torch
    orc = orclib.load()
    shapes = [(40, 30), (256, 128), (8, 8), (90, 64), (200, 128)]
    S, maxd = len(shapes), 128
    dev = L.DeviceByteTrack(S, 512, maxd)
    streams = [SynthStream(P, M, 777 + i) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(orclib.BYTETRACK) for _ in range(S)]
    F = 36
    soa = np.zeros((F, S, 6, maxd), np.float32)
    counts = np.zeros((F, S), np.int32)
    want = []
    for f in range(F):
        per = []
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            if (f + 2 * s) % 9 == 4:
                d = d[:0]
            counts[f, s] = len(d)
            soa[f, s, :, :len(d)] = d.T
            per.append(oracles[s].update(d))
        want.append(per)
    ddets = torch.from_numpy(soa).cuda()
    rows = L.pinned_array(dev.ctx, (S * maxd * 2, 8), np.float32)
    cnt = L.pinned_array(dev.ctx, (S,), np.int32)
    cap = rows.shape[0]

    def check(f):
        total = dev.collect_packed(rows, cnt)
        assert total == sum(w.shape[0] for w in want[f]), f
        off = np.concatenate([[0], np.cumsum(cnt)])
        for s in range(S):
            assert np.array_equal(rows[off[s]:off[s + 1]], want[f][s]), (f, s)

    ptr = lambda f: ddets.data_ptr() + f * S * 6 * maxd * 4
    dev.enqueue_packed(ptr(0), counts[0], cap)
    for f in range(1, F):
        dev.enqueue_packed(ptr(f), counts[f].copy(), cap)
        if f == 5:
            with pytest.raises(L.MotError):
                dev.enqueue_packed(ptr(f), counts[f], cap)  # two frames are pending
        check(f - 1)
    check(F - 1)
    with pytest.raises(L.MotError):
        dev.collect_packed(rows, cnt)  # nothing pending
    # the synchronous call still works afterwards and continues the same streams
    total = dev.step_packed(ptr(F - 1), np.zeros(S, np.int32), rows, cnt)
    assert total >= 0
    dev.close()


@pytest.mark.parametrize("which", ["sort", "ocsort"])
def test_two_frames_in_flight_sort_and_ocsort(which):
    """mot_sort_* / mot_oc_* enqueue_packed / collect_packed (round 3): two frames in flight from one host thread, every frame's packed rows
    equal to the oracle's tables in order (SORT bit for bit; OC-SORT ids / detection indices exactly and boxes within 1e-4: its direction cost
    goes through acos), the device-resident copy holds the same bytes, a third enqueue and a collect with nothing pending are refused,
    and a synchronous step while frames are pending is an error."""
    import torch
    from motcpp_amd import dist as mdist
    orc = orclib.load()
    shapes = [(40, 30), (256, 128), (8, 8), (90, 64), (200, 128)]
    S, maxd, cap_tracks = len(shapes), 128, 1024
    if which == "sort":
        dev, kind = L.DeviceSort(S, cap_tracks, maxd), orclib.SORT
    else:
        dev, kind = L.DeviceOCSort(S, cap_tracks, maxd), orclib.OCSORT
    streams = [SynthStream(P, M, 4100 + i) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(kind) for _ in range(S)]
    F = 30
    soa = np.zeros((F, S, 6, maxd), np.float32)
    counts = np.zeros((F, S), np.int32)
    want = []
    for f in range(F):
        per = []
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            if (f + 2 * s) % 9 == 4:
                d = d[:0]
            counts[f, s] = len(d)
            soa[f, s, :, :len(d)] = d.T
            per.append(oracles[s].update(d))
        want.append(per)
    ddets = torch.from_numpy(soa).cuda()
    rows = L.pinned_array(dev.ctx, (S * cap_tracks, 8), np.float32)
    cnt = L.pinned_array(dev.ctx, (S,), np.int32)
    cap = rows.shape[0]

    def check(f):
        total = dev.collect_packed(rows, cnt)
        assert total == sum(w.shape[0] for w in want[f]), f
        off = np.concatenate([[0], np.cumsum(cnt)])
        for s in range(S):
            got, w = rows[off[s]:off[s + 1]], want[f][s]
            assert got.shape == w.shape and np.array_equal(got[:, 4:], w[:, 4:]), (f, s)
            if which == "sort":
                assert np.array_equal(got, w), (f, s)
            else:
                assert np.allclose(got[:, :4], w[:, :4], rtol=1e-4, atol=1e-3), (f, s)
        if f % 7 == 6 and total:
            r_ptr, o_ptr, c_ptr = dev.device_output()
            dv = torch.device("cuda", 0)
            assert np.array_equal(mdist.device_view(r_ptr, (total, 8), torch.float32, dv).cpu().numpy(), rows[:total])
            assert np.array_equal(mdist.device_view(c_ptr, (S,), torch.int32, dv).cpu().numpy(), cnt)

    ptr = lambda f: ddets.data_ptr() + f * S * 6 * maxd * 4
    dev.enqueue_packed(ptr(0), counts[0], cap)
    for f in range(1, F):
        dev.enqueue_packed(ptr(f), counts[f].copy(), cap)
        if f == 5:
            with pytest.raises(L.MotError):
                dev.enqueue_packed(ptr(f), counts[f], cap)  # two frames are pending
            with pytest.raises(L.MotError):
                dev.step_packed(ptr(f), counts[f], rows, cnt)  # synchronous step while frames are pending
        check(f - 1)
    check(F - 1)
    with pytest.raises(L.MotError):
        dev.collect_packed(rows, cnt)  # nothing pending
    dev.close()

