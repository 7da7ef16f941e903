"""Misuse of the frames-in-flight interface, per device lifecycle (mot_sort_* / mot_bt_* / mot_oc_* / mot_bot_*): what each call
does when it is made at the wrong time is part of the boundary (include/motcpp_amd.h), so it is tested like the results are:

* reset with two frames pending: both are dropped with the tracks (collect is refused afterwards), the next frames equal those of a tracker after reset()
  (ids keep counting where the reference's do);
* an enqueue whose rows_cap is smaller than the frame's table: the collect reports MOT_ERR_CAPACITY, consumes the frame, and the tracks
  are intact — the following frames (with a large enough rows_cap again) equal the oracle's, which saw every frame;
* a collect into a host buffer smaller than the table: same, MOT_ERR_CAPACITY and nothing lost on the device side;
* a third enqueue, a collect with nothing pending and a synchronous step between enqueue and collect are refused without side effects.
"""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu

KINDS = {"sort": (L.DeviceSort, orclib.SORT), "bytetrack": (L.DeviceByteTrack, orclib.BYTETRACK), "ocsort": (L.DeviceOCSort, orclib.OCSORT),
         "botsort": (L.DeviceBotSort, orclib.BOTSORT)}


def same(which, got, want):
    if got.shape != want.shape or not np.array_equal(got[:, 4:], want[:, 4:]):
        return False
    if which == "ocsort":  # its direction cost goes through acos: boxes within 1e-4 (tests/test_gpu_device_lifecycle.py)
        return np.allclose(got[:, :4], want[:, :4], rtol=1e-4, atol=1e-3)
    return np.array_equal(got, want)


@pytest.mark.parametrize("which", list(KINDS))
def test_misuse_between_enqueue_and_collect(which):
    import torch
    orc = orclib.load()
    cls, kind = KINDS[which]
    shapes = [(60, 40), (128, 96), (12, 12)]
    S, maxd, cap_tracks = len(shapes), 128, 512
    dev = cls(S, cap_tracks, maxd)
    streams = [SynthStream(P, M, 7700 + i) for i, (P, M) in enumerate(shapes)]
    oracles = [orc.tracker(kind) for _ in range(S)]
    F = 34
    RESET_AT = 20  # frames RESET_AT - 2 and RESET_AT - 1 are pending when the reset comes
    soa = np.zeros((F, S, 6, maxd), np.float32)
    counts = np.zeros((F, S), np.int32)
    want = []
    for f in range(F):
        if f == RESET_AT:
            for o in oracles:
                o.reset()  # what BaseTracker::reset does: tracks gone; ids keep counting for SORT / ByteTrack / OC-SORT, restart for BoT-SORT
        per = []
        for s, st in enumerate(streams):
            d, _ = st.next_frame()
            counts[f, s] = len(d)
            soa[f, s, :, :len(d)] = d.T
            per.append(oracles[s].update(d))
        want.append(per)
    ddets = torch.from_numpy(soa).cuda()
    rows = L.pinned_array(dev.ctx, (S * cap_tracks, 8), np.float32)
    small = L.pinned_array(dev.ctx, (4, 8), np.float32)
    cnt = L.pinned_array(dev.ctx, (S,), np.int32)
    cap = rows.shape[0]
    ptr = lambda f: ddets.data_ptr() + f * S * 6 * maxd * 4

    def check(f):
        total = dev.collect_packed(rows, cnt)
        assert total == sum(w.shape[0] for w in want[f]), f
        off = np.concatenate([[0], np.cumsum(cnt)])
        for s in range(S):
            assert same(which, rows[off[s]:off[s + 1]], want[f][s]), (f, s)

    # --- warm up: tracks exist from here on
    for f in range(6):
        dev.enqueue_packed(ptr(f), counts[f].copy(), cap)
        check(f)
    assert sum(w.shape[0] for w in want[5]) > 8

    # --- frame 6 queued with a rows_cap the table does not fit in: the collect says so, the frame is consumed, the tracks are not harmed
    dev.enqueue_packed(ptr(6), counts[6].copy(), 4)
    dev.enqueue_packed(ptr(7), counts[7].copy(), cap)
    with pytest.raises(L.MotError):
        dev.collect_packed(rows, cnt)
    check(7)
    with pytest.raises(L.MotError):
        dev.collect_packed(rows, cnt)  # nothing pending any more

    # --- frame 8 queued properly, collected into a host buffer that is too small: reported, consumed, nothing lost on the device
    dev.enqueue_packed(ptr(8), counts[8].copy(), cap)
    with pytest.raises(L.MotError):
        dev.collect_packed(small, cnt)
    dev.enqueue_packed(ptr(9), counts[9].copy(), cap)
    check(9)

    # --- a third enqueue and a synchronous step between enqueue and collect are refused and change nothing
    dev.enqueue_packed(ptr(10), counts[10].copy(), cap)
    dev.enqueue_packed(ptr(11), counts[11].copy(), cap)
    with pytest.raises(L.MotError):
        dev.enqueue_packed(ptr(12), counts[12].copy(), cap)
    with pytest.raises(L.MotError):
        dev.step_packed(ptr(12), counts[12].copy(), rows, cnt)
    check(10)
    check(11)
    for f in range(12, RESET_AT - 2):
        dev.enqueue_packed(ptr(f), counts[f].copy(), cap)
        check(f)

    # --- reset with two frames pending: they go with the tracks
    dev.enqueue_packed(ptr(RESET_AT - 2), counts[RESET_AT - 2].copy(), cap)
    dev.enqueue_packed(ptr(RESET_AT - 1), counts[RESET_AT - 1].copy(), cap)
    dev.reset()
    with pytest.raises(L.MotError):
        dev.collect_packed(rows, cnt)
    dev.enqueue_packed(ptr(RESET_AT), counts[RESET_AT].copy(), cap)
    for f in range(RESET_AT + 1, F):
        dev.enqueue_packed(ptr(f), counts[f].copy(), cap)
        check(f - 1)
    check(F - 1)
    dev.close()
