"""Host logic test of the GPU assignment solver: motcpp_amd/csrc/lap_core.hpp (the code the gfx950
kernel runs) executed on T host threads through tests/emu, compared index-for-index with the oracle's
lapjv restatement. Covers the single-step fast path, the general shortest-path search (forced with
quantised costs and with thresholds that make the dummy extension expensive) and odd lane counts."""
import ctypes as C

import numpy as np
import pytest

from tests.emu.build import build_lap_emu


@pytest.fixture(scope="module", params=[False, True, "mem"], ids=["serial_replay", "closed_form_tie_runs", "tie_runs_through_memory"])
def emu(request):
    lib = C.CDLL(build_lap_emu(request.param))

    def run(cost, th, T):
        cost = np.ascontiguousarray(cost, np.float32)
        n, m = cost.shape
        x, y = np.zeros(n, np.int32), np.zeros(m, np.int32)
        lib.emu_lap(cost.ctypes.data_as(C.c_void_p), n, m, m, C.c_float(th), T, x.ctypes.data_as(C.c_void_p),
                    y.ctypes.data_as(C.c_void_p))
        return x, y
    return run


@pytest.fixture(scope="module", params=[False, True], ids=["serial_replay", "closed_form_tie_runs"])
def emu_iou(request):
    lib = C.CDLL(build_lap_emu(request.param))

    def run(a, b, conf, mode, th, T, rpl=0):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        conf = np.ascontiguousarray(conf, np.float32)
        x, y = np.zeros(a.shape[0], np.int32), np.zeros(b.shape[0], np.int32)
        lib.emu_lap_iou(a.ctypes.data_as(C.c_void_p), a.shape[0], b.ctypes.data_as(C.c_void_p), b.shape[0],
                        conf.ctypes.data_as(C.c_void_p), mode, C.c_float(th), T, rpl, x.ctypes.data_as(C.c_void_p),
                        y.ctypes.data_as(C.c_void_p))
        return x, y
    lib.emu_void_real_sweeps.restype = C.c_long
    run.void_real_sweeps = lambda reset=1: lib.emu_void_real_sweeps(reset)
    return run


def test_on_the_fly_cost_functor_matches_materialised_matrix(orc, emu_iou):
    # the solver recomputing fuse(1 - IoU) from boxes must make the same decisions as the oracle on the matrix
    r = np.random.default_rng(2)
    for n, m in [(30, 20), (120, 70), (64, 64)]:
        cx, cy = r.uniform(0, 500, n), r.uniform(0, 300, n)
        w = r.uniform(30, 90, n)
        a = np.stack([cx - w / 2, cy - w, cx + w / 2, cy + w], 1).astype(np.float32)
        b = a[r.permutation(n)[:m] if m <= n else r.integers(0, n, m)] + r.normal(0, 3, (m, 4)).astype(np.float32)
        b[::5] = a[: len(b[::5])]  # exact duplicates -> exact cost ties
        conf = r.uniform(0.3, 1, m).astype(np.float32)
        for mode, th in ((1, 0.7), (2, 0.8), (3, -0.3)):
            cost = {1: orc.iou_distance(a, b), 2: orc.fuse_score(orc.iou_distance(a, b), conf), 3: -orc.iou_batch(a, b)}[mode]
            xo, yo = orc.linear_assignment(cost, th)
            for rpl in (0, 2, 4, 8):  # 0: boxes from memory; 2/4/8: lane-owned register cache (+ leftover columns when 8*rpl < m)
                xe, ye = emu_iou(a, b, conf, mode, th, 8, rpl)
                assert (xo == xe).all() and (yo == ye).all(), (n, m, mode, rpl)


def gen(r, kind, n, m):
    if kind == "dense":
        return r.uniform(0, 1, (n, m)).astype(np.float32), 0.8
    if kind == "dense_forced":  # half > every cost: a genuine min(n,m)-cardinality assignment
        return r.uniform(0, 1, (n, m)).astype(np.float32), 10.0
    if kind == "neg":  # OC-SORT style (ocsort.cpp:700-701)
        return (-r.uniform(0, 1, (n, m))).astype(np.float32), -0.3
    if kind == "quant":  # many exact ties
        return (r.integers(0, 6, (n, m)) / 5.0).astype(np.float32), 0.7
    if kind == "quant_forced":
        return (r.integers(0, 4, (n, m)) / 3.0).astype(np.float32), 5.0
    if kind == "const":
        return np.full((n, m), 0.3, np.float32), 0.8
    c = np.ones((n, m), np.float32)  # IoU-like
    for i in range(n):
        if r.uniform() < 0.8:
            c[i, r.integers(m)] = r.uniform(0.05, 0.6)
    e = r.uniform(0, 1, (n, m)) < 0.03
    c[e] = r.uniform(0.2, 0.95, e.sum()).astype(np.float32)
    return c, 0.8


@pytest.mark.parametrize("kind", ["dense", "dense_forced", "neg", "quant", "quant_forced", "const", "sparse"])
def test_emulated_kernel_matches_oracle(orc, emu, kind):
    r = np.random.default_rng(hash(kind) % 1000)
    for n, m in [(1, 1), (2, 3), (5, 5), (17, 9), (9, 17), (40, 64), (64, 40), (100, 130)]:
        for T in (1, 3, 8):
            c, th = gen(r, kind, n, m)
            xo, yo = orc.linear_assignment(c, th)
            xe, ye = emu(c, th, T)
            assert (xo == xe).all() and (yo == ye).all(), (kind, n, m, T)


def test_emulated_kernel_north_star_shape(orc, emu):
    r = np.random.default_rng(7)
    c, th = gen(r, "sparse", 500, 260)
    xo, yo = orc.linear_assignment(c, th)
    xe, ye = emu(c, th, 8)
    assert (xo == xe).all() and (yo == ye).all()


def test_unmatched_tracks_and_detections_closed_form_runs(orc, emu_iou):
    # tracking-like scenes: most tracks have no detection near them (rows that can only take dummy columns) and many
    # detections start new tracks (dummy rows) — the closed-form run handling of lap_core.hpp must reproduce lapjv's
    # serial displacement chains exactly, including runs interrupted by contested rows and partially filled chunks
    r = np.random.default_rng(11)
    for n, m, near, T, rpl in [(300, 100, 60, 16, 8), (150, 200, 90, 64, 4), (90, 40, 0, 8, 8), (64, 64, 64, 8, 8),
                               (257, 129, 100, 64, 4), (33, 17, 9, 3, 8)]:
        cx, cy = r.uniform(0, 6000, n), r.uniform(0, 4000, n)
        w = r.uniform(30, 90, n)
        a = np.stack([cx - w / 2, cy - w, cx + w / 2, cy + w], 1).astype(np.float32)
        b = np.zeros((m, 4), np.float32)
        pick = r.permutation(n)[:near]
        b[:near] = a[pick] + r.normal(0, 6, (near, 4)).astype(np.float32)
        k = m - near
        fx, fy, fw = r.uniform(0, 6000, k), r.uniform(0, 4000, k), r.uniform(30, 90, k)
        b[near:] = np.stack([fx - fw / 2, fy - fw, fx + fw / 2, fy + fw], 1)
        b = b[r.permutation(m)]
        conf = r.uniform(0.3, 1, m).astype(np.float32)
        for mode, th in ((1, 0.7), (2, 0.8), (1, 0.3)):
            cost = {1: orc.iou_distance(a, b), 2: orc.fuse_score(orc.iou_distance(a, b), conf)}[mode]
            xo, yo = orc.linear_assignment(cost, th)
            xe, ye = emu_iou(a, b, conf, mode, th, T, rpl)
            assert (xo == xe).all() and (yo == ye).all(), (n, m, near, T, rpl, mode, th)


def test_rows_without_an_entry_below_half_skip_their_sweeps(orc, emu_iou):
    # (round 4) the shortest-path search skips the sweep of a real row whose costs are all >= thresh / 2 once a dummy row has been
    # swept with at least its h (lap_core.hpp: nolow). Scenes that force long tied sets through the search: many unmatched tracks,
    # detections that are exact copies of SEVERAL tracks' boxes (duplicated tracks: equal costs, non-unique optima), pile-ups.
    r = np.random.default_rng(5)
    emu_iou.void_real_sweeps(1)
    total = 0
    for trial in range(24):
        T, rpl = ((8, 108), (16, 104), (64, 8), (8, 4), (64, 102), (32, 2))[trial % 6]  # (102 / 2: two columns per lane — the four-wavefront launches)
        cap = (rpl % 100) * T
        n = int(r.integers(60, 400))
        m = int(r.integers(10, min(cap, 200) + 1))
        cx, cy = r.uniform(0, 3000, n), r.uniform(0, 1500, n)
        w, h = r.uniform(30, 80, n), r.uniform(60, 150, n)
        a = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        ndup = n // 6
        a[r.permutation(n)[:ndup]] = a[r.integers(0, n, ndup)]  # duplicated tracks
        src = r.integers(0, n, m)
        b = (a[src] + r.normal(0, 3.0, (m, 4))).astype(np.float32)
        b[::2] = a[src[::2]]  # detections exactly on (possibly duplicated) tracks
        far = r.random(m) < 0.3
        b[far] += 5000.0  # detections with no track near them (dummy rows)
        conf = r.uniform(0.3, 1, m).astype(np.float32)
        for mode, th in ((1, 0.7), (2, 0.8), (1, 0.95)):
            cost = {1: orc.iou_distance(a, b), 2: orc.fuse_score(orc.iou_distance(a, b), conf)}[mode]
            xo, yo = orc.linear_assignment(cost, th)
            xe, ye = emu_iou(a, b, conf, mode, th, T, rpl)
            assert (xo == xe).all() and (yo == ye).all(), (trial, n, m, T, rpl, mode, th)
        total += emu_iou.void_real_sweeps(1)
    assert total > 0  # the path under test ran


def test_sparse_column_minima_make_the_same_decisions(orc, emu_iou):
    # the plain-cost variants (rpl 104 / 108 = register cache 4 / 8 + Cost::kPlain) replace phase 1's sweep over all
    # rows by candidate ranges from the x-sorted rows; everything downstream must come out identical to the oracle on the
    # materialised matrix — with clustered boxes (long candidate lists), exact duplicates (cost ties, equal x1), boxes
    # that touch without intersecting, NaN rows / columns, and every plain cost mode
    r = np.random.default_rng(11)
    T = 8
    cases = 0
    for trial in range(60):
        rpl = (4, 8, 2)[trial % 3]
        n = int(r.integers(32, 16 * T + 1))
        m = int(r.integers(1, rpl * T + 1))
        spread = (60, 400, 2000)[trial % 3]
        cx, cy = r.uniform(0, spread, n), r.uniform(0, spread / 2, n)
        w, h = r.uniform(10, 90, n), r.uniform(20, 160, n)
        a = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        if trial % 4 == 0:
            a = np.round(a / 8) * 8  # a coarse grid: equal x1 among rows, touching boxes
        src = r.integers(0, n, m)
        b = (a[src] + r.normal(0, 2.5, (m, 4))).astype(np.float32)
        b[::3] = a[src[::3]]  # exact copies: exact ties between columns and rows
        if trial % 5 == 1:
            a[r.integers(0, n, 3), r.integers(0, 4, 3)] = np.nan
        if trial % 5 == 2:
            b[r.integers(0, m), r.integers(0, 4)] = np.nan
        if trial % 7 == 3:
            a[:, [0, 2]] -= spread  # negative coordinates
            b[:, [0, 2]] -= spread
        conf = r.uniform(0.3, 1, m).astype(np.float32)
        for mode, th in ((0, 0.3), (1, 0.7), (2, 0.8), (3, -0.3)):
            iou = orc.iou_batch(a, b)
            cost = {0: iou, 1: orc.iou_distance(a, b), 2: orc.fuse_score(orc.iou_distance(a, b), conf), 3: -iou}[mode]
            xo, yo = orc.linear_assignment(cost, th)
            xe, ye = emu_iou(a, b, conf, mode, th, T, 100 + rpl)
            assert (xo == xe).all() and (yo == ye).all(), (trial, n, m, mode, rpl)
            cases += 1
    assert cases == 240


# ---- row lists: the parallel scan steps + sparse real-row sweeps of the shortest-path search (mot_lap_task.rowlist) ----
# the last two: the state the wide matrix launches keep in LDS since round 5 (16-bit y / cols / inv, byte-sized list lengths, the matched
# pairs' costs kept next to y) — lap_emu.cpp::emu_lap_rl16, which also checks that the matched costs are current at the end
@pytest.fixture(scope="module", params=[False, True, "mem", "lds16", "lds16_tie"],
                ids=["serial_replay", "closed_form_tie_runs", "tie_runs_through_memory", "lds16_state", "lds16_state_tie_runs"])
def emu_rl(request):
    state16 = isinstance(request.param, str) and request.param.startswith("lds16")
    lib = C.CDLL(build_lap_emu({"lds16": False, "lds16_tie": True}.get(request.param, request.param)))

    def run(cost, th, T, rowlists=1):
        cost = np.ascontiguousarray(cost, np.float32)
        n, m = cost.shape
        x, y = np.zeros(n, np.int32), np.zeros(m, np.int32)
        if state16 and rowlists:
            rc = lib.emu_lap_rl16(cost.ctypes.data_as(C.c_void_p), n, m, m, C.c_float(th), T, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
            assert rc == 0, "matched costs went stale"
        else:
            lib.emu_lap_rl(cost.ctypes.data_as(C.c_void_p), n, m, m, C.c_float(th), T, rowlists, x.ctypes.data_as(C.c_void_p),
                           y.ctypes.data_as(C.c_void_p))
        return x, y
    return run


@pytest.mark.parametrize("kind", ["dense", "dense_forced", "neg", "quant", "quant_forced", "const", "sparse"])
def test_row_lists_generic_costs(orc, emu_rl, kind):
    # every cost family of the plain test above, now with the row lists in use: short lists, lists that overflow (rows of
    # `neg` / `dense_forced` have most entries below thresh/2 once m > 64), exact ties (events on every step), T = 8 (one
    # real row per step) and 16
    r = np.random.default_rng(hash(kind) % 997)
    for n, m, T in [(9, 17, 8), (64, 40, 16), (90, 110, 8)]:
        if True:
            c, th = gen(r, kind, n, m)
            xo, yo = orc.linear_assignment(c, th)
            xe, ye = emu_rl(c, th, T)
            assert (xo == xe).all() and (yo == ye).all(), (kind, n, m, T)


def ocsort_like_problems(orc, P, M, frames, seed, tmp_path):
    """First-association and rematch problems of an OC-SORT run of the oracle on a world as crowded as BASELINE config C4
    (4096 objects on 1920 x 1080: every box overlaps a dozen others; quirk Q4 leaves exactly duplicated tracks)."""
    import os
    import motcpp_amd.synth as sy
    from tests import orclib
    sc = (P / 4096.0) ** 0.5
    w0, h0 = sy.W, sy.H
    os.environ["ORC_LAP_DUMP"] = str(tmp_path)
    try:
        sy.W, sy.H = 1920.0 * sc, 1080.0 * sc
        trk = orc.tracker(orclib.OCSORT)
        s = sy.SynthStream(P, M, seed)
        for _ in range(frames):
            d, _e = s.next_frame()
            trk.update(d)
    finally:
        sy.W, sy.H = w0, h0
        del os.environ["ORC_LAP_DUMP"]
    out = []
    for f in sorted(os.listdir(tmp_path)):
        nr, nc = map(int, f[:-4].split("_")[-1].split("x"))
        raw = np.fromfile(os.path.join(tmp_path, f), np.float32)
        out.append((float(raw[0]), raw[1:].reshape(nr, nc).copy()))
    return out


def test_row_lists_ocsort_crowded_scene(orc, emu_rl, tmp_path):
    probs = ocsort_like_problems(orc, 110, 56, 8, 5, tmp_path)
    assert len(probs) >= 10
    dup = 0
    for th, c in probs:
        if c.shape[1] > 1:
            dup += int(len({c[:, j].tobytes() for j in range(c.shape[1])}) < c.shape[1])
        xo, yo = orc.linear_assignment(c, th)
        for T in (8, 64):
            xe, ye = emu_rl(c, th, T)
            assert (xo == xe).all() and (yo == ye).all(), (c.shape, T)
        xe, ye = emu_rl(c, th, 16, 0)  # and without the lists: the dense sweeps
        assert (xo == xe).all() and (yo == ye).all(), c.shape
    assert dup >= 1  # problems with exactly duplicated tracks (quirk Q4) are among them
