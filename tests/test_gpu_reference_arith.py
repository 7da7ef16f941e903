"""GPU parity against the oracle in the REFERENCE-PLAUSIBLE arithmetic (oracle arith_mode 1), not in the order the kernels happen to use.

The reference's `embedding_distance` (`src/utils/matching.cpp:78-90`) is Eigen `dot()` / `norm()` in an `-O2` build without `-march`
(`CMakeLists.txt:231-236`): SSE2 packets of four lanes, separate multiply and add — no fused multiply-add exists under those flags. The
oracle's default order (mode 0: one k-ordered fmaf chain) is what the fp32 MFMA and the gated VALU kernel compute, so "bit-exact against
mode 0" says the kernels follow the oracle, not that the oracle follows the reference. Mode 1 (`oracle/orc_math.hpp::dot_chain`,
`oracle/orc_kf.hpp`) is the other side: four lane sums with mul-then-add combined pairwise for the dot products, fused multiply-adds in the
small Kalman products, a right-looking Cholesky, row-dot solves, a cofactor inverse. Here the GPU path runs against THAT oracle:

  integer results — every assignment of every stage (index for index), every emitted id, confidence, class and detection index, the list
  of live track ids in list order — `array_equal`;
  floats — output boxes, Kalman means and covariances, smoothed features, cosine distances — within 1e-4 relative (north_star's tolerance).

BoT-SORT at the C3 shape (1024 objects x 512 detections, 256-d embeddings) on the device lifecycle, 48 frames, 8 seeds; the same through the
host lifecycle (which exposes the assignments); DeepOC-SORT (raw dot products into adaptive weights) and StrongSORT (sample library)."""
import numpy as np
import pytest

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative on Kalman state/cost floats"


@pytest.fixture()
def orc1():
    orc = orclib.load()
    orc.set_arith_mode(1)
    try:
        yield orc
    finally:
        orc.set_arith_mode(0)


def rel_close(a, b, scale_floor):
    """|a - b| <= 1e-4 * max(|b|, scale_floor): relative, with a floor at the quantity's natural scale (a pixel for boxes, 1 for
    unit-vector components and cosine distances) so that entries near zero are not compared with a tolerance of zero"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= RTOL * np.maximum(np.abs(b), scale_floor)))


def max_rel(a, b, scale_floor):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), scale_floor))) if a.size else 0.0


def check_states(tag, sg, so):
    """Kalman states `[id | mean(d) | covariance(d x d)]` per live track. A state is a VECTOR: the error that the arithmetic order leaves in it
    comes from the gain times the innovation (hundreds of pixels when a track is re-found far from its prediction, gain entries differing at
    1e-6 relative), so it scales with the vector, not with the entry that happens to be near zero — a lost track drifting through y = 5 px with a
    6e-4 px difference is 1.2e-4 of that ENTRY and 6e-7 of the state. So: every entry within 1e-4 of its vector's largest entry (mean and covariance
    each against their own), and entry by entry (floor: one pixel) for all but one entry in ten thousand. Returns the worst entry-wise figure."""
    assert sg.shape == so.shape and np.array_equal(sg[:, 0], so[:, 0]), tag  # the live tracks' ids in the reference's list order
    if not sg.size:
        return 0.0
    d = 8 if sg.shape[1] == 73 else 7
    assert sg.shape[1] == 1 + d + d * d, sg.shape
    a, b = sg.astype(np.float64), so.astype(np.float64)
    # 8-state filters (cx, cy, a|w, h + velocities; pixels): mean and covariance each against their own largest entry. The 7-state filter
    # carries the box AREA in its state (thousands of square pixels), which would dwarf everything else: there the centre is compared against
    # the centre's scale and the rest entry by entry below.
    blocks = ((1, 1 + d), (1 + d, 1 + d + d * d)) if d == 8 else ((1, 3),)
    for lo, hi in blocks:
        scale = np.maximum(np.abs(b[:, lo:hi]).max(axis=1, keepdims=True), 1.0)
        assert np.all(np.abs(a[:, lo:hi] - b[:, lo:hi]) <= RTOL * scale), (tag, float((np.abs(a[:, lo:hi] - b[:, lo:hi]) / scale).max()))
    ew = np.abs(a[:, 1:] - b[:, 1:]) / np.maximum(np.abs(b[:, 1:]), 1.0)
    assert np.mean(ew > RTOL) <= 1e-4, (tag, float(np.mean(ew > RTOL)), float(ew.max()))
    return float(ew.max())


def check_table(tag, got, want):
    assert got.shape == want.shape, (tag, got.shape, want.shape)
    assert np.array_equal(got[:, 4:], want[:, 4:]), tag  # id, conf, cls, det_ind
    assert rel_close(got[:, :4], want[:, :4], 1.0), (tag, max_rel(got[:, :4], want[:, :4], 1.0))


def test_the_two_orders_differ_on_this_input(orc1):
    """guards the whole file: if mode 1 were silently mode 0 the tests below would be the old bit-exact tests under another name"""
    r = np.random.default_rng(0)
    a, b = r.standard_normal((64, 256)).astype(np.float32), r.standard_normal((48, 256)).astype(np.float32)
    c1 = orc1.cosine_distance(a, b)
    orc1.set_arith_mode(0)
    c0 = orc1.cosine_distance(a, b)
    orc1.set_arith_mode(1)
    assert not np.array_equal(c0, c1)


@pytest.mark.parametrize("n,m,d", [(1024, 512, 256), (300, 170, 128), (65, 33, 96)])
def test_cosine_distances_against_the_sse_order(orc1, n, m, d):
    """`mot_cosine_cost` (fp32 MFMA) and `mot_cosine_cost_gated`'s arithmetic (the same chains) against `matching.cpp:78-90` as an SSE2 build
    computes it: within 1e-4 relative of the distance's scale — measured: a few 1e-7"""
    r = np.random.default_rng(d)
    a, b = r.standard_normal((n, d)).astype(np.float32), r.standard_normal((m, d)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[: m // 3] = (a[: m // 3] + 0.05 * b[: m // 3])  # a third of the columns close to a row, like a re-observed object (unnormalised on purpose)
    ctx = L.Context(0)
    got = ctx.cosine_cost(a, b)
    want = orc1.cosine_distance(a, b)
    assert rel_close(got, want, 1.0), max_rel(got, want, 1.0)
    assert max_rel(got, want, 1.0) < 1e-5  # what a reordered 256-term fp32 reduction can move


@pytest.mark.parametrize("seed", [1234 + k for k in range(8)])
def test_botsort_device_lifecycle_c3_shape(orc1, seed):
    """BASELINE configs[2] on the device lifecycle (`mot_bot_*`: the path `bench.py --workload C3` times), 48 frames per seed"""
    P, M, D, frames = 1024, 512, 256, 48
    dev = L.DeviceBotSort(1, 2048, M, D)
    to = orc1.tracker(orclib.BOTSORT)
    st = SynthStream(P, M, seed, D)
    rows, worst_box, worst_state, worst_feat = 0, 0.0, 0.0, 0.0
    for f in range(frames):
        d, e = st.next_frame()
        dets = np.zeros((1, M, 6), np.float32)
        embs = np.zeros((1, M, D), np.float32)
        dets[0, :len(d)] = d
        embs[0, :len(d)] = e
        tables = dev.step(dets, np.array([len(d)], np.int32), embs, None, None)
        oo = to.update(d, e)
        check_table((seed, f), tables[0], oo)
        rows += oo.shape[0]
        worst_box = max(worst_box, max_rel(tables[0][:, :4], oo[:, :4], 1.0))
        if f % 8 == 7 or f == frames - 1:
            ids, mean, cov, feats, has = dev.dump(0)
            so = to.dump_states()
            assert np.array_equal(ids, so[:, 0].astype(np.int32)), (seed, f)  # the live tracks, in the reference's list order
            check_states((seed, f), np.concatenate([ids[:, None].astype(np.float32), mean, cov.reshape(len(ids), -1)], axis=1), so)
            fo = to.dump_features()
            assert fo.shape[0] == len(ids) and rel_close(feats[has != 0], fo[has != 0], 1.0), (seed, f)
            worst_state = max(worst_state, max_rel(mean, so[:, 1:9], 1.0))
            worst_feat = max(worst_feat, max_rel(feats[has != 0], fo[has != 0], 1.0))
    dev.close()
    assert rows > 0.5 * M * (frames - 3)
    print(f"seed {seed}: {rows} rows, max relative difference boxes {worst_box:.2e}, means {worst_state:.2e}, features {worst_feat:.2e}")


def run_host(orc1, kind_g, kind_o, P, M, D, seed, frames, params=None, feature_check=True):
    """host-lifecycle tracker (assignments of every stage are exposed) against the mode-1 oracle; returns the number of assignments compared"""
    tg, to = L.Tracker(kind_g, params), orc1.tracker(kind_o, params)
    st = SynthStream(P, M, seed, D)
    n_laps = rows = 0
    worst = 0.0
    for f in range(frames):
        d, e = st.next_frame()
        og, oo = tg.update(d, e), to.update(d, e)
        lg, lo = tg.laps(), to.laps()
        assert len(lg) == len(lo), (seed, f, len(lg), len(lo))
        for k, ((xa, ya), (xb, yb)) in enumerate(zip(lg, lo)):
            assert np.array_equal(xa, xb) and np.array_equal(ya, yb), (seed, f, k)
        n_laps += len(lo)
        check_table((seed, f), og, oo)
        rows += oo.shape[0]
        worst = max(worst, check_states((seed, f), tg.dump_states(), to.dump_states()))
        if feature_check and f % 6 == 5:
            fg, fo = tg.dump_features(), to.dump_features()
            assert fg.shape == fo.shape and rel_close(fg, fo, 1.0), (seed, f)
    tg.close()
    assert rows > 0 and n_laps > 0.7 * frames
    print(f"{kind_g} seed {seed}: {n_laps} assignments index for index, {rows} rows, worst entry-wise relative state difference {worst:.2e}")
    return n_laps


@pytest.mark.parametrize("seed", [1234, 1235])
def test_botsort_host_lifecycle_c3_shape_every_assignment(orc1, seed):
    run_host(orc1, "botsort", orclib.BOTSORT, 1024, 512, 256, seed, 48)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_deepocsort_every_assignment(orc1, seed):
    """DeepOC-SORT's cost takes raw inner products (`deepocsort.cpp:351-475`) weighted by the gap between a row's / column's two largest —
    the tracker most exposed to the reduction order"""
    run_host(orc1, "deepocsort", orclib.DEEPOCSORT, 256, 128, 128, seed, 48)


def test_deepocsort_wide_embeddings(orc1):
    run_host(orc1, "deepocsort", orclib.DEEPOCSORT, 400, 220, 256, 5, 40)


@pytest.mark.parametrize("seed", [21, 22])
def test_strongsort_every_assignment(orc1, seed, monkeypatch):
    """the reference's CI mode (`strongsort.cpp:61-76`: tracks confirmed at birth), so that the appearance stage — nearest stored sample by
    inner product — decides assignments from the second frame on"""
    monkeypatch.setenv("GITHUB_ACTIONS", "true")
    run_host(orc1, "strongsort", orclib.STRONGSORT, 150, 90, 128, seed, 40, feature_check=False)
