"""BASELINE config 1: SORT on the MOT17-mini pre-dumped detections (det.txt fixtures), oracle on CPU and — on the GPU
box — the HIP path frame-for-frame against it."""
import json
import os

import numpy as np
import pytest

from tests import mot17, orclib
from tests.golden.make_mot17_digest import digest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fixture_shape():
    for seq, n in mot17.SEQS.items():
        frames = mot17.load(seq)
        assert len(frames) == n
        total = sum(f.shape[0] for f in frames)
        assert total == {"MOT17-02-FRCNN": 8186, "MOT17-04-FRCNN": 28406}[seq]
        assert max(f.shape[0] for f in frames) <= 34


def test_oracle_sort_matches_committed_digest(orc):
    want = json.load(open(os.path.join(HERE, "golden", "mot17_sort_digest.json")))
    for seq in mot17.SEQS:
        got = digest(seq, orc.tracker(orclib.SORT, [0.3, 1, 50, 3, 0.3]))
        assert got == want[seq], seq


@pytest.mark.gpu
@pytest.mark.parametrize("kind_g,kind_o,params", [("sort", orclib.SORT, [0.3, 1, 50, 3, 0.3]), ("bytetrack", orclib.BYTETRACK, None),
                                                   ("ocsort", orclib.OCSORT, None)])
def test_gpu_path_on_mot17_mini(orc, kind_g, kind_o, params):
    from motcpp_amd import _lib as L
    for seq in mot17.SEQS:
        tg, to = L.Tracker(kind_g, params), orc.tracker(kind_o, params)
        rows = 0
        for f, d in enumerate(mot17.load(seq)):
            og, oo = tg.update(d), to.update(d)
            assert og.shape == oo.shape, (seq, f)
            assert np.array_equal(og[:, 4:], oo[:, 4:]), (seq, f)
            if kind_g == "ocsort":
                assert np.allclose(og[:, :4], oo[:, :4], rtol=1e-4, atol=1e-3), (seq, f)
            else:
                assert np.array_equal(og, oo), (seq, f)
            rows += og.shape[0]
        assert rows > 0
        tg.close()
