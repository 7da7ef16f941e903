"""Regenerates tests/golden/mot17_sort_digest.json: per-sequence digests of the ORACLE's SORT(0.3,1,50,3,0.3) outputs on
the MOT17-mini detections (BASELINE C1; tools/motcpp_eval.cpp:101-111 parameters). These are oracle outputs — the reference
cannot be executed in this image — kept to detect drift of the oracle itself and checked against the GPU path on the box."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import mot17, orclib  # noqa: E402


def digest(seq, tracker):
    h = hashlib.sha256()
    rows, ids = 0, set()
    for d in mot17.load(seq):
        out = tracker.update(d)
        rows += out.shape[0]
        ids.update(out[:, 4].astype(int).tolist())
        h.update(out.astype("<f4").tobytes())
    return {"rows": rows, "distinct_ids": len(ids), "sha256": h.hexdigest()}


if __name__ == "__main__":
    orc = orclib.load()
    out = {seq: digest(seq, orc.tracker(orclib.SORT, [0.3, 1, 50, 3, 0.3])) for seq in mot17.SEQS}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "mot17_sort_digest.json"), "w"), indent=1)
    print(out)
