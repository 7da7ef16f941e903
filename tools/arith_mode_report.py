"""Do the oracle's results hinge on the summation orders it CHOSE for what the reference computes through Eigen (absent here)?
Every stream is tracked twice — arithmetic mode 0 (the canonical orders, which the gfx950 kernels reproduce bit for bit) and mode 1
(fused multiply-adds in the small matrix products, right-looking Cholesky, row-dot triangular solves, cofactor 4x4 inverse, an
SSE-style four-lane dot product; oracle/orc_kf.hpp) — and compared frame by frame: every assignment of every stage index for index,
every emitted id and detection index, and the floats of the output boxes.
  python tools/arith_mode_report.py [frames=200] [seeds=8] [out.json]      (CPU only; the oracle is test infrastructure)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motcpp_amd.synth as sy
from tests import orclib

CONFIGS = {  # name: (tracker kind, P, M, emb_dim, world scale) — C4x: OC-SORT on a world as crowded as C4 (4096 objects on 1920 x 1080)
    "C2": (orclib.BYTETRACK, 256, 128, 0, 1.0), "NS": (orclib.BYTETRACK, 1000, 500, 0, 1.0), "C3": (orclib.BOTSORT, 1024, 512, 256, 1.0),
    "C4x": (orclib.OCSORT, 400, 200, 0, (400 / 4096.0) ** 0.5), "SORT": (orclib.SORT, 256, 128, 0, 1.0),
}


def compare_stream(orc, cfg, seed, frames):
    kind, P, M, D, sc = CONFIGS[cfg]
    w0, h0 = sy.W, sy.H
    sy.W, sy.H = 1920.0 * sc, 1080.0 * sc
    try:
        s = sy.SynthStream(P, M, seed, D)
        t0, t1 = orc.tracker(kind), orc.tracker(kind)
        out = {"frames": 0, "problems": 0, "assignment_mismatches": 0, "id_mismatch_frames": 0, "rows": 0, "max_rel_box_diff": 0.0, "first_divergence": None}
        for f in range(frames):
            d, e = s.next_frame()
            orc.set_arith_mode(0)
            o0 = t0.update(d, e)
            l0 = t0.laps()
            orc.set_arith_mode(1)
            o1 = t1.update(d, e)
            l1 = t1.laps()
            orc.set_arith_mode(0)
            out["frames"] += 1
            out["problems"] += len(l0)
            bad = len(l0) != len(l1) or any(not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])) for a, b in zip(l0, l1))
            out["assignment_mismatches"] += int(bad)
            same_ids = o0.shape == o1.shape and np.array_equal(o0[:, 4], o1[:, 4]) and np.array_equal(o0[:, 7], o1[:, 7])
            out["id_mismatch_frames"] += int(not same_ids)
            if (bad or not same_ids) and out["first_divergence"] is None:
                out["first_divergence"] = f
            if same_ids and o0.size:
                out["rows"] += o0.shape[0]
                rel = np.abs(o0[:, :4] - o1[:, :4]) / np.maximum(np.abs(o0[:, :4]), 1.0)
                out["max_rel_box_diff"] = max(out["max_rel_box_diff"], float(rel.max()))
        return out
    finally:
        sy.W, sy.H = w0, h0
        orc.set_arith_mode(0)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dst = sys.argv[3] if len(sys.argv) > 3 else None
    orc = orclib.load()
    rep = {"frames_per_stream": frames, "seeds": seeds, "modes": "0 = canonical orders (= the kernels), 1 = alternative orders (oracle/orc_kf.hpp)", "configs": {}}
    for cfg in CONFIGS:
        t0 = time.time()
        per = [compare_stream(orc, cfg, 1234 + k, frames) for k in range(seeds)]
        agg = {k: (max(p[k] for p in per) if k == "max_rel_box_diff" else sum(p[k] for p in per)) for k in per[0] if k != "first_divergence"}
        agg["streams_with_a_divergence"] = sum(1 for p in per if p["first_divergence"] is not None)
        agg["seconds"] = round(time.time() - t0, 1)
        rep["configs"][cfg] = agg
        print(cfg, agg, flush=True)
    if dst:
        json.dump(rep, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
