"""Throughput of the trackers of SURVEY §8 f3 (host stage machines over the device primitives) in a StreamBatch: S streams stepped in lockstep,
one launch per kernel family and stage. Usage: python tools/bench_f3.py [S] [P M]   -> one JSON line"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from motcpp_amd import _lib as L  # noqa: E402
from motcpp_amd.synth import SynthStream  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    P, M = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 128)
    F, warm = 40, 15
    base = [SynthStream(P, M, 900 + s, 32).frames(F) for s in range(min(S, 16))]
    out = {"streams": S, "shape": [P, M], "frames_timed": F - warm, "frames_per_s": {}}
    for kind, emb in (("strongsort", True), ("deepocsort", True), ("ucmc", False), ("boosttrack", False), ("hybridsort", False)):
        b = L.Batch(kind, S, threads=min(16, S), record_laps=False)
        dets = np.stack([base[s % len(base)][0] for s in range(S)])  # [S, F, M, 6]
        embs = np.stack([base[s % len(base)][1] for s in range(S)]) if emb else None
        cnt = np.full(S, M, np.int32)
        c0 = b.counters()  # (the Device and its counters outlive the batch: differences)
        t0 = None
        for f in range(F):
            if f == warm:
                t0 = time.perf_counter()
            b.step(np.ascontiguousarray(dets[:, f]), cnt, np.ascontiguousarray(embs[:, f]) if emb else None)
        dt = time.perf_counter() - t0
        c = b.counters()
        out["frames_per_s"][kind] = {"value": round(S * (F - warm) / dt), "ms_per_step": round(1e3 * dt / (F - warm), 3),
                                     "flushes_per_step": round((c["flushes"] - c0["flushes"]) / F, 2)}
        b.close()
    # the same trackers through the reference's own surface: T objects of the public classes on T host threads (motcpp_bench_threads), whose
    # concurrent update() calls the library merges into lockstep frames (round 5: run_frame_combined; a per-GPU mutex before)
    F2, warm2 = 40, 15
    dets_t = np.stack([base[s % len(base)][0] for s in range(S)])  # [S, F, M, 6]
    cnt_t = np.full((S, F2), M, np.int32)
    out["basetracker_update_threads"] = {}
    for kind, code in (("deepocsort", 4), ("strongsort", 5), ("ucmc", 6), ("boosttrack", 7), ("hybridsort", 8)):
        for T in (1, S):
            res, _ = L.bench_threads(code, dets_t[:T, :F2], cnt_t[:T], warm2, frames=warm2 + 100)
            out["basetracker_update_threads"].setdefault(kind, {})[f"T{T}"] = {"frames_per_s": round(res["frames_per_s"]), "ms_per_update_p50": round(res["latency_ms_p50"], 3),
                                                                                "ms_per_update_p99": round(res["latency_ms_p99"], 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
