"""BaseTracker::update through the public C++ classes, T tracker objects on T host threads (motcpp_bench_threads): the number a user
of the reference's own surface sees. Usage: python tools/bench_pooled.py [workload] [T ...]  (workload: NS | C2 | SORT | C3 | OC)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from motcpp_amd import _lib as L  # noqa: E402
from motcpp_amd.synth import SynthStream  # noqa: E402

WORK = {"NS": ("bytetrack", 1000, 500, 70, 40), "C2": ("bytetrack", 256, 128, 70, 40), "SORT": ("sort", 256, 128, 50, 20),
        "OC": ("ocsort", 256, 128, 50, 20), "SORTNS": ("sort", 1000, 500, 50, 20), "OCNS": ("ocsort", 1000, 500, 50, 20)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "NS"
    Ts = [int(a) for a in sys.argv[2:]] or [1, 16, 64, 256, 1024]
    kind, P, M, F, warm = WORK[name]
    Tmax = max(Ts)
    base = [SynthStream(P, M, 1234 + t).frames(F)[0] for t in range(min(Tmax, 32))]  # 32 distinct cameras, reused with an offset
    dets = np.zeros((Tmax, F, M, 6), np.float32)
    for t in range(Tmax):
        dets[t] = base[t % len(base)]
        if t >= len(base):
            dets[t, :, :, 0:4:2] += 0.25 * (t // len(base))  # a slightly shifted copy: different floats, same structure
    counts = np.full((Tmax, F), M, np.int32)
    out = {}
    timed = int(__import__("os").environ.get("MOT_POOLED_TIMED_FRAMES", "300"))  # timed update() calls per object (the F frames played back and forth)
    ctx = L.Context(0)
    for T in Ts:
        L.pool_stats(reset=True)
        ctx.lap_fast_stats(reset=True)
        t0 = time.time()
        res, _ = L.bench_threads(kind, dets[:T], counts[:T], warm, frames=warm + timed)
        res["wall_s"] = round(time.time() - t0, 3)
        res["pool"] = L.pool_stats()
        fs = ctx.lap_fast_stats()  # (device-wide counters of the assignment fast path: a declined problem goes to the exact kernel, 2.3 ms at this shape, and its round waits)
        res["assignments"] = {"fast_path": fs["fast"], "declined": sum(v for k, v in fs.items() if k.startswith("declined") or k in ("search_too_large", "certificate_arith", "too_many_tight", "not_unique"))}
        out[f"T{T}"] = res
        print(name, "T =", T, json.dumps(res), file=sys.stderr, flush=True)
    print(json.dumps({"workload": name, "shape": [P, M], "frames": F, "warm": warm, "basetracker_update": out}))


if __name__ == "__main__":
    main()
