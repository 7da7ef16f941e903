"""One worker of bench.py's all-cores CPU baseline: the oracle tracker (test infrastructure: the CPU restatement of the
reference path) on its own seeded stream for a fixed wall time. Prints `frames seconds_inside_update`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motcpp_amd.synth import SynthStream  # noqa: E402
from tests import orclib  # noqa: E402


def main():
    kind, P, M, D, seed, warm, seconds = (int(a) if i != 6 else float(a) for i, a in enumerate(sys.argv[1:8]))
    trk = orclib.load().tracker(kind)
    s = SynthStream(P, M, seed, D)
    for _ in range(warm):
        d, e = s.next_frame()
        trk.update(d, e)
    n, tc, t0 = 0, 0.0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:  # wall-time bound; only the tracker calls are timed, like the 1-core leg
        d, e = s.next_frame()
        ta = time.perf_counter()
        trk.update(d, e)
        tc += time.perf_counter() - ta
        n += 1
    print(n, tc)


if __name__ == "__main__":
    main()
