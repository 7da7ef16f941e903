"""Stress of reset() on pooled tracker objects: ByteTrack and SORT keep counting ids through reset() (bytetrack.cpp:157-165, sort.cpp:97-100);
an object whose first id after a reset is not above its last id before it lost its counter.  python tools/pooled_reset_stress.py [rounds] [kind]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kind = sys.argv[2] if len(sys.argv) > 2 else "bytetrack"
bad = 0
for r in range(rounds):
    t = L.Tracker(kind, pooled=True)
    s = SynthStream(90, 50, 77 + r % 5, 0)
    top = 0
    for f in range(32):
        d, _ = s.next_frame()
        if f % 11 == 7:
            d = d[:0]
        if f == 25:
            t.reset()
        o = t.update(d, None)
        ids = o[:, 4]
        if f == 25 and len(ids) and ids.min() <= top:
            bad += 1
            print("round", r, "frame", f, "ids after reset start at", ids.min(), "last id before", top, "level", t.pool_level())
        if len(ids):
            top = max(top, ids.max())
    t.close()
print(kind, "rounds", rounds, "lost counters", bad)
