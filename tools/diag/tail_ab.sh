cd /root/repo
python - <<'P'
import sys, json
sys.path.insert(0, ".")
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
ctx = L.Context(0)
P, M, F, warm = 1000, 500, 70, 40
base = [SynthStream(P, M, 1234 + t).frames(F)[0] for t in range(32)]
dets = np.stack(base)
counts = np.full((32, F), M, np.int32)
ctx.lap_fast_stats(reset=True)
res, _ = L.bench_threads("bytetrack", dets, counts, warm, frames=warm + 300)
fs = ctx.lap_fast_stats()
print({k: v for k, v in fs.items() if not k.startswith("cycles") and v})
P
