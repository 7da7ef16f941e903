"""Quick throughput probe of the device-lifecycle ByteTrack (mot_bt_*): python tools/bench_device_lifecycle.py P M S [pipe]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream

P, M, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
PIPE = int(sys.argv[4]) if len(sys.argv) > 4 else 1
W, K = 40, 30
F = W + K
G = min(S, 256)  # generate G distinct streams and tile them (throughput probe only)
host = np.zeros((F, G, M, 6), np.float32)
for s in range(G):
    st = SynthStream(P, M, 1234 + s)
    for f in range(F):
        host[f, s] = st.next_frame()[0]
host = np.tile(host, (1, (S + G - 1) // G, 1, 1))[:, :S]
dev = torch.from_numpy(np.ascontiguousarray(host.transpose(0, 1, 3, 2))).cuda(0)
torch.cuda.synchronize()
bounds = [S * p // PIPE for p in range(PIPE + 1)]
bts = [L.DeviceByteTrack(bounds[p + 1] - bounds[p], 2 * P, M) for p in range(PIPE)]
counts = [np.full(bounds[p + 1] - bounds[p], M, np.int32) for p in range(PIPE)]
pools = [ThreadPoolExecutor(1) for _ in range(PIPE)]


def sub(p, f):
    return bts[p].step(resident_ptr=dev.data_ptr() + (f * S + bounds[p]) * 6 * M * 4, counts=counts[p], cap=2 * M)


def step(f):
    for fut in [pools[p].submit(sub, p, f) for p in range(PIPE)]:
        fut.result()


for f in range(W):
    step(f)
t0 = time.perf_counter()
for f in range(W, F):
    step(f)
dt = time.perf_counter() - t0
print(f"P={P} M={M} S={S} pipe={PIPE}: {S * K / dt:.0f} frames/s, {dt / K * 1e3:.2f} ms/step")
