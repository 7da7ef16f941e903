"""North-star shape, one sub-batch of 2048 ByteTrack streams on the device: time of the two assignment launches of a frame (HIP
events) next to the sparse solver's in-kernel cycle counters — how much of a launch is problem latency and how much is occupancy.
Run on the GPU box: python tools/ns_lap_breakdown.py [streams]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P, M, F = 1000, 500, 60
dev = L.DeviceByteTrack(S, 2048, M)
streams = [SynthStream(P, M, 1234 + s) for s in range(min(S, 64))]
host = np.zeros((F, len(streams), 6, M), np.float32)
for f in range(F):
    for s, st in enumerate(streams):
        host[f, s] = st.next_frame()[0].T
tile = np.tile(host, (1, (S + len(streams) - 1) // len(streams), 1, 1))[:, :S]
d = torch.from_numpy(np.ascontiguousarray(tile)).cuda()
rows = L.pinned_array(dev.ctx, (S * M * 2, 8), np.float32)
cnt = L.pinned_array(dev.ctx, (S,), np.int32)
counts = np.full(S, M, np.int32)
ctx = L.Context(0)
for f in range(40):
    dev.step_packed(d.data_ptr() + f * S * 6 * M * 4, counts, rows, cnt)
dev.profile(True)
ctx.lap_fast_stats(reset=True)
for f in range(40, F):
    dev.step_packed(d.data_ptr() + f * S * 6 * M * 4, counts, rows, cnt)
ps = dev.profile_stats()
fs = ctx.lap_fast_stats()
n = ps["frames"]
prob = fs["fast"] + fs["not_unique"]
out = {"streams": S, "frames": n, "lap1_ms_per_launch": ps["lap1_ms"] / n, "lap23_ms_per_launch": ps["lap23_ms"] / n, "frame_ms": ps["frame_ms"] / n,
       "problems": prob, "cycles_per_problem": {k: fs[k] / max(prob, 1) for k in fs if k.startswith("cycles")},
       "lap1_mean_n_m": (ps["lap1_nm"] / max(ps["lap1_problems"], 1)), "lap23_mean_n_m": (ps["lap23_nm"] / max(ps["lap23_problems"], 1))}
print(json.dumps(out, indent=1))
