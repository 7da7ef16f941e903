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
if os.environ.get("MOT_SP_WIDE_ONLY"):  # the cycle counters of the first association's kernel alone (four wavefronts per problem)
    import ctypes as C
    ctx.lib.mot_debug_sparse_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    ctx.lib.mot_debug_sparse_timeline(ctx.h, None, -1)
ctx.lap_fast_stats(reset=True)
for f in range(40, F):
    dev.step_packed(d.data_ptr() + f * S * 6 * M * 4, counts, rows, cnt)
ps = dev.profile_stats()
fs = ctx.lap_fast_stats()
n = ps["frames"]
prob = (fs["fast"] + fs["not_unique"]) // (3 if os.environ.get("MOT_SP_WIDE_ONLY") else 1)
out = {"streams": S, "frames": n, "lap1_ms_per_launch": ps["lap1_ms"] / n, "lap23_ms_per_launch": ps["lap23_ms"] / n, "frame_ms": ps["frame_ms"] / n,
       "problems": prob, "wall_us_per_problem_in_kernel": fs["wall_ticks_kernel_100MHz"] / max(prob, 1) / 100.0,
       "effective_GHz": fs["cycles_kernel"] / max(fs["wall_ticks_kernel_100MHz"], 1) / 10.0, "cycles_per_problem": {k: fs[k] / max(prob, 1) for k in fs if k.startswith("cycles")},
       "searches_per_problem": fs["searches"] / max(prob, 1), "short_searches_per_problem": fs["short_searches"] / max(prob, 1), "retries_per_problem": fs["search_retries"] / max(prob, 1),
       "column_scans_per_problem": fs["column_scans"] / max(prob, 1),
       "lap1_mean_n_m": (ps["lap1_nm"] / max(ps["lap1_problems"], 1)), "lap23_mean_n_m": (ps["lap23_nm"] / max(ps["lap23_problems"], 1))}
print(json.dumps(out, indent=1))
if os.environ.get("MOT_SP_TIMELINE"):  # residence of the last first-association launch's workgroups (diagnostics)
    import ctypes as C
    tl = np.zeros((min(S, 8192), 2), np.uint64)
    ctx.lib.mot_debug_sparse_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    ctx.lib.mot_debug_sparse_timeline(ctx.h, tl.ctypes.data_as(C.c_void_p), len(tl))
    t0 = tl[:, 0].min()
    st, en = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0
    o = np.argsort(st)
    print("workgroup starts (us after the first), sorted: ", np.round(st[o][:: max(1, len(o) // 24)], 1).tolist())
    print("last end", round(float(en.max()), 1), "us; residence mean", round(float((en - st).mean()), 1), "us; starts by block id (first 16):", np.round(st[:16], 1).tolist())
    grid = np.arange(0, en.max(), 5.0)
    print("workgroups resident at t = 0, 5, 10 ... us:", [int(((st <= g) & (en > g)).sum()) for g in grid])
