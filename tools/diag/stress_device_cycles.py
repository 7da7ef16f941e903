"""Diagnostic (round 6): the GPU suite aborted twice in 19 full runs with glibc's "double free or corruption (!prev)" inside the one-object
motcpp_bench_threads calls of tests/test_gpu_error_isolation.py::test_f3_checksums[deepocsort] — each of those calls creates and destroys a Device
(context, stream, arenas). This repeats exactly that, thousands of times in one process, in the suite's setting (torch imported and initialised first,
so the library binds to the HIP runtime torch ships; faulthandler on, as pytest has it).
  python tools/diag/stress_device_cycles.py [cycles] [faulthandler 0/1] [torch 0/1] [kind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
use_fh = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
use_torch = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
kind = sys.argv[4] if len(sys.argv) > 4 else "deepocsort"
keep_device = (sys.argv[5] != "0") if len(sys.argv) > 5 else False  # hold one tracker for the whole run: the Device (context, stream, arenas) is never destroyed
if use_fh:
    import faulthandler
    faulthandler.enable()
if use_torch:
    import torch
    x = torch.zeros(1 << 20).cuda()
    y = (x + 1).cpu().numpy()
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream

T, F = 16, 24
dets = np.zeros((T, F, 48, 6), np.float32)
counts = np.zeros((T, F), np.int32)
for t in range(T):
    s = SynthStream(60 + 2 * t, 40 - t % 5, 300 + t)
    for f in range(F):
        d, _ = s.next_frame()
        dets[t, f, :len(d)] = d
        counts[t, f] = len(d)
keep = L.Tracker("ucmc") if keep_device else None
t0 = time.time()
res, cs = L.bench_threads(kind, dets, counts, warm=0)
n = 0
while n < cycles:
    if n % 200 == 199:
        res, cs = L.bench_threads(kind, dets, counts, warm=0)  # the 16-thread call in between, as the test has it once per kind
    t = n % T
    _, one = L.bench_threads(kind, dets[t:t + 1], counts[t:t + 1], warm=0)
    assert one[0] == cs[t], (n, t)
    n += 1
print(f"{cycles} one-object calls (a Device each) of {kind}, faulthandler {use_fh}, torch first {use_torch}, device kept {keep_device}: no abort, {time.time() - t0:.1f} s")
