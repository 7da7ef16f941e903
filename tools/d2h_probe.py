"""Which engine carries a large device-to-host copy into page-locked memory on this box (a blit kernel shows up in rocprofv3's kernel trace as
__amd_rocclr_copyBuffer, an SDMA copy does not) and at what rate. GPU box: rocprofv3 --kernel-trace --stats -- python tools/d2h_probe.py"""
import time, torch
n = 30 * 1024 * 1024 // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
h = torch.empty(n, dtype=torch.float32).pin_memory()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        h.copy_(x, non_blocking=True)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        h.copy_(x, non_blocking=True)
    s.synchronize()
    dt = (time.perf_counter() - t0) / 20
print("30 MiB D2H: %.1f us, %.1f GB/s" % (dt * 1e6, n * 4 / dt / 1e9))
