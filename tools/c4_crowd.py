"""K copies of ONE dumped first-association problem of an OC-SORT 4096 x 2048 run solved in ONE launch of the exact assignment kernel
(mot_lap_solve over a device task array, every copy with its own cost matrix and scratch): how a problem's cycles change when every CU holds
one — same problem, so the launch has no slowest instance. GPU box: python tools/c4_crowd.py c4dumps/lap_0018_1998x3392.bin 1,32,128,256"""
import ctypes as C, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motcpp_amd import _lib as L


class IouTask(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("a", C.c_void_p), ("lda", C.c_int32), ("aidx", C.c_void_p), ("b", C.c_void_p), ("ldb", C.c_int32),
                ("bidx", C.c_void_p), ("bconf", C.c_void_p), ("cost", C.c_void_p), ("ldc", C.c_int32), ("mode", C.c_int32), ("emb", C.c_void_p),
                ("lde", C.c_int32), ("prox_thresh", C.c_float), ("app_thresh", C.c_float), ("fuse", C.c_int32), ("pairs", C.c_void_p),
                ("npairs", C.c_void_p), ("pairs_cap", C.c_int32), ("pair_thresh", C.c_float), ("age_a", C.c_void_p), ("age_b", C.c_void_p),
                ("dup_a", C.c_void_p), ("dup_b", C.c_void_p), ("assoc", C.c_int32), ("frame_diag", C.c_float)]


class LapTask(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("cost", C.c_void_p), ("ldc", C.c_int32), ("thresh", C.c_float), ("x", C.c_void_p), ("y", C.c_void_p),
                ("mode", C.c_int32), ("iou", C.c_void_p), ("ldi", C.c_int32), ("gate", C.c_float), ("xval", C.c_void_p), ("info", C.c_void_p),
                ("work", C.c_void_p), ("geom", IouTask), ("prof", C.c_void_p), ("rowlist", C.c_void_p)]


f = sys.argv[1]
Ks = [int(k) for k in (sys.argv[2] if len(sys.argv) > 2 else "1,32,128,256").split(",")]
nr, nc = map(int, f.split("_")[-1][:-4].split("x"))
raw = np.fromfile(f, np.float32)
th, cost = float(raw[0]), np.ascontiguousarray(raw[1:].reshape(nr, nc))
exp = np.load(f + ".npz")
ctx = L.Context(0)
lib = ctx.lib
lib.mot_lap_work_bytes.restype = C.c_size_t
lib.mot_lap_rowlist_bytes.restype = C.c_size_t
lib.mot_lap_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
wb, rb = int(lib.mot_lap_work_bytes(nr, nc)), int(lib.mot_lap_rowlist_bytes(nr))
dev = torch.device("cuda:0")
c1 = torch.from_numpy(cost).to(dev)
for K in Ks:
    costs = c1.unsqueeze(0).repeat(K, 1, 1).contiguous()
    work = torch.zeros((K, (wb + 255) // 256 * 256), dtype=torch.uint8, device=dev)
    rl = torch.zeros((K, (rb + 255) // 256 * 256), dtype=torch.uint8, device=dev)
    x = torch.zeros((K, nr), dtype=torch.int32, device=dev)
    y = torch.zeros((K, nc), dtype=torch.int32, device=dev)
    prof = torch.zeros((K, 36), dtype=torch.int64, device=dev)
    tasks = (LapTask * K)()
    for k in range(K):
        t = tasks[k]
        t.n, t.m, t.cost, t.ldc, t.thresh = nr, nc, costs[k].data_ptr(), nc, th
        t.x, t.y, t.mode, t.work, t.rowlist, t.prof = x[k].data_ptr(), y[k].data_ptr(), 0, work[k].data_ptr(), rl[k].data_ptr(), prof[k].data_ptr()
    dt = torch.from_numpy(np.frombuffer(bytes(tasks), np.uint8).copy()).to(dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        ctx._chk(lib.mot_lap_solve(ctx.h, C.c_void_p(dt.data_ptr()), K, nr, nc, 0))
        ctx.sync() if hasattr(ctx, "sync") else None
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3
    p = prof.cpu().numpy()
    ok = all(np.array_equal(x[k].cpu().numpy(), exp["x"]) and np.array_equal(y[k].cpu().numpy(), exp["y"]) for k in (0, K - 1))
    tot = p[:, :4].sum(1)
    print("K %4d  launch %.1f ms  %s | Mcycles per problem: mean %.1f min %.1f max %.1f (augmentation mean %.1f) -> %.2f GHz if the slowest spans the launch" %
          (K, ms, "OK" if ok else "MISMATCH", tot.mean() / 1e6, tot.min() / 1e6, tot.max() / 1e6, p[:, 3].mean() / 1e6, tot.max() / ms / 1e6), flush=True)
