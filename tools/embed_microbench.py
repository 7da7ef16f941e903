"""The embedding-distance kernel alone (embed_kernel<cosine>, fp32 MFMA) on device-resident features: TFLOP/s against the 157.3 TFLOP/s
fp32 MFMA peak at the C3 shape (1024 x 512 x 256-d per task) for a batch of tasks. GPU box: python tools/embed_microbench.py [tasks] [n m d]"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motcpp_amd import _lib as L


class CosTask(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("d", C.c_int32), ("a", C.c_void_p), ("lda", C.c_int32), ("aidx", C.c_void_p),
                ("b", C.c_void_p), ("ldb", C.c_int32), ("bidx", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int32),
                ("norm_a", C.c_void_p), ("norm_b", C.c_void_p)]


def run(nt=512, n=1024, m=512, d=256, reps=10):
    """nt tasks of an n x m x d cosine-distance matrix in one launch of embed_kernel<cosine> (mot_cosine_cost), device-resident operands"""
    ctx = L.Context(0)
    lib = ctx.lib
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((nt, n, d), device="cuda", generator=g)
    b = torch.randn((nt, m, d), device="cuda", generator=g)
    out = torch.zeros((nt, n, m), device="cuda")
    tasks = (CosTask * nt)()
    for k in range(nt):
        t = tasks[k]
        t.n, t.m, t.d = n, m, d
        t.a, t.lda, t.b, t.ldb = a[k].data_ptr(), d, b[k].data_ptr(), d
        t.out, t.ldo = out[k].data_ptr(), m
    dt = torch.empty(C.sizeof(tasks), dtype=torch.uint8, device="cuda")
    lib.mot_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    ctx._chk(lib.mot_memcpy_h2d(ctx.h, C.c_void_p(dt.data_ptr()), C.cast(tasks, C.c_void_p), C.sizeof(tasks)))
    lib.mot_cosine_cost.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    torch.cuda.synchronize()
    res = {}
    for rep in range(3):
        ctx._chk(lib.mot_cosine_cost(ctx.h, C.c_void_p(dt.data_ptr()), nt, n, m))
    ctx._chk(lib.mot_ctx_sync(ctx.h))
    R = reps
    t0 = time.perf_counter()
    for rep in range(R):
        ctx._chk(lib.mot_cosine_cost(ctx.h, C.c_void_p(dt.data_ptr()), nt, n, m))
    ctx._chk(lib.mot_ctx_sync(ctx.h))
    dtm = (time.perf_counter() - t0) / R
    flops = 2.0 * nt * n * m * d
    # spot check against torch (fp32; not the bit-exact oracle: that is tests/test_gpu_primitives.py)
    ref = 1.0 - (a[0] @ b[0].T) / (a[0].norm(dim=1)[:, None] * b[0].norm(dim=1)[None, :] + 1e-10)
    err = float((out[0] - ref.clamp(min=0)).abs().max())
    res = {"tasks": nt, "n": n, "m": m, "d": d, "ms_per_launch": dtm * 1e3, "TFLOP/s": flops / dtm / 1e12, "mfma_f32_frac": flops / dtm / 1e12 / 157.3,
           "max_abs_diff_vs_torch_fp32": err}
    return res


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n, m, d = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (1024, 512, 256)
    print(json.dumps(run(nt, n, m, d)))


if __name__ == "__main__":
    main()
