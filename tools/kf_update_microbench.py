"""Kalman update (XYAH, predict-first) on a large slab under different gather patterns: how much of the HBM roof the kernel
reaches when the records are contiguous, a random subset in list order (what a tracker's matched tracks look like), or a
random permutation. Run on the GPU box:  python tools/kf_update_microbench.py [tracks_per_stream streams]
MOT_KF_UPDATE_LANE_PER_TRACK=1 selects the one-lane-per-track kernel for comparison."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motcpp_amd import _lib as L


class KfTask(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("cov", C.c_void_p), ("cap", C.c_int32), ("n", C.c_int32), ("src", C.c_void_p), ("dst", C.c_void_p),
                ("flags", C.c_void_p), ("meas", C.c_void_p), ("ldm", C.c_int32), ("midx", C.c_void_p), ("boxes", C.c_void_p), ("ldb", C.c_int32),
                ("q", C.c_float * 3), ("reserved", C.c_int32), ("warp", C.c_float * 9), ("conf", C.c_void_p), ("mean_dense", C.c_void_p), ("cov_blocks", C.c_void_p),
                ("dense_flag", C.c_void_p), ("meas4", C.c_void_p)]  # include/motcpp_amd.h: mot_kf_task


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    ctx = L.Context(0)
    lib = ctx.lib
    lib.mot_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.mot_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]

    def dev(a):
        p = C.c_void_p()
        ctx._chk(lib.mot_malloc(ctx.h, a.nbytes, C.byref(p)))
        ctx._chk(lib.mot_memcpy_h2d(ctx.h, p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        ctx._chk(lib.mot_ctx_sync(ctx.h))
        return p

    r = np.random.default_rng(0)
    rec = np.zeros((S * cap, 72), np.float32)
    rec[:, 0:2] = r.uniform(100, 1000, (S * cap, 2)); rec[:, 2] = 0.5; rec[:, 3] = r.uniform(50, 200, S * cap)
    for k in range(8):
        rec[:, 8 + 9 * k] = 4.0
    slab = dev(rec)
    meas = np.ascontiguousarray(np.stack([rec[:cap, 0] + 1, rec[:cap, 1] - 1, rec[:cap, 2], rec[:cap, 3]], 0))
    d_meas = dev(meas)
    # MOT_KFMB_LIKE_TRACKER=1: what a tracker's launch looks like - a measurement table per stream, records written to OTHER slots than
    # they are read from (the lifecycles keep two halves of a slab), a launch bound well above the items of a task
    like = os.environ.get("MOT_KFMB_LIKE_TRACKER") is not None
    parts = set((os.environ.get("MOT_KFMB_LIKE_TRACKER") or "").split(",")) - {"", "1"} or {"dst", "meas", "bound"}  # (a subset isolates one of them)
    if like:
        d_meas_all = dev(np.ascontiguousarray(np.tile(meas[None], (S, 1, 1))))
    out = {}
    for name, frac, order in (("contiguous_all", 1.0, "id"), ("subset_0.45_list_order", 0.45, "sorted"), ("subset_0.45_random_order", 0.45, "perm")):
        n = int(cap * frac)
        if like and 2 * n > cap:
            continue
        idx = np.arange(n, dtype=np.int32) if order == "id" else np.sort(r.choice(cap, n, replace=False)).astype(np.int32)
        if order == "perm":
            idx = r.permutation(idx).astype(np.int32)
        flags = np.full(n, 0 if like else 8, np.uint8)  # MOT_KF_PREDICT_FIRST (a tracker predicts in its own launch)
        d_idx, d_flags = dev(idx), dev(flags)
        d_midx = dev(r.permutation(cap)[:n].astype(np.int32)) if like else d_idx
        d_dst = dev((np.sort(r.choice(cap // 2, n, replace=False)) + cap // 2).astype(np.int32)) if like else None
        if like:
            idx2 = np.sort(r.choice(cap // 2, n, replace=False)).astype(np.int32); d_idx = dev(idx2)
        tasks = (KfTask * S)()
        for s in range(S):
            t = tasks[s]
            t.mean = slab.value + s * cap * 288; t.cov = t.mean + 32; t.cap = cap; t.n = n; t.src = d_idx.value; t.dst = None
            t.flags = d_flags.value; t.meas = (d_meas_all.value + s * 4 * cap * 4) if (like and 'meas' in parts) else d_meas.value; t.ldm = cap; t.midx = (d_midx if (like and 'meas' in parts) else d_idx).value; t.boxes = None; t.conf = None
            if like and 'dst' in parts: t.dst = d_dst.value
        d_tasks = dev(np.frombuffer(bytes(tasks), np.uint8))
        lib.mot_kf_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        ms = C.c_float()
        for it in range(3):
            ctx._chk(lib.mot_timer_start(ctx.h))
            for _ in range(5):
                ctx._chk(lib.mot_kf_update(ctx.h, L.KF_XYAH, d_tasks, S, (cap if (like and 'bound' in parts) else n)))
            ctx._chk(lib.mot_timer_stop(ctx.h, C.byref(ms)))
        per = ms.value / 5
        out[name] = {"items": S * n, "ms": round(per, 4), "GB/s_records_in_plus_out": round(S * n * 576 / per / 1e6, 1)}
    out["kernel"] = "lane per track" if os.environ.get("MOT_KF_UPDATE_LANE_PER_TRACK") else "lane per covariance row"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
