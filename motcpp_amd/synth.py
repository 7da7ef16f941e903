"""Seeded synthetic detection streams; shared by tests, bench.py and the CPU baseline.

World 1920x1080, P persistent objects on a jittered grid, constant velocity U(-3,3) px/frame plus
N(0, 0.5^2) jitter, w ~ U(30,90), h = w * U(1.8,2.6), reflected at the borders. Every frame emits M of
the P objects (seeded permutation) with N(0,1 px) box noise; 90 % of the emitted detections get
conf ~ U(0.5,1.0), 10 % get U(0.12,0.44) (so ByteTrack's low-score stage runs); cls = 0.
Optional D-dim appearance: one unit vector per object + N(0, 0.05^2) noise, renormalised.
"""
import numpy as np

W, H = 1920.0, 1080.0


class SynthStream:
    def __init__(self, P, M, seed=1234, emb_dim=0):
        self.P, self.M, self.D = int(P), int(M), int(emb_dim)
        self.rng = np.random.Generator(np.random.MT19937(int(seed)))
        r = self.rng
        gx = int(np.ceil(np.sqrt(P * W / H)))
        gy = int(np.ceil(P / gx))
        idx = np.arange(P)
        cx = (idx % gx + 0.5) * (W / gx) + r.uniform(-0.3, 0.3, P) * (W / gx)
        cy = (idx // gx + 0.5) * (H / gy) + r.uniform(-0.3, 0.3, P) * (H / gy)
        self.c = np.stack([cx, cy], 1)
        self.v = r.uniform(-3.0, 3.0, (P, 2))
        self.w = r.uniform(30.0, 90.0, P)
        self.h = self.w * r.uniform(1.8, 2.6, P)
        if self.D:
            e = r.standard_normal((P, self.D))
            self.e = e / np.linalg.norm(e, axis=1, keepdims=True)
        self.frame = 0

    def next_frame(self):
        r = self.rng
        self.c = self.c + self.v + r.normal(0.0, 0.5, (self.P, 2))
        for k, lim in ((0, W), (1, H)):
            lo = self.c[:, k] < 0
            hi = self.c[:, k] > lim
            self.c[lo, k] = -self.c[lo, k]
            self.c[hi, k] = 2 * lim - self.c[hi, k]
            self.v[lo | hi, k] = -self.v[lo | hi, k]
        sel = r.permutation(self.P)[: self.M]
        n = r.normal(0.0, 1.0, (self.M, 4))
        x1 = self.c[sel, 0] - self.w[sel] / 2 + n[:, 0]
        y1 = self.c[sel, 1] - self.h[sel] / 2 + n[:, 1]
        x2 = self.c[sel, 0] + self.w[sel] / 2 + n[:, 2]
        y2 = self.c[sel, 1] + self.h[sel] / 2 + n[:, 3]
        low = r.uniform(0, 1, self.M) < 0.10
        conf = np.where(low, r.uniform(0.12, 0.44, self.M), r.uniform(0.5, 1.0, self.M))
        dets = np.stack([x1, y1, x2, y2, conf, np.zeros(self.M)], 1).astype(np.float32)
        embs = None
        if self.D:
            e = self.e[sel] + r.normal(0.0, 0.05, (self.M, self.D))
            embs = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
        self.frame += 1
        return dets, embs

    def frames(self, n):
        """n frames as arrays: dets [n, M, 6] (and embs [n, M, D] or None)."""
        ds, es = [], []
        for _ in range(n):
            d, e = self.next_frame()
            ds.append(d)
            es.append(e)
        return np.stack(ds), (np.stack(es) if self.D else None)


CONFIGS = {
    # name: (tracker, P, M, emb_dim)
    "C2": ("bytetrack", 256, 128, 0),
    "C3": ("botsort", 1024, 512, 256),
    "C4": ("ocsort", 4096, 2048, 0),
    "C5": ("bytetrack", 1000, 512, 0),
    "NS": ("bytetrack", 1000, 500, 0),
}


# ---- many streams at once (bench.py: tens of thousands of streams x dozens of frames before anything is timed) ----------------------------
def _fill_streams(args):
    """worker: frames of the streams [s0, s1) written into the shared arrays (np.memmap files)"""
    path_d, path_e, F, S, P, M, D, seeds, s0, s1 = args
    dets = np.memmap(path_d, np.float32, "r+", shape=(F, S, M, 6))
    embs = np.memmap(path_e, np.float32, "r+", shape=(F, S, M, D)) if D else None
    for s in range(s0, s1):
        st = SynthStream(P, M, seeds[s], D)
        for f in range(F):
            d, e = st.next_frame()
            dets[f, s] = d
            if D:
                embs[f, s] = e
    dets.flush()
    if D:
        embs.flush()
    return s1 - s0


def generate_streams(P, M, D, seeds, F, workers=1):
    """dets [F, S, M, 6] (and embs [F, S, M, D] or None) of the streams SynthStream(P, M, seeds[s], D), frame by frame — the same arrays whatever
    `workers` is: every stream has its own generator. workers > 1: the streams are dealt to that many fresh interpreter processes (spawn: the
    caller may hold a GPU context) that write into arrays shared through /dev/shm."""
    S = len(seeds)
    total_bytes = 4.0 * F * S * M * (6 + D)
    if workers <= 1 or S * F < 20000 or total_bytes > 48e9:  # (shared-memory files count against the box's memory like the arrays themselves: bounded)
        dets = np.zeros((F, S, M, 6), np.float32)
        embs = np.zeros((F, S, M, D), np.float32) if D else None
        for s in range(S):
            st = SynthStream(P, M, seeds[s], D)
            for f in range(F):
                d, e = st.next_frame()
                dets[f, s] = d
                if D:
                    embs[f, s] = e
        return dets, embs
    import json
    import os
    import subprocess
    import sys
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd_d, path_d = tempfile.mkstemp(prefix="motcpp_synth_d_", dir=base)
    os.close(fd_d)
    path_e = ""
    try:
        np.memmap(path_d, np.float32, "w+", shape=(F, S, M, 6)).flush()
        if D:
            fd_e, path_e = tempfile.mkstemp(prefix="motcpp_synth_e_", dir=base)
            os.close(fd_e)
            np.memmap(path_e, np.float32, "w+", shape=(F, S, M, D)).flush()
        # plain child interpreters (no multiprocessing: nothing of the caller's __main__ is imported or re-run in them)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        per = (S + workers - 1) // workers
        procs = []
        for s0 in range(0, S, per):
            job = json.dumps([path_d, path_e, F, S, P, M, D, [int(x) for x in seeds], s0, min(S, s0 + per)])
            procs.append(subprocess.Popen([sys.executable, "-c", "import sys, json; sys.path.insert(0, sys.argv[1]); from motcpp_amd.synth import _fill_streams; "
                                           "_fill_streams(json.loads(sys.stdin.read()))", root], stdin=subprocess.PIPE))
            procs[-1].stdin.write(job.encode())
            procs[-1].stdin.close()
        for pr in procs:
            if pr.wait(timeout=1800) != 0:
                raise RuntimeError("synthetic-stream worker failed")
        # the mappings outlive the files (unlinked below): no second copy of tens of GB
        dets = np.memmap(path_d, np.float32, "r+", shape=(F, S, M, 6))
        embs = np.memmap(path_e, np.float32, "r+", shape=(F, S, M, D)) if D else None
        return dets, embs
    finally:
        for pth in (path_d, path_e):
            if pth and os.path.exists(pth):
                os.remove(pth)
