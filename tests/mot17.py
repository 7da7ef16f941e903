"""Reader for the MOT17-mini detection fixtures (tests/golden/MOT17-mini/*.det.txt.gz — data files copied from the
reference's assets/MOT17-mini/train/*/det/det.txt). Row format `frame,-1,x,y,w,h,conf[,...]` -> per-frame N x 6
[x1,y1,x2,y2,conf,cls=0] in FILE ORDER (rows are not sorted by frame; src/data/mot17_dataset.cpp:149-241)."""
import gzip
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SEQS = {"MOT17-02-FRCNN": 600, "MOT17-04-FRCNN": 1050}


def load(seq):
    path = os.path.join(HERE, "golden", "MOT17-mini", seq + ".det.txt.gz")
    rows = np.loadtxt(gzip.open(path, "rt"), delimiter=",", dtype=np.float64)
    frames = {}
    for r in rows:
        x, y, w, h = np.float32(r[2]), np.float32(r[3]), np.float32(r[4]), np.float32(r[5])
        frames.setdefault(int(r[0]), []).append([x, y, np.float32(x + w), np.float32(y + h), np.float32(r[6]), 0.0])
    n = SEQS[seq]
    return [np.asarray(frames.get(f, []), np.float32).reshape(-1, 6) for f in range(1, n + 1)]
