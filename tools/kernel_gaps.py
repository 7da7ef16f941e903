"""Per-frame picture of a rocprofv3 --kernel-trace CSV of a single-stream run: kernels per frame, summed kernel time, span from the
first kernel's start to the last kernel's end, and the gaps between consecutive kernels (launch / dependency latency).
Usage: python tools/kernel_gaps.py <dir with *_kernel_trace.csv> <first kernel name substring of a frame>"""
import csv
import glob
import sys

import numpy as np


def main():
    d, first = sys.argv[1], sys.argv[2]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    frames = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    frames = frames[len(frames) // 2:]  # steady state
    if not frames:
        print("no frames found", len(rows))
        return
    nk = np.array([len(f) for f in frames])
    busy = np.array([sum(e - s for s, e, _ in f) for f in frames]) / 1e3
    span = np.array([f[-1][1] - f[0][0] for f in frames]) / 1e3
    period = np.array([b[0][0] - a[0][0] for a, b in zip(frames[:-1], frames[1:])]) / 1e3
    print(f"frames {len(frames)}  kernels/frame {nk.mean():.1f}  busy {busy.mean():.1f} us  span {span.mean():.1f} us  period {period.mean():.1f} us")
    f = frames[len(frames) // 2]
    prev = None
    for s, e, n in f:
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"  gap {gap:7.1f} us  dur {(e - s) / 1e3:7.1f} us  {n[:90]}")
        prev = e


if __name__ == "__main__":
    main()
