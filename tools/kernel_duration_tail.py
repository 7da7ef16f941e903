"""Per-kernel duration distribution out of a rocprofv3 --kernel-trace run directory (the *_kernel_trace.csv): calls, median, p99, max
and the launches longer than a threshold — what decides a launch sequence that is as slow as its slowest kernel.
  python tools/kernel_duration_tail.py <dir> [threshold_us]"""
import csv, glob, os, sys, collections
import numpy as np
src = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
tr = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
d = collections.defaultdict(list)
first = {}
t0 = None
for r in csv.DictReader(open(tr[0])):
    st = int(r['Start_Timestamp'])
    t0 = st if t0 is None else min(t0, st)
    first.setdefault(r['Kernel_Name'], st)
    first[r['Kernel_Name']] = min(first[r['Kernel_Name']], st)
    d[r['Kernel_Name']].append((int(r['End_Timestamp']) - st) / 1e3)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
print("%-70s %7s %10s %10s %10s %10s %6s %10s" % ("kernel", "calls", "total_ms", "median_us", "p99_us", "max_us", ">thr", "first_ms"))
for k, v in rows[:40]:
    a = np.array(v)
    print("%-70s %7d %10.1f %10.1f %10.1f %10.1f %6d %10.1f" % (k[:70], len(a), a.sum() / 1e3, np.median(a), np.percentile(a, 99), a.max(), (a > thr).sum(), (first[k] - t0) / 1e6))
