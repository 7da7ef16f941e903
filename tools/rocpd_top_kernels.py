"""Kernel table (name, calls, total us, average us, % of kernel time) out of a rocprofv3 --kernel-trace run directory: reads
the *_kernel_stats.csv if rocprofv3 wrote one, else the rocpd SQLite database's kernel dispatches.
  python tools/rocpd_top_kernels.py <dir> <out.csv>"""
import csv, glob, os, sqlite3, sys, collections

src, dst = sys.argv[1], sys.argv[2]
rows = []
st = glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True)
if st:
    for r in csv.DictReader(open(st[0])):
        rows.append((r['Name'], int(r['Calls']), float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
else:
    tr = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    if tr:
        for r in csv.DictReader(open(tr[0])):
            a = agg[r['Kernel_Name']]
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    else:
        db = glob.glob(os.path.join(src, '**', '*.db'), recursive=True)[0]
        con = sqlite3.connect(db)
        try:
            q = con.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
            for name, calls, tot, avg, pct in q:
                rows.append((name, int(calls), tot / 1e3, avg / 1e3, pct))
        except sqlite3.Error:
            q = con.execute("select name from sqlite_master where type in ('view','table')").fetchall()
            raise SystemExit('no top_kernels view; objects: ' + ', '.join(x[0] for x in q))
    if agg:
        tot = sum(a[1] for a in agg.values())
        rows = [(k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot) for k, a in agg.items()]
rows.sort(key=lambda r: -r[2])
with open(dst, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['name', 'calls', 'total_us', 'average_us', 'percent'])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2], 1), round(r[3], 2), round(r[4], 2)])
for r in rows[:14]:
    print(f'{r[4]:6.2f}%  {r[3]:10.1f} us x {r[1]:6d}  {r[0][:90]}')
