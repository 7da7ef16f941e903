"""Average duration of the LAST n launches of a kernel in a rocprofv3 --kernel-trace CSV (the launches of bench.py's timed region when the command
has no leg behind it): python tools/last_launches_avg.py <trace dir> <kernel name substring> <n>"""
import csv, glob, sys
d, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
last = rows[-n:]
print(f"{name}: {len(rows)} launches; last {len(last)}: average {sum(e - s for s, e in last) / max(len(last), 1) / 1e3:.1f} us, "
      f"all: average {sum(e - s for s, e in rows) / max(len(rows), 1) / 1e3:.1f} us")
