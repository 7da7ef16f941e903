mkdir -p gpurun_out /tmp/prof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
dump() { # dir outname
python - <<PY
import sqlite3, glob, csv
dbs=glob.glob('$1/**/*.db', recursive=True)
print('dbs', dbs)
con=sqlite3.connect(dbs[0])
rows=con.execute('select * from top_kernels').fetchall()
cols=[d[0] for d in con.execute('select * from top_kernels').description]
w=csv.writer(open('$R/gpurun_out/$2','w')); w.writerow(cols); w.writerows(rows)
for r in rows[:12]: print(r)
PY
}
rm -rf /tmp/prof/*; timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/c2 -- python $R/bench.py --workload C2 --steps 30 --warmup 40 --streams 4096 --threads 32 --no-cpu-baseline > $R/gpurun_out/r01b_bench_under_rocprof_C2.json 2> /tmp/c2.err; dump /tmp/prof/c2 r01b_kernel_stats_C2.csv
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/ns -- python $R/bench.py --workload NS --steps 8 --warmup 30 --streams 2048 --threads 32 --pipeline 2 --no-cpu-baseline > $R/gpurun_out/r01b_bench_under_rocprof_NS.json 2> /tmp/ns.err; dump /tmp/prof/ns r01b_kernel_stats_NS.csv
# HBM traffic counters, C2 small, separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof/pmc; timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/prof/pmc -o p -- python $R/bench.py --workload C2 --streams 1024 --threads 16 --pipeline 1 --steps 2 --warmup 4 --no-cpu-baseline > /tmp/pmc_$c.json 2>/tmp/pmc.err
  python - <<PY
import csv, collections, glob, json
agg=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob('/tmp/prof/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        k='lap' if 'lap_kernel<' in k else ('kf' if 'kf_kernel' in k else k[:30])
        agg[k]+=float(r['Counter_Value']); n[k]+=1
out={k:{'sum':agg[k],'dispatches':n[k]} for k in agg}
b=json.load(open('/tmp/pmc_$c.json'))
out['_bench']={'problems_per_launch':b['roofline'].get('problems_per_launch'),'lap_launches':b['kernels']['lap']['launches'],'streams':1024}
json.dump(out, open('$R/gpurun_out/r01b_pmc_$c.json','w'), indent=1); print('$c', out)
PY
done
