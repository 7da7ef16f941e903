"""Sums rocprofv3 --pmc counter_collection CSVs per kernel into one small JSON (the raw CSVs are too big to bring back from
the GPU box). Run on the box after `rocprofv3 --pmc ... -d <dir>`:  python tools/pmc_aggregate.py <dir> <out.json>"""
import csv, collections, json, sys, glob, os

src = sys.argv[1] if len(sys.argv) > 1 else '/tmp/pmcout'
dst = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/pmc_summary.json'


def short(k):
    for fam in ('lap_sparse_kernel<', 'lap_kernel<', 'kf_kernel<', 'embed_kernel<', 'bt_dups<'):
        if fam in k:
            return fam + k.split(fam)[1].split('>')[0] + '>'
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0][:48]


out = {}
for f in sorted(glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        o = out.setdefault(k, {})
        for c, val in v.items():
            o[c] = {'sum': val, 'dispatches': n[(k, c)]}
json.dump(out, open(dst, 'w'), indent=1)
for k, v in out.items():
    print(k, {c: (round(x['sum'], 1), x['dispatches']) for c, x in v.items()})
