"""Sums rocprofv3 --pmc counter_collection CSVs per kernel into one small JSON (the raw CSVs are too big to bring back from
the GPU box). Run on the box after `rocprofv3 --pmc ... -d /tmp/pmcout`."""
import csv, collections, json, sys, glob
out={}
for f in sorted(glob.glob('/tmp/pmcout/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'lap_kernel<' in k: k='lap_kernel<'+k.split('lap_kernel<')[1].split('>')[0]+'>'
        elif 'kf_kernel<' in k: k='kf_kernel<'+k.split('kf_kernel<')[1].split('>')[0]+'>'
        else: k=k[:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        o=out.setdefault(k,{})
        for c,val in v.items(): o[c]={'sum':val,'dispatches':n[(k,c)]}
json.dump(out, open('gpurun_out/pmc_summary.json','w'), indent=1)
for k,v in out.items(): print(k, {c: (round(x['sum'],1), x['dispatches']) for c,x in v.items()})
