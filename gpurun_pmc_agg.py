import csv, collections, json, sys, glob
out={}
for f in sorted(glob.glob('/tmp/pmcout/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        k='lap_kernel<'+k.split('lap_kernel<')[1].split('>')[0]+'>' if 'lap_kernel<' in k else k[:50]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        o=out.setdefault(k,{})
        for c,val in v.items(): o[c]={'sum':val,'dispatches':n[(k,c)]}
json.dump(out, open('gpurun_out/pmc_summary.json','w'), indent=1)
for k,v in out.items():
    if 'lap' in k: print(k, {c: round(x['sum']/1e6,1) for c,x in v.items()}, 'dispatches', max(x['dispatches'] for x in v.values()))
