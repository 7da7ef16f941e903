import sys, time; sys.path.insert(0,'.')
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
ctx = L.Context(0)
for P,M in ((256,128),(1000,500)):
    s=SynthStream(P,M,1)
    for fr in range(3):
        d,_=s.next_frame()
        tb=np.stack([s.c[:,0]-s.w/2,s.c[:,1]-s.h/2,s.c[:,0]+s.w/2,s.c[:,1]+s.h/2],1).astype(np.float32)
        tb=tb+np.random.RandomState(fr).randn(*tb.shape).astype(np.float32)*2
        hi=d[d[:,4]>0.45]
        x,y,xv,info=ctx.lap_geom(tb,hi[:,:4],0.8,L.COST_IOU_DIST_FUSE,hi[:,4])
        p=[int(v) for v in ctx._prof]
        print(P,M,'nr',len(tb),'nc',len(hi),'matched',int((x>=0).sum()),'cycles',p[:4],'n_uniq',p[4]//10**6,'bulk_carr',p[4]%10**6,'n_carr',p[5]//10**6,'dummy_cached',(p[5]//1000)%1000,'dummy_scan',p[5]%1000,'n_paths',p[6]//10**6,'bulk_aug',(p[6]//1000)%1000,'aug_walk',p[6]%1000,'n',p[7]//10**6,'general',p[7]%10**6)
