"""Per-phase cycle counters of ONE assignment problem (mot_lap_task.prof) at the C2 and north-star shapes: the tool behind
the latency numbers in DESIGN.md §4. Needs a gfx950 GPU: `gpurun -- python tools/lap_phase_cycles.py`."""
import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib
o = orclib.load(); ctx = L.Context(0)
for P,M in ((256,128),(1000,500),(4096,2048)):
    s=SynthStream(P,M,1); d,_=s.next_frame()
    tb=np.stack([s.c[:,0]-s.w/2,s.c[:,1]-s.h/2,s.c[:,0]+s.w/2,s.c[:,1]+s.h/2],1).astype(np.float32)
    hi=d[d[:,4]>0.45]
    for rep in range(2):
        t0=time.time(); x,y,xv,info=ctx.lap_geom(tb,hi[:,:4],0.8,L.COST_IOU_DIST_FUSE,hi[:,4],prof=True); dt=time.time()-t0
    p=ctx._prof
    print(P,M,"host ms",round(dt*1e3,2),"cycles: colmin",p[0],"transfer",p[1],"carr",p[2],"aug",p[3],"| n_uniq",p[4],"n_carr",p[5],"n_paths",p[6],"n",p[7])
    print("   per pass cycles: transfer",p[1]/max(p[4],1),"carr",p[2]/max(p[5],1),"aug",p[3]/max(p[6],1), " total ms @2.4GHz", (p[:4].sum())/2.4e6)
