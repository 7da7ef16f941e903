import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from motcpp_amd import _lib as L
from motcpp_amd.synth import SynthStream
from tests import orclib
orc = orclib.load()
S=2048; P,M=256,128
dev = L.DeviceByteTrack(S, 512, M)
streams=[SynthStream(P,M,1234+i) for i in range(S)]
for f in range(40):
    dets=np.stack([st.next_frame()[0] for st in streams])
    try:
        out,cnt = dev.step(dets, cap=M)
    except Exception as e:
        print("frame",f,"error",e)
        cnt = dev._cnt
        bad=np.where(cnt<0)[0]; print("streams with overflow", bad[:10], cnt[bad[:10]])
        s=bad[0]
        to=orc.tracker(orclib.BYTETRACK); st=SynthStream(P,M,1234+int(s))
        for g in range(f+1):
            d,_=st.next_frame(); o=to.update(d)
        print("oracle rows for that stream at that frame:", len(o))
        break
    if f%10==0: print(f, cnt.max())
