import sys, os, ctypes as C, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import gnsscorr_loader; gc=gnsscorr_loader.load(); import oracle as orc
rng=np.random.default_rng(1)
n=16*8192
data=rng.integers(-60,61,size=(n,2),dtype=np.int8)
for carr,remcarr,remcode in ((0.0,0.0,0.0),(1000.0,0.0,0.0),(1000.0,1.0,0.3),(-2000.0,0.0,0.0)):
    eng=gc.Engine(0); eng.ring_create(1,2,n); eng.ring_push_raw(1,data,n)
    ch=gc.Channel(1,dtype=2,f_if=0.0,corrn=2,corrd=3,corrp=3); eng.set_channels([ch])
    st=dict(carrfreq=carr,codefreq=ch.crate,remcode=remcode,remcarr=remcarr,buffloc=16)
    eng.trk_set_state([st]); eng.trk_run(1); II,QQ,ns=eng.trk_fetch()
    o=orc.make_chan(1,dtype=2,f_if=0.0,corrn=2,corrd=3,corrp=3)
    o.carrfreq,o.codefreq,o.remcode,o.remcarr=carr,ch.crate,remcode,remcarr
    ring=orc.make_ring(data,n,n); orc.lib().orc_sdrtracking(C.byref(o),C.byref(ring),16,1)
    print(carr,remcarr,remcode,'GPU II',II[0,0],'ORC',[o.II[t] for t in range(5)])
    print('   GPU QQ',QQ[0,0],'ORC',[o.QQ[t] for t in range(5)])
    eng.close()
