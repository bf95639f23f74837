import sys, time, threading; sys.path.insert(0,'/root/repo')
from gomilp_amd import lp, synth
import numpy as np
m,seed=synth.CONFIGS['M']
for B in (1,2,4,8):
    probs=[]
    for i in range(B):
        c,A,b=synth.dense_lp_standard_form(m,seed+i)
        cx=lp.Context(chunk=64); probs.append((cx,cx.upload(c,A,b)))
    res=[None]*B
    def work(i):
        res[i]=probs[i][1].solve(0.0)
    for rep in range(2):
        th=[threading.Thread(target=work,args=(i,)) for i in range(B)]
        t=time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; dt=time.perf_counter()-t
    piv=sum(r.stats['pivots_phase2'] for r in res)
    print('concurrent',B,'pivots',piv,'wall %.1f ms'%(dt*1e3),'pivots/s %.0f'%(piv/dt))
    for cx,_ in probs: cx.close()
