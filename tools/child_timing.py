import sys, time; sys.path.insert(0,'/root/repo')
from gomilp_amd import lp, synth
import numpy as np
m,seed=512,3
c,A,b=synth.dense_lp_standard_form(m,seed)
cx=lp.Context()
root=cx.upload(c,A,b); r0=root.solve(0.0)
mask=synth.integrality_mask(m,m)
children=synth.frontier_children(r0.x,mask,8)
for idx in (0,1,255,37):
    for rep in range(3):
        t0=time.perf_counter(); ch=root.child(children[idx]); t1=time.perf_counter(); r=ch.solve(0.0); t2=time.perf_counter(); ch.free(); t3=time.perf_counter()
    s=r.stats
    print(idx, lp.STATUS_NAMES[r.status], 'p1',s['pivots_phase1'],'p2',s['pivots_phase2'],'bland',s['bland_steps'],'launches',s['kernel_launches'],
          'upload_child %.0fus solve %.0fus free %.0fus | loop %.0f final %.0f (dev %.0f host %.0f)' % ((t1-t0)*1e6,(t2-t1)*1e6,(t3-t2)*1e6,s['seconds_pivot_loop']*1e6,s['seconds_final_solve']*1e6,s['seconds_final_device']*1e6,s['seconds_final_host']*1e6))
