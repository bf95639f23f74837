"""Developer tool: projects the N-GPU time of the C5 frontier wave by solving the N shards of
frontier.shard_indices one after the other on one GPU."""
import sys, time; sys.path.insert(0,'/root/repo')
from gomilp_amd import lp, synth, frontier
import numpy as np
m,seed=512,3
c,A,b=synth.dense_lp_standard_form(m,seed)
cx=lp.Context(); r0=cx.upload(c,A,b).solve(0.0); cx.close()
mask=synth.integrality_mask(m,m)
children=synth.frontier_children(r0.x,mask,8)
for workers in (4,8,16):
    pool=lp.FrontierPool(workers=workers); pool.set_root(c,A,b)
    pool.solve(children[:32])
    t=time.perf_counter(); pool.solve(children); t1=time.perf_counter()-t
    times=[]
    for world in (2,4,8):
        ts=[]
        for r in range(world):
            idx=frontier.shard_indices(len(children),r,world)
            t=time.perf_counter(); pool.solve([children[i] for i in idx]); ts.append(time.perf_counter()-t)
        times.append((world, round(max(ts)*1e3,2), round(t1/max(ts),2)))
    print('workers',workers,'1gpu ms',round(t1*1e3,2),'shards (world, max ms, speedup):',times)
    pool.close()
