import sys; sys.path.insert(0,'/root/repo')
from gomilp_amd import lp, synth
import numpy as np
for name in ('C3','C2','M'):
    m,seed=synth.CONFIGS[name]
    c,A,b=synth.dense_lp_standard_form(m,seed)
    cx=lp.Context(); r=cx.upload(c,A,b).solve(0.0); r=cx.upload(c,A,b).solve(0.0)
    nstruct=int((r.basis < m).sum())
    print(name, m, 'structural in basis', nstruct, 'dense LU steps', r.stats['lu_dense_steps'], 'final dev %.2f ms host %.2f ms'%(r.stats['seconds_final_device']*1e3, r.stats['seconds_final_host']*1e3))
    cx.close()
