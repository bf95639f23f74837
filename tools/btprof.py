"""Developer tool: per-phase cycle counts of k_bt_inner (s_memtime stamps, enabled by GOMILP_BT_PROF=1).
Run on the GPU box: gpurun -- python tools/btprof.py"""
import sys, os, ctypes as C; sys.path.insert(0,'/root/repo')
os.environ['GOMILP_BT_PROF']='1'
from gomilp_amd import lp, synth
L=lp.lib()
for m,seed in ((512,3),(2048,2)):
    c,A,b=synth.dense_lp_standard_form(m,seed)
    cx=lp.Context(); rl=cx.upload(c,A,b); rl.solve(0.0)
    buf=(C.c_longlong*16)(); L.gomilp_dbg_bt_prof(buf,1)
    r=rl.solve(0.0); L.gomilp_dbg_bt_prof(buf,0)
    n=buf[15]; names=['loop-top','argmin r','column','ratio+argmin','-','row+update','-']
    print(m,'pivots',n, {names[i]: round(buf[i]/n/100*1.0,2) for i in range(7)}, 'sum(us @100MHz?)', round(sum(buf[:7])/n/100,2))
    cx.close()
