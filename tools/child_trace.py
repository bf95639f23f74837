"""Developer tool: time the children of the C5 wave one by one (1 worker) to see the fixed cost per relaxation.
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ct -o ct -- python3 tools/child_trace.py"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS['C5']
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(); root = cx.upload(c, A, b); r0 = root.solve(0.0)
mask = synth.integrality_mask(m, m)
children = synth.frontier_children(r0.x, mask, 8)
for idx in list(range(6)) + [37, 255]:
    for rep in range(2):
        t0 = time.perf_counter(); ch = root.child(children[idx]); t1 = time.perf_counter(); r = ch.solve(0.0); t2 = time.perf_counter(); ch.free(); t3 = time.perf_counter()
    s = r.stats
    print(idx, lp.STATUS_NAMES[r.status], 'p1', s['pivots_phase1'], 'p2', s['pivots_phase2'], 'bland', s['bland_steps'], 'launches', s['kernel_launches'],
          'child %.0f us solve %.0f us free %.0f us | loop %.0f final %.0f (dev %.0f host %.0f)' % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6,
          s['seconds_pivot_loop'] * 1e6, s['seconds_final_solve'] * 1e6, s['seconds_final_device'] * 1e6, s['seconds_final_host'] * 1e6), flush=True)
cx.close()
