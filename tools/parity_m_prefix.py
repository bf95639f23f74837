"""Developer tool: the first N pivots of the metric LP (2048x4096) against the CPU oracle (whole solve: 20 min of CPU)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from gomilp_amd import lp, synth
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
O.set_threads(16)
m, seed = synth.CONFIGS['M']
c, A, b = synth.dense_lp_standard_form(m, seed)
t0 = time.time()
o = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True, stop_after_pivots=N)
print('oracle', len(o.pivots), 'pivots in %.0f s' % (time.time() - t0), flush=True)
cx = lp.Context(); rl = cx.upload(c, A, b); g = rl.solve(0.0, trace=True)
gp = [(p[0], p[2], p[3], p[4], p[5]) for p in g.pivots[:len(o.pivots)]]
op = [(p[0], p[2], p[3], p[4], p[5]) for p in o.pivots]
print('gpu total pivots', len(g.pivots), 'prefix identical:', gp == op)
cx.close()
