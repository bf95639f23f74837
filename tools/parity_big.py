"""Developer tool: a few large parity cases against the (slow) CPU oracle: the 1024-thread kernels and the split
triangular solves (m >= 1024)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
from oracle import oracle as O
O.set_threads(16)
cx = lp.Context()
for (m, seed) in ((640, 11), (900, 12), (1024, 13), (1100, 14), (1300, 15)):
    c, A, b = synth.dense_lp_standard_form(m, seed)
    t0 = time.time()
    o = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True)
    t1 = time.time()
    rl = cx.upload(c, A, b); g = rl.solve(0.0, trace=True); rl.free()
    same = g.status == o.status and [(p[0], p[2], p[3], p[4], p[5]) for p in g.pivots] == [(p[0], p[2], p[3], p[4], p[5]) for p in o.pivots] \
        and np.array_equal(g.basis, o.basis) and np.array_equal(g.x, o.x) and g.z == o.z
    print(m, seed, 'pivots', len(o.pivots), 'oracle %.0f s' % (t1 - t0), 'gpu %.1f ms' % (g.stats['seconds_total'] * 1e3), 'IDENTICAL' if same else 'MISMATCH', flush=True)
cx.close()
