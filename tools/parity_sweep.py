"""Developer tool: wide random parity sweep of the GPU path against the CPU oracle (status, pivot sequence, basis, x bits).
   gpurun -- python tools/parity_sweep.py [cases] [max_m]"""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
from oracle import oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
max_m = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12345)
cx = lp.Context()
bad = 0
t0 = time.time()
for k in range(cases):
    m = int(rng.integers(8, max_m)); seed = int(rng.integers(1, 10**6))
    c, A, b = synth.dense_lp_standard_form(m, seed)
    kind = 'root'
    if k % 2 == 1:   # a random child: 1..6 branching rows on fractional variables of the root
        r0 = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True)
        frac = [j for j in range(m) if r0.x[j] != np.floor(r0.x[j])]
        if frac:
            K = int(rng.integers(1, min(6, len(frac)) + 1))
            cons = []
            for j in rng.choice(frac, size=K, replace=False):
                fl = float(np.floor(r0.x[j]))
                cons.append((int(j), 1, fl) if rng.random() < 0.5 else (int(j), -1, -(fl + 1)))
            c, A, b = O.child_standard_form(c, A, b, cons)
            kind = 'child K=%d' % K
    o = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True)
    rl = cx.upload(c, A, b); g = rl.solve(0.0, trace=True); rl.free()
    ok = g.status == o.status
    if ok and o.x is not None:
        ok = [(p[0], p[2], p[3], p[4], p[5]) for p in g.pivots] == [(p[0], p[2], p[3], p[4], p[5]) for p in o.pivots] \
             and np.array_equal(g.basis, o.basis) and np.array_equal(g.x, o.x) and g.z == o.z
    if not ok:
        bad += 1
        print('MISMATCH', k, m, seed, kind, lp.STATUS_NAMES.get(g.status), O.STATUS_NAMES.get(o.status), len(g.pivots), len(o.pivots), flush=True)
print('cases', cases, 'mismatches', bad, '%.1f s' % (time.time() - t0))
cx.close()
