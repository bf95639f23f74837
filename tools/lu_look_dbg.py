"""Developer script: the LU bit test's first case on the diagnostic flavour (HIP errors are printed there)."""
import os, sys
os.environ.setdefault("GOMILP_DEBUG_BUILD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
c, A, b = synth.dense_lp_standard_form(m, 2)
res = []
for blocked in (3, 2, 0):
    cx = lp.Context(lu_blocked=blocked)
    rl = cx.upload(c, A, b)
    r = rl.solve(0.0)
    print("blocked", blocked, "status", r.status, "z", r.z, {k: r.stats[k] for k in ("lu_rounds", "lu_dense_steps")}, flush=True)
    res.append(r)
    rl.free(); cx.close()
print("x equal", [bool(np.array_equal(r.x, res[-1].x)) for r in res[:-1]])
