"""Developer tool: one LP of the column-scaled family, oracle vs GPU, with the engine's condition-check prints."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from gomilp_amd import lp
from oracle import oracle as O
from cond_probe import scaled_cols_lp
seed, m = int(sys.argv[1]), int(sys.argv[2])
c, A, b = scaled_cols_lp(seed, m)
o = O.simplex(c, A, b, 0.0, None, trace=True)
print("oracle", o.status, o.pivots_phase2, "z", o.z, flush=True)
cx = lp.Context()
g = cx.upload(c, A, b).solve(0.0, trace=True)
print("gpu", g.status, g.stats["pivots_phase2"], "z", g.z, "fallbacks", g.stats["cond_fallbacks"])
n = min(len(o.pivots), len(g.pivots))
d = [t for t in range(n) if tuple(o.pivots[t])[2:] != tuple(g.pivots[t])[2:]]
print("first differing pivot", d[0] if d else -1, "of", len(o.pivots), len(g.pivots))
if o.basis is not None:
    B = A[:, [int(v) for v in (g.basis if g.basis is not None else o.basis)]]
    print("numpy kappa_1 of the GPU's last basis %.6g" % np.linalg.cond(B, 1))
