"""Developer tool: the one badly scaled LP of tests' family on which engine and oracle differed in status (seed 1079): both traces side by side."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gomilp_amd import lp
from oracle import oracle as O
from cond_probe import scaled_lp
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1079
c, A, b = scaled_lp(seed)
o = O.simplex(c, A, b, 0.0, None, trace=True)
cx = lp.Context(); g = cx.upload(c, A, b).solve(0.0, trace=True)
print("oracle: status", o.status, "pivots", list(o.pivots), "basis", list(o.basis), "z", o.z)
print("engine: status", g.status, "pivots", list(getattr(g, "pivots", [])), "basis", list(g.basis) if g.basis is not None else None, "z", g.z)
cx.close()
print("engine stats:", {k: g.stats[k] for k in ("pivots_phase1", "pivots_phase2", "bland_steps", "cond_fallbacks", "pipeline", "phase1_used")})
