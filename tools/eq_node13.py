"""Developer tool: node 13 of the equality-constrained tree (tests/golden/milp_EQ.npz: the REFERENCE ends there with panic:mat.Condition)
on the single-relaxation engine with exact_degenerate = 1 (default for such roots) and 2 (every pivot decided on fresh solves)."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, bnb
from gen_golden import eq_problem
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "milp_EQ.npz"))
c, A, b, G, h, integ = eq_problem()
c0, A0, b0 = bnb.convert_to_equalities(c, A, b, G, h)
k = int(fx["ncons"][13])
cons = [(int(v), int(s), float(r)) for (v, s, r) in fx["constraints"][13, :k]]
print("node 13 constraints", cons, "oracle status", int(fx["status"][13]))
for ed in (1, 2):
    cx = lp.Context(exact_degenerate=ed)
    g = cx.upload(c0, A0, b0).child(cons).solve(0.0)
    print("exact_degenerate %d: status %d (%s) z %.15g pivots %d + %d bland %d exact steps %d refreshes %d" % (
        ed, g.status, lp.STATUS_NAMES.get(g.status), g.z, g.stats["pivots_phase1"], g.stats["pivots_phase2"], g.stats["bland_steps"], g.stats["cond_fallbacks"], g.stats["refreshes"]), flush=True)
    cx.close()
