"""Developer tool: one node of the C1 tree on the GPU engine against the oracle, pivot by pivot."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from gomilp_amd import lp, bnb
from oracle import oracle as O
from gen_golden import c1_problem
node = int(sys.argv[1]) if len(sys.argv) > 1 else 15
knobs = dict((k, int(v)) for k, v in (a.split("=") for a in sys.argv[2:]))
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "milp_C1.npz"))
c, G, h, integ = c1_problem(False)
c0, A0, b0 = O.convert_to_equalities(c, None, None, G, h)
i = list(fx["ids"]).index(node)
k = int(fx["ncons"][i])
cons = [(int(v), int(s), float(r)) for (v, s, r) in fx["constraints"][i, :k]]
print("node", node, "constraints", cons)
cc, AA, bb = O.child_standard_form(c0, A0, b0, cons)
o = O.simplex(cc, AA, bb, 0.0, None, trace=True)
cx = lp.Context(**knobs)
root = cx.upload(c0, A0, b0)
g = root.child(cons).solve(0.0, trace=True)
print("oracle status", o.status, "z %.17g" % o.z, "pivots", o.pivots_phase1, o.pivots_phase2, "bland", o.bland_steps)
print("gpu    status", g.status, "z %.17g" % g.z, "pivots", g.stats["pivots_phase1"], g.stats["pivots_phase2"], "bland", g.stats["bland_steps"], "exact refreshes", g.stats["cond_fallbacks"], "pipeline", g.stats["pipeline"])
for t, (po, pg) in enumerate(zip(o.pivots, g.pivots)):
    print(t, "oracle", tuple(po), "gpu", tuple(pg), "" if tuple(po)[2:] == tuple(pg)[2:] and po[0] == pg[0] else "   <-- differs")
if o.x is not None and g.x is not None:
    print("x equal", np.array_equal(o.x, g.x), "max diff", np.abs(o.x - g.x).max())
    print("oracle x", o.x); print("gpu x   ", g.x)
cx.close()
