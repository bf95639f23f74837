"""Developer tool: child 1 of the small equality-constrained MILP on the single-relaxation engine, oracle vs GPU traces."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, bnb
from oracle import oracle as O
me, nv = 60, 100
rng = np.random.default_rng(78)
x0 = np.abs(rng.standard_normal(nv))
A, G = rng.standard_normal((me, nv)), rng.standard_normal((me, nv))
b, h = A @ x0, G @ x0 + np.abs(rng.standard_normal(me))
c = np.abs(rng.standard_normal(nv))
integ = [j % 5 == 0 for j in range(nv)]
c0, A0, b0 = O.convert_to_equalities(c, A, b, G, h)
int0 = integ + [False] * (len(c0) - len(c))
O.set_threads(8)
node = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tree = O.solve_milp(c, A, b, G, h, integ, max_nodes=node)
cons = [nd for nd in tree.nodes if nd.id == node][0].constraints
print("node", node, "constraints", cons)
cc, AA, bb = O.child_standard_form(c0, A0, b0, cons)
o = O.simplex(cc, AA, bb, 0.0, None, trace=True)
print("oracle status", o.status, "z %.15g" % o.z, "p1/p2", o.pivots_phase1, o.pivots_phase2)
for ed in (0, 1):
    cx = lp.Context(exact_degenerate=ed)
    g = cx.upload(c0, A0, b0).child(cons).solve(0.0, trace=True)
    print("exact_degenerate", ed, "status", g.status, "z %.15g" % g.z, "p1/p2", g.stats["pivots_phase1"], g.stats["pivots_phase2"], "exact steps", g.stats["cond_fallbacks"])
    n = min(len(o.pivots), len(g.pivots))
    d = [t for t in range(n) if tuple(o.pivots[t])[2:] != tuple(g.pivots[t])[2:] or o.pivots[t][0] != g.pivots[t][0]]
    print("  first differing pivot", d[0] if d else -1, "of", len(o.pivots), len(g.pivots))
    if d:
        t = d[0]
        for u in range(max(0, t - 1), min(n, t + 3)): print("   ", u, tuple(o.pivots[u]), tuple(g.pivots[u]))
    cx.close()
