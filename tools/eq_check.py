"""Developer tool: a small equality-constrained MILP tree, oracle vs the pool (children of a non-slack root in the batched schedule)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, bnb
from oracle import oracle as O
me, nv, budget = (int(a) for a in (sys.argv[1:4] + ["60", "100", "15"][len(sys.argv) - 1:]))
rng = np.random.default_rng(78)
x0 = np.abs(rng.standard_normal(nv))
A, G = rng.standard_normal((me, nv)), rng.standard_normal((me, nv))
b, h = A @ x0, G @ x0 + np.abs(rng.standard_normal(me))
c = np.abs(rng.standard_normal(nv))
integ = [j % 5 == 0 for j in range(nv)]
O.set_threads(8)
want = O.solve_milp(c, A, b, G, h, integ, max_nodes=budget)
pool = lp.FrontierPool(workers=4)
got = bnb.solve_milp(c, A, b, G, h, integ, max_nodes=budget, pool=pool)
ws = [nd for nd in want.nodes if nd.status != -1]; gs = [nd for nd in got.nodes if nd.status != -1]
print("nodes", len(ws), len(gs), "errors", want.error, got.error)
bad = 0
for w, g in zip(ws, gs):
    same = g.status == w.status and g.decision == w.decision and (w.status != 0 or abs(g.z - w.z) <= 1e-9 * max(1, abs(w.z)))
    exact = w.status != 0 or (g.z == w.z and np.array_equal(g.x[: len(w.x)], w.x))
    if not same: bad += 1
    print(w.id, w.status, g.status, w.decision, g.decision, "%.15g %.15g" % (w.z, g.z), "ok" if same else "MISMATCH", "exact" if exact else "")
kids = [nd.constraints for nd in gs[1:9]]
res = pool.solve(kids)
print("batched", res.stats["batched_relaxations"], "fallbacks", res.stats["host_fallbacks"], "status", list(res.status), [nd.status for nd in gs[1:9]])
pool.close()
print("MISMATCHES", bad)
