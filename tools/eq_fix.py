"""Developer tool: every node of the milp_EQ fixture on the single-relaxation engine and through the pool, against the fixture."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from gomilp_amd import lp, bnb
from gen_golden import eq_problem
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "milp_EQ.npz"))
c, A, b, G, h, integ = eq_problem()
c0, A0, b0 = bnb.convert_to_equalities(c, A, b, G, h)
nodes = [[tuple(t) for t in fx["constraints"][i, : int(fx["ncons"][i])]] for i in range(len(fx["ids"]))]
nodes = [[(int(j), int(s), float(v)) for j, s, v in nd] for nd in nodes]
only = [int(a) for a in os.environ.get("EQ_NODES", "").split(",") if a]
import time
for ed in (int(a) for a in (sys.argv[1:] or ["0", "1"])):
    cx = lp.Context(exact_degenerate=ed)
    root = cx.upload(c0, A0, b0)
    for i, cons in enumerate(nodes):
        if only and i not in only: continue
        print("ed", ed, "node", i, cons, flush=True); t0 = time.time()
        g = root.child(cons).solve(0.0) if cons else root.solve(0.0)
        print("ed", ed, "node", i, "status", g.status, int(fx["status"][i]), "z %.15g %.15g" % (g.z, fx["z"][i]), "p1/p2", g.stats["pivots_phase1"], g.stats["pivots_phase2"], "exact", g.stats["cond_fallbacks"], "%.2f s" % (time.time() - t0), flush=True)
    cx.close()
if os.environ.get("EQ_NOPOOL"): sys.exit(0)
pool = lp.FrontierPool(workers=4)
pool.set_root(c0, A0, b0)
res = pool.solve(nodes)
print("pool status", list(res.status), "batched", res.stats["batched_relaxations"], "fallbacks", res.stats["host_fallbacks"])
print("pool z", ["%.12g" % z for z in res.z])
pool.close()
