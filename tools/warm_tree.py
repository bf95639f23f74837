"""Developer tool: the C3 tree (BASELINE config 3) cold and with the opt-in warm start — pivots per node, wall time, agreement node by node.
usage: warm_tree.py [nodes] [dual_budget]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth, bnb
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 127
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m, seed = synth.CONFIGS["C3"]
c, G, h = synth.dense_lp_inequality_form(m, seed)
integ = [j % 4 == 0 for j in range(m)]
pool = lp.FrontierPool(workers=4)
runs = {}
for warm in (False, True, False, True):
    t0 = time.perf_counter()
    r = bnb.solve_milp(c, None, None, G, h, integ, max_nodes=nodes, pool=pool, warm=warm, dual_budget=budget)
    dt = time.perf_counter() - t0
    runs[warm] = r
    print("warm %-5s: %d relaxations, %d waves, %d pivots (%.1f per node; dual %d), warm started %d, handed back %d, %.1f ms = %.0f relaxations/s, result %s" % (
        warm, r.relaxations, r.waves, r.pivots, r.pivots / max(r.relaxations, 1), r.pivots_dual, r.warm_started, r.warm_fallbacks, 1e3 * dt, r.relaxations / dt, r.error or "optimal"), flush=True)
cold, wm = runs[False], runs[True]
cs = [nd for nd in cold.nodes if nd.status != -1]; ws = [nd for nd in wm.nodes if nd.status != -1]
same_dec = sum(1 for a, b in zip(cs, ws) if a.status == b.status and a.decision == b.decision and a.constraints == b.constraints)
zerr = max([abs(a.z - b.z) / max(1.0, abs(a.z)) for a, b in zip(cs, ws) if a.status == 0 and b.status == 0] or [0.0])
bits = sum(1 for a, b in zip(cs, ws) if a.status == 0 and b.status == 0 and a.z == b.z and np.array_equal(a.x, b.x))
print("nodes compared %d: identical status + decision + constraints %d, max relative z difference %.2e, z / x bit-identical %d" % (min(len(cs), len(ws)), same_dec, zerr, bits))
pool.close()
