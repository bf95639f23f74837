"""Developer tool: the C3 tree (node budget 127) cold vs warm-started."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth, bnb
m, seed = synth.CONFIGS["C3"]
c, G, h = synth.dense_lp_inequality_form(m, seed)
integ = [j % 4 == 0 for j in range(m)]
res = {}
for warm in (False, True):
    for rep in range(2):
        t0 = time.perf_counter(); r = bnb.solve_milp(c, None, None, G, h, integ, max_nodes=127, workers=4, warm_start=warm); dt = time.perf_counter() - t0
    res[warm] = r
    print("warm" if warm else "cold", "%.1f ms" % (1e3 * dt), "relaxations", r.relaxations, "waves", r.waves, "pivots", r.pivots, "->", "%.0f relax/s" % (r.relaxations / dt), r.error)
a, b = [n for n in res[False].nodes if n.status != -1], [n for n in res[True].nodes if n.status != -1]
print("nodes", len(a), len(b), "same status", [x.status for x in a] == [x.status for x in b], "same decisions", [x.decision for x in a] == [x.decision for x in b],
      "max |dz| %.3g" % max(abs(x.z - y.z) for x, y in zip(a, b) if x.status == 0 and y.status == 0))
