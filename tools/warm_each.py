"""Developer tool: per-child pivots cold vs warm for the feasible children of the C5 wave and a few infeasible ones."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
pool = lp.FrontierPool(workers=2); pool.set_root(c, A, b); root = pool.solve_root(0.0)
children = synth.frontier_children(root.x, mask, 8)
res = pool.solve(children)
idx = [i for i in range(256) if res.status[i] == 0] + [3, 7, 100, 255]
for i in idx:
    out = []
    for mode in (0, 1):
        pool.set("warm_start", mode)
        pool.solve([children[i]])
        t0 = time.perf_counter(); r = pool.solve([children[i]]); dt = time.perf_counter() - t0
        out.append((r.stats["pivots_phase1"], r.stats["pivots_phase2"], 1e3 * dt, int(r.status[0]), r.z[0]))
    print(i, "cold piv %d+%d %.2f ms st %d | warm piv %d+%d %.2f ms st %d | dz %.2g" % (out[0][0], out[0][1], out[0][2], out[0][3], out[1][0], out[1][1], out[1][2], out[1][3], abs(out[0][4] - out[1][4]) if out[0][3] == 0 else 0))
pool.close()
