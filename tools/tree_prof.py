"""Developer tool: where the C3 tree run spends its time (C call vs Python driver)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth, bnb
m, seed = synth.CONFIGS["C3"]
c, G, h = synth.dense_lp_inequality_form(m, seed)
integ = [j % 4 == 0 for j in range(m)]
orig = lp.FrontierPool.solve
acc = {"t": 0.0, "n": 0, "batch": 0.0, "busy": 0.0, "widths": [], "ss": 0, "blocks": 0, "piv": 0}
def timed(self, children, tol=0.0, roots=None):
    t0 = time.perf_counter(); r = orig(self, children, tol, roots); acc["t"] += time.perf_counter() - t0; acc["n"] += 1
    acc["batch"] += r.stats["seconds_batch"]; acc["busy"] += r.stats["seconds_busy_sum"]; acc["widths"].append(len(children)); acc["ss"] += r.stats["supersteps"]; acc["blocks"] += r.stats["blocks"]; acc["piv"] += r.stats["pivots_phase1"] + r.stats["pivots_phase2"]
    return r
lp.FrontierPool.solve = timed
for rep in range(2):
    for k in acc: acc[k] = [] if k == "widths" else 0
    t0 = time.perf_counter(); r = bnb.solve_milp(c, None, None, G, h, integ, max_nodes=127, workers=4); dt = time.perf_counter() - t0
print("supersteps", acc["ss"], "blocks", acc["blocks"], "pivots", acc["piv"]); print("total %.1f ms; inside pool.solve %.1f ms over %d waves (batch schedule %.1f ms, worker busy %.1f ms); widths %s" % (1e3 * dt, 1e3 * acc["t"], acc["n"], 1e3 * acc["batch"], 1e3 * acc["busy"], acc["widths"]))
