"""Developer tool: the C3 tree (BASELINE config 3) wave by wave — wall time of every pool.solve against the C side's own clock (batched
schedule, supersteps, pivots) and the Python share between the waves.  usage: c3_waves.py [max_nodes]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth, bnb

max_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 127
m3, seed3 = synth.CONFIGS["C3"]
c3, G3, h3 = synth.dense_lp_inequality_form(m3, seed3)
int3 = [j % 4 == 0 for j in range(m3)]
pool = lp.FrontierPool(workers=8)
bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=15, pool=pool)
rows = []
orig = pool.solve
def timed(kids, *a, **k):
    t0 = time.perf_counter(); r = orig(kids, *a, **k); dt = time.perf_counter() - t0
    s = r.stats
    rows.append((len(kids), dt, s["seconds_total"], s["seconds_batch"], s["supersteps"], s["blocks"], s["pivots_phase1"] + s["pivots_phase2"], s["kernel_launches"], int((r.status == 0).sum()), t0))
    return r
pool.solve = timed
t0 = time.perf_counter()
res = bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=max_nodes, pool=pool)
tot = time.perf_counter() - t0
pool.close()
print("nodes %d waves %d total %.1f ms -> %.0f relaxations / s" % (res.relaxations, res.waves, 1e3 * tot, res.relaxations / tot))
prev_end = None
for i, (n, dt, st, sb, ss, bl, pv, kl, ok, ts) in enumerate(rows):
    gap = 0.0 if prev_end is None else ts - prev_end
    prev_end = ts + dt
    print("wave %2d: %2d nodes (%d feasible) wall %.3f ms | C side %.3f batch %.3f ms, supersteps %d blocks %d pivots %d launches %d | python before it %.3f ms" % (i, n, ok, 1e3 * dt, 1e3 * st, 1e3 * sb, ss, bl, pv, kl, 1e3 * gap))
print("sum of waves %.1f ms, C side %.1f ms, batch %.1f ms" % (1e3 * sum(r[1] for r in rows), 1e3 * sum(r[2] for r in rows), 1e3 * sum(r[3] for r in rows)))
