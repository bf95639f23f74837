"""Developer tool: the degenerate-tree leg of bench.py alone (240 integer-data MILPs, 15 nodes each, every relaxation through the pool's
workers with the exact steps): relaxations / s.  usage: degen_rate.py [reps]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth, bnb
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pool = lp.FrontierPool(workers=4)
fam = [synth.degenerate_integer_milp(sd) for sd in range(240)]
for c, G, h, integ in fam[:8]:
    bnb.solve_milp(c, None, None, G, h, integ, max_nodes=15, pool=pool)
for rep in range(reps):
    t0 = time.perf_counter(); nrel = 0
    for c, G, h, integ in fam:
        nrel += bnb.solve_milp(c, None, None, G, h, integ, max_nodes=15, pool=pool).relaxations
    dt = time.perf_counter() - t0
    print("degenerate trees: %d relaxations in %.3f s = %.0f relaxations / s" % (nrel, dt, nrel / dt), flush=True)
pool.close()
