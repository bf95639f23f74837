"""Developer tool: four metric-size LPs through gomilp_frontier_solve_roots, repeated — every repetition's time (a slow one = a wait
inside a final solve's look-ahead launch that gave up: the diagnostic flavour says so with GOMILP_DEBUG_LOOP=1).  usage: batch4_rep.py [reps]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["M"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lps = [synth.dense_lp_standard_form(m, seed + 100 + i) for i in range(4)]
pool = lp.FrontierPool(workers=4)
for kv in sys.argv[2:]:
    k, v = kv.split("="); pool.set(k, int(v))
pool.set_root(*lps[0]); roots = [0] + [pool.add_root(*q) for q in lps[1:]]
for rep in range(reps):
    t0 = time.perf_counter(); r = pool.solve([[] for _ in roots], roots=roots); dt = time.perf_counter() - t0
    piv = r.stats["pivots_phase1"] + r.stats["pivots_phase2"]
    print("rep %d: %.2f ms, %.0f k pivots/s, ok %s" % (rep, 1e3 * dt, piv / dt / 1e3, bool((r.status == 0).all())), flush=True)
pool.close()
