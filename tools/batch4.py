"""Developer tool: four metric-size LPs through gomilp_frontier_solve_roots — batched schedule against the workers' loop kernels
(knob large_loop) with a cap on each launch's update workgroups (loop_upd)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["M"]
lps = [synth.dense_lp_standard_form(m, seed + 100 + i) for i in range(4)]
for spec in sys.argv[1:]:
    knobs = dict((k, int(v)) for k, v in (kv.split("=") for kv in spec.split(",") if kv))
    pool = lp.FrontierPool(workers=4)
    for k, v in knobs.items(): pool.set(k, v)
    pool.set_root(*lps[0]); roots = [0] + [pool.add_root(*q) for q in lps[1:]]
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); r = pool.solve([[] for _ in roots], roots=roots); best = min(best, time.perf_counter() - t0)
    piv = r.stats["pivots_phase1"] + r.stats["pivots_phase2"]
    print(spec, "pivots", piv, "best %.2f ms" % (1e3 * best), "%.0f k pivots/s" % (piv / best / 1e3), "batched", r.stats["batched_relaxations"], "ok", bool((r.status == 0).all()), flush=True)
    pool.close()
