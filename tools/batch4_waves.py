"""Developer tool: 4 independent metric-size LPs through the pool, wave by wave — which waves are slow and what their stats say.
usage: batch4_waves.py [waves] [workers]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["M"]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pool = lp.FrontierPool(workers=int(sys.argv[2]) if len(sys.argv) > 2 else 4)
lps = [synth.dense_lp_standard_form(m, seed + 100 + i) for i in range(4)]
pool.set_root(*lps[0])
roots = [0] + [pool.add_root(*q) for q in lps[1:]]
for w in range(nw):
    t0 = time.perf_counter(); r = pool.solve([[] for _ in roots], roots=roots); dt = time.perf_counter() - t0
    s = r.stats
    print("wave %2d %.2f ms | C side %.2f batch %.2f busy %.2f | supersteps %d fallbacks %d batched %d | %s" % (
        w, 1e3 * dt, 1e3 * s["seconds_total"], 1e3 * s["seconds_batch"], 1e3 * s["seconds_busy_sum"], s["supersteps"], s["host_fallbacks"], s["batched_relaxations"],
        {k: v for k, v in s.items() if ("retr" in k or "fault" in k or "timeout" in k) }), flush=True)
pool.close()
