"""Developer tool: B independent LPs of one config through the device-batched schedule vs one at a time.  conc_n.py M 4"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "M"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m, seed = synth.CONFIGS[name]
lps = [synth.dense_lp_standard_form(m, seed + 100 + i) for i in range(B)]
cx = lp.Context()
single = []
for q in lps:
    p = cx.upload(*q); p.solve(0.0); t0 = time.perf_counter(); r = p.solve(0.0); single.append((time.perf_counter() - t0, r)); p.free()
cx.close()
print("one at a time: %.2f ms total, pivots %d" % (1e3 * sum(t for t, _ in single), sum(r.stats["pivots_phase2"] for _, r in single)))
pool = lp.FrontierPool(workers=min(16, B), sample_batch=int(os.environ.get("SAMPLE", "0"))); pool.set_root(*lps[0]); roots = [0] + [pool.add_root(*q) for q in lps[1:]]
for rep in range(4):
    t0 = time.perf_counter(); res = pool.solve([[] for _ in roots], roots=roots); dt = time.perf_counter() - t0
    piv = res.stats["pivots_phase1"] + res.stats["pivots_phase2"]
    print("batched: %.2f ms pivots %d -> %.0f pivots/s; batch %.2f ms supersteps %d batched %d fallbacks %d" % (1e3 * dt, piv, piv / dt, 1e3 * res.stats["seconds_batch"], res.stats["supersteps"], res.stats["batched_relaxations"], res.stats["host_fallbacks"]),
          "inner %.1f us update %.1f us per block step" % (1e6 * res.stats["seconds_inner_kernels"] / max(1, res.stats["blocks_sampled"]), 1e6 * res.stats["seconds_update_kernels"] / max(1, res.stats["blocks_sampled"])))
same = all(np.array_equal(res.x[i][: lps[i][1].shape[1]], single[i][1].x) and res.z[i] == single[i][1].z for i in range(B))
print("identical to the single path:", same)
pool.close()
