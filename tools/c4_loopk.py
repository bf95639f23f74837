"""Developer tool: one full C4 (or other) solve per loop_k setting: pivots, loop time, z."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
for lk, lg in (tuple(int(v) for v in a.split(":")) for a in (sys.argv[2:] or ["8:8", "16:8", "16:16"])):
    cx = lp.Context(loop_k=lk, loop_g=lg)
    p = cx.upload(c, A, b)
    for rep in range(2):
        r = p.solve(0.0)
        n = r.stats["pivots_phase1"] + r.stats["pivots_phase2"]
        print(name, "loop_k", lk, "loop_g", lg, "status", r.status, "pivots", n, "loop_ms %.2f" % (1e3 * r.stats["seconds_pivot_loop"]),
              "us/pivot %.2f" % (1e6 * r.stats["seconds_pivot_loop"] / max(n, 1)), "z %.17g" % r.z, flush=True)
    cx.close()
