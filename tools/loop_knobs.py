"""Developer tool: full solves of a config under loop-kernel knob settings ("key=value,key=value" per argument)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1]
m, seed = synth.CONFIGS[name] if name in synth.CONFIGS else (int(name[1:]), 21)   # "m640": a 640-row LP of the same family
c, A, b = synth.dense_lp_standard_form(m, seed)
for spec in sys.argv[2:]:
    knobs = dict((k, int(v)) for k, v in (kv.split("=") for kv in spec.split(",") if kv))
    cx = lp.Context(**knobs)
    p = cx.upload(c, A, b)
    best = 1e9
    for rep in range(4):
        r = p.solve(0.0)
        best = min(best, r.stats["seconds_pivot_loop"])
    n = r.stats["pivots_phase1"] + r.stats["pivots_phase2"]
    print(name, spec, "status", r.status, "pivots", n, "total_ms %.3f" % (1e3 * r.stats["seconds_total"]), "best loop_ms %.3f" % (1e3 * best), "us/pivot %.3f" % (1e6 * best / n), "z %.17g" % r.z, flush=True)
    cx.close()
