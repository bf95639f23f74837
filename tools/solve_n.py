"""Developer tool: time full solves of one config under context knobs.  usage: solve_n.py M [reps] [knob=value ...]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "M"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
knobs = dict((k, int(v)) for k, v in (a.split("=") for a in sys.argv[3:]))
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(sample_events=64, chunk=64, **knobs)
p = cx.upload(c, A, b)
for i in range(reps):
    t0 = time.perf_counter(); r = p.solve(0.0); dt = time.perf_counter() - t0
    ks = r.stats["pivot_kernel_seconds"]
    print(name, knobs, "status", r.status, "pivots", r.stats["pivots_phase2"], "total %.2f ms loop %.2f ms" % (1e3 * dt, 1e3 * r.stats["seconds_pivot_loop"]),
          "inner %.2f us/launch update %.2f us/launch" % (1e6 * ks[0] / max(ks[1], 1), 1e6 * ks[2] / max(ks[1], 1)), "z %.17g" % r.z, flush=True)
cx.close()
