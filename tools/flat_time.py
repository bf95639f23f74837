"""Developer tool: the flat drop-in call (host buffers in / out) against the resident solve, metric size."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "M"
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
for rep in range(4):
    t0 = time.perf_counter(); r = lp.simplex(c, A, b, 0.0, None); dt = time.perf_counter() - t0
    print("flat %s: %.2f ms (upload %.2f ms, loop %.2f, final %.2f) status %d" % (name, 1e3 * dt, 1e3 * r.stats["seconds_upload"], 1e3 * r.stats["seconds_pivot_loop"], 1e3 * r.stats["seconds_final_solve"], r.status), flush=True)
cx = lp.Context(); p = cx.upload(c, A, b)
for rep in range(3):
    t0 = time.perf_counter(); r = p.solve(0.0); dt = time.perf_counter() - t0
    print("resident %s: %.2f ms" % (name, 1e3 * dt), flush=True)
cx.close()
