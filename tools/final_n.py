"""Developer tool: cost of the bit-exact final solve at one config (stats of Context.solve).  usage: final_n.py C3 [reps] [knob=value ...]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
knobs = {k: int(v) for k, v in (a.split('=') for a in sys.argv[3:])}
cx = lp.Context(**knobs)
p = cx.upload(c, A, b)
for i in range(reps):
    t0 = time.perf_counter(); r = p.solve(0.0); dt = time.perf_counter() - t0
    s = r.stats
    print(name, "total %.3f ms loop %.3f final %.3f (device %.3f host %.3f)" % (1e3 * dt, 1e3 * s["seconds_pivot_loop"], 1e3 * s["seconds_final_solve"], 1e3 * s.get("seconds_final_device", 0), 1e3 * s.get("seconds_final_host", 0)),
          {k: v for k, v in s.items() if k.startswith("lu_") or k.startswith("final")}, flush=True)
cx.close()
