"""Developer tool: final-solve time of a config (GOMILP_LUC_SLOTS picks the panel shape in the diagnostic flavour: GOMILP_DEBUG_BUILD=1)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1]
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(); p = cx.upload(c, A, b)
best = 1e9
for rep in range(4):
    r = p.solve(0.0); best = min(best, r.stats["seconds_final_solve"])
print(name, "cfg", os.environ.get("GOMILP_LUC_SLOTS"), "final solve %.3f ms (device %.3f) rounds %d dense %d z %.17g" % (1e3 * best, 1e3 * r.stats["seconds_final_device"], r.stats["lu_rounds"], r.stats["lu_dense_steps"], r.z), flush=True)
cx.close()
