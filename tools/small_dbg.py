"""Developer script: one small dense LP on the diagnostic flavour (HIP errors are printed there).  usage: small_dbg.py m seed"""
import os, sys
os.environ.setdefault("GOMILP_DEBUG_BUILD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
m, seed = int(sys.argv[1]), int(sys.argv[2])
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context()
r = cx.upload(c, A, b).solve(0.0)
print("status", r.status, "z", r.z, {k: r.stats[k] for k in ("pivots_phase2", "cond_fallbacks", "lu_rounds", "lu_dense_steps")})
cx.close()
