"""Developer tool: stats.cond1_final / condinf_final (exact kappa from the tableau) against numpy on the final basis."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
for name in sys.argv[1:] or ["C2"]:
    m, seed = synth.CONFIGS[name]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    cx = lp.Context()
    r = cx.upload(c, A, b).solve(0.0)
    cx.close()
    B = A[:, r.basis]
    print(name, "status", r.status, "kappa1 %.12g (numpy %.12g)" % (r.stats["cond1_final"], np.linalg.cond(B, 1)),
          "kappa_inf %.12g (numpy %.12g)" % (r.stats["condinf_final"], np.linalg.cond(B, np.inf)), flush=True)
