"""Developer tool: the engine alone on the cond_sweep family, per knob setting (compare with the oracle numbers of a logged cond_sweep run)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
from cond_sweep import moderately_scaled_lp, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for knobs in ({"cond_guard": 0}, {"cond_guard": 0, "exact_degenerate": 2}):
    print("knobs", knobs, flush=True)
    for seed, m, dec in cases(n):
        c, A, b = moderately_scaled_lp(seed, m, dec)
        cx = lp.Context(**knobs); root = cx.upload(c, A, b)
        t0 = time.time(); g = root.solve(0.0); dt = time.time() - t0
        mask = [True] * (A.shape[1] - m) + [False] * m
        line = "seed %d m %d dec %.1f root: status %d pivots %d exact %d z %.15g (%.2f s)" % (seed, m, dec, g.status, g.stats["pivots_phase2"], g.stats["cond_fallbacks"], g.z, dt)
        if g.status == 0:
            ch = synth.frontier_children(g.x, mask, 1)[0]
            t0 = time.time(); gc = root.child(ch).solve(0.0); dt = time.time() - t0
            line += " | child: status %d pivots %d+%d bland %d exact %d z %.15g (%.2f s)" % (gc.status, gc.stats["pivots_phase1"], gc.stats["pivots_phase2"], gc.stats["bland_steps"], gc.stats["cond_fallbacks"], gc.z, dt)
        print(line, flush=True)
        cx.close()
