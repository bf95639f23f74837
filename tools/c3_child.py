"""Developer tool: the two children of the C3 root (BASELINE config 3) — one alone / both, through the batched schedule with and without the
loop kernel, and on the single-relaxation engine: pivots per phase, time of the batch part.  usage: c3_child.py [reps]"""
import sys, time, os, math; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth, bnb
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m3, seed3 = synth.CONFIGS["C3"]
c3, G3, h3 = synth.dense_lp_inequality_form(m3, seed3)
int3 = [j % 4 == 0 for j in range(m3)]
c0, A0, b0 = bnb.convert_to_equalities(c3, None, None, G3, h3)
cx = lp.Context(); rootp = cx.upload(c0, A0, b0); root = rootp.solve(0.0)
j = bnb.max_fun_branch_point(c0, int3 + [False] * (len(c0) - len(c3)))
fl = math.floor(root.x[j])
kids = [[(j, 1, float(fl))], [(j, -1, -float(fl + 1))]]
for i, ch in enumerate(kids):
    for r in range(reps):
        t0 = time.perf_counter(); g = rootp.child(ch).solve(0.0); dt = time.perf_counter() - t0
    s = g.stats
    print("single engine child %d: status %d pivots %d + %d bland %d total %.3f ms loop %.3f final %.3f" % (i, g.status, s["pivots_phase1"], s["pivots_phase2"], s["bland_steps"], 1e3 * dt, 1e3 * s["seconds_pivot_loop"], 1e3 * s["seconds_final_solve"]), flush=True)
cx.close()
for loop in (1, 0):
    pool = lp.FrontierPool(workers=4, batched=1)
    pool.set("batch_loop", loop)
    pool.set_root(c0, A0, b0)
    for sel, name in ((kids[:1], "child 0"), (kids[1:], "child 1"), (kids, "both")):
        for r in range(reps):
            t0 = time.perf_counter(); res = pool.solve(sel); dt = time.perf_counter() - t0
        st = res.stats
        print("batch_loop %d %s: total %.3f ms batch %.3f ms supersteps %d blocks %d launches %d pivots %d + %d bland %d" % (loop, name, 1e3 * dt, 1e3 * st["seconds_batch"], st["supersteps"], st["blocks"], st["kernel_launches"], st["pivots_phase1"], st["pivots_phase2"], st["bland_steps"]), flush=True)
    pool.close()
