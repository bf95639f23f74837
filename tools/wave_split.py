"""Developer tool: the C5 wave as a whole, its Phase-I group alone and its feasible-start group alone (knob split_phase)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
children = synth.frontier_children(root.x, mask, 8)
feas = [i for i, ch in enumerate(children) if all(r >= -1e-13 for (_, _, r) in ch)]
infe = [i for i in range(len(children)) if i not in feas]
print("feasible-start children:", feas)
pool = lp.FrontierPool(workers=4, batched=1)
if len(sys.argv) > 1: pool.set("batch_loop", int(sys.argv[1]))
pool.set_root(c, A, b)
def t(label, chs, n=6):
    best = 1e9
    for r in range(n):
        t0 = time.perf_counter(); res = pool.solve(chs); dt = time.perf_counter() - t0
        best = min(best, dt)
    print("%-28s best %.2f ms (batch %.2f ms, supersteps %d, blocks %d)" % (label, 1e3 * best, 1e3 * res.stats["seconds_batch"], res.stats["supersteps"], res.stats["blocks"]), flush=True)
for sp in (1, 0):
    pool.set("split_phase", sp)
    t("all, split_phase=%d" % sp, children)
t("phase-I group alone", [children[i] for i in infe])
t("feasible group alone", [children[i] for i in feas])
pool.close()
# single path: block kernel thread count at 512 rows
for nt in ():
    cx = lp.Context(bt_nt=nt); p = cx.upload(c, A, b)
    for _ in range(3): r = p.solve(0.0)
    print("C3 root bt_nt", nt, "loop %.3f ms pivots %d" % (1e3 * r.stats["seconds_pivot_loop"], r.stats["pivots_phase2"]), flush=True)
    cx.close()
