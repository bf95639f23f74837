"""Developer tool: degenerate children beyond 256 rows — stacked copies of one branch row (what the reference's maxFun quirk produces,
branching.go:54-72) on a 300- / 512-row root — against the live oracle, default knobs: status, z / x bits, pivots, Bland steps.
usage: degen_big.py [m0 ...]"""
import sys, os, math, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
from oracle import oracle as O
O.set_threads(8)


def family(root_x, mask):
    picks = [j for j in range(len(mask) - 1, -1, -1) if mask[j] and root_x[j] != math.floor(root_x[j])][:2]
    v1, v2 = picks
    f1, f2 = float(math.floor(root_x[v1])), float(math.floor(root_x[v2]))
    le1, ge1, le2, ge2 = (v1, 1, f1), (v1, -1, -(f1 + 1.0)), (v2, 1, f2), (v2, -1, -(f2 + 1.0))
    return [[le1] * 3, [ge1] * 3, [le1, le2, le1, le2], [le1, ge2, le1, ge2], [ge1, le2] * 2, [le1] * 4, [le2, le2, ge1]]


if __name__ == "__main__":
    for m0 in [int(a) for a in sys.argv[1:]] or [300, 512]:
        c, A, b = synth.dense_lp_standard_form(m0, 3)
        mask = synth.integrality_mask(m0, m0)
        cx = lp.Context()
        root = cx.upload(c, A, b)
        r0 = root.solve(0.0)
        kids = family(r0.x, mask)
        pool = lp.FrontierPool(workers=2)
        pool.set_root(c, A, b)
        res = pool.solve(kids)
        for i, ch in enumerate(kids):
            t0 = time.time()
            o = O.simplex(*O.child_standard_form(c, A, b, ch), 0.0, None, fast_initial_basis=True)
            g = root.child(ch).solve(0.0)
            same = g.status == o.status and (o.status != 0 or (g.z == o.z and np.array_equal(g.x, o.x)))
            samep = res.status[i] == o.status and (o.status != 0 or (res.z[i] == o.z and np.array_equal(res.x[i][: A.shape[1]], o.x[: A.shape[1]])))
            print("m0 %d child %d K %d: oracle status %d pivots %d+%d bland %d | engine status %d pivots %d+%d bland %d exact steps %d | bit-exact: single %s pool %s (%.1f s)" % (
                m0, i, len(ch), o.status, o.pivots_phase1, o.pivots_phase2, o.bland_steps, g.status, g.stats["pivots_phase1"], g.stats["pivots_phase2"],
                g.stats["bland_steps"], g.stats["cond_fallbacks"], same, samep, time.time() - t0), flush=True)
        pool.close(); cx.close()
