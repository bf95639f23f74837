"""Developer tool: the condition guards on column-scaled LPs whose entries span LESS than nine decades (the degenerate-pivot guard of
badly scaled inputs stays off: only the pivot-element floor BTArgs::cguard and the final-basis measurement act), 320 - 512 rows, roots
and Phase-I children, against the live oracle: status, pivot counts, z.  usage: cond_sweep.py [cases]"""
import sys, os, math, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
from oracle import oracle as O


def moderately_scaled_lp(seed, m, decades):
    c, G, h = synth.dense_lp_inequality_form(m, 2000 + seed)
    nv = G.shape[1]
    sc = 10.0 ** (decades * np.arange(nv) / nv)
    return np.concatenate([c * sc, np.zeros(m)]), np.hstack([G * sc, np.eye(m)]), h


def cases(n):
    out = []
    for s in range(n):
        m = (320, 384, 512)[s % 3] if s % 5 else 320
        out.append((s, m, (2.0, 5.0, 8.0, 8.9)[s % 4]))
    return out


if __name__ == "__main__":
    O.set_threads(16)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    agree = tot = exact = 0
    for seed, m, dec in cases(n):
        c, A, b = moderately_scaled_lp(seed, m, dec)
        t0 = time.time()
        o = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True)
        cx = lp.Context(); root = cx.upload(c, A, b); g = root.solve(0.0)
        same = o.status == g.status and (o.status != 0 or (abs(o.z - g.z) <= 1e-9 * max(1.0, abs(o.z)) and o.pivots_phase2 == g.stats["pivots_phase2"]))
        tot += 1; agree += int(same); exact += g.stats["cond_fallbacks"]
        print("root seed %d m %d decades %.1f: oracle status %d pivots %d | engine status %d pivots %d exact steps %d kappa_1 %.3g | %s (%.0f s)" % (
            seed, m, dec, o.status, o.pivots_phase2, g.status, g.stats["pivots_phase2"], g.stats["cond_fallbacks"], g.stats["cond1_final"], "same" if same else "DIFFERENT", time.time() - t0), flush=True)
        if g.status == 0:
            mask = [True] * (A.shape[1] - m) + [False] * m
            kids = synth.frontier_children(g.x, mask, 1)
            for ch in kids:
                t0 = time.time()
                oc = O.simplex(*O.child_standard_form(c, A, b, ch), 0.0, None, fast_initial_basis=True)
                gc = root.child(ch).solve(0.0)
                same = oc.status == gc.status and (oc.status != 0 or (abs(oc.z - gc.z) <= 1e-9 * max(1.0, abs(oc.z)) and (oc.pivots_phase1, oc.pivots_phase2) == (gc.stats["pivots_phase1"], gc.stats["pivots_phase2"])))
                tot += 1; agree += int(same); exact += gc.stats["cond_fallbacks"]
                print("  child %s: oracle status %d pivots %d+%d | engine status %d pivots %d+%d exact steps %d | %s (%.0f s)" % (
                    ch, oc.status, oc.pivots_phase1, oc.pivots_phase2, gc.status, gc.stats["pivots_phase1"], gc.stats["pivots_phase2"], gc.stats["cond_fallbacks"], "same" if same else "DIFFERENT", time.time() - t0), flush=True)
        cx.close()
    print("cases %d, agreement %d, exact steps taken %d" % (tot, agree, exact))
