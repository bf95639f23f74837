"""Developer tool: the strict mode (knob exact_degenerate = 3: EVERY pivot and the stop test decided on fresh gonum-order solves, with the
reference's condition guard) on the three inputs where the default mode leaves the reference's path, against the live
oracle: node 13 of the equality-constrained tree (reference: panic mat.Condition), the 385-row child of tools/cond_sweep.py (2448 vs 2452
pivots), the 512-row root over 8 decades (4355 vs 4396).  usage: strict_check.py [node13] [child385] [root512]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, bnb, synth
from oracle import oracle as O


def report(name, o, g, t_o, t_g):
    same = o.status == g.status and (o.pivots_phase1, o.pivots_phase2) == (g.stats["pivots_phase1"], g.stats["pivots_phase2"])
    bits = o.x is not None and g.x is not None and o.z == g.z and np.array_equal(np.asarray(o.x), np.asarray(g.x))
    print("%s: oracle status %d pivots %d+%d (%.0f s) | strict engine status %d pivots %d+%d exact steps %d (%.0f s) | %s%s" % (
        name, o.status, o.pivots_phase1, o.pivots_phase2, t_o, g.status, g.stats["pivots_phase1"], g.stats["pivots_phase2"],
        g.stats["cond_fallbacks"], t_g, "SAME status + pivot counts" if same else "DIFFERENT", ", z / x bit-identical" if bits else ""), flush=True)
    return same


def run(c, A, b, child, mode, fast):
    t0 = time.time()
    o = O.simplex(*(O.child_standard_form(c, A, b, child) if child else (c, A, b)), 0.0, None, fast_initial_basis=fast)
    t1 = time.time()
    cx = lp.Context(exact_degenerate=mode)
    try:
        root = cx.upload(c, A, b)
        g = (root.child(child) if child else root).solve(0.0)
    finally:
        cx.close()
    return o, g, t1 - t0, time.time() - t1


if __name__ == "__main__":
    O.set_threads(16)
    which = sys.argv[1:] or ["node13", "child385", "root512"]
    ok = True
    if "node13" in which:
        from gen_golden import eq_problem
        fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "milp_EQ.npz"))
        c, A, b, G, h, integ = eq_problem()
        c0, A0, b0 = bnb.convert_to_equalities(c, A, b, G, h)
        k = int(fx["ncons"][13])
        cons = [(int(v), int(s), float(r)) for (v, s, r) in fx["constraints"][13, :k]]
        for mode in (1, 3):
            o, g, to, tg = run(c0, A0, b0, cons, mode, False)
            ok &= report("EQ-tree node 13, exact_degenerate %d" % mode, o, g, to, tg) or mode != 3
    if "seed1079" in which:   # the one badly scaled 2-4-row LP whose status differs at default knobs (tests: test_condition_guards_on_badly_scaled_lps)
        from cond_probe import scaled_lp
        c, A, b = scaled_lp(1079)
        for mode in (1, 3):
            o, g, to, tg = run(c, A, b, None, mode, False)
            report("badly scaled seed 1079 (%d x %d), exact_degenerate %d" % (A.shape[0], A.shape[1], mode), o, g, to, tg)
    if "child385" in which or "root512" in which:
        from cond_sweep import moderately_scaled_lp
    if "child385" in which:
        c, A, b = moderately_scaled_lp(4, 384, 2.0)
        for mode in (1, 3):
            o, g, to, tg = run(c, A, b, [(380, 1, 0.0)], mode, True)
            ok &= report("385-row child, exact_degenerate %d" % mode, o, g, to, tg) or mode != 3
    if "root512" in which:
        c, A, b = moderately_scaled_lp(2, 512, 8.0)
        o, g, to, tg = run(c, A, b, None, 3, True)
        ok &= report("512-row root over 8 decades, exact_degenerate 3", o, g, to, tg)
    print("strict mode follows the reference on every case:", bool(ok))
