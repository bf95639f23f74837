"""Developer tool: oracle vs GPU traces for the small cases of the exactness tests.  usage: dbg_exact.py tree SEED | extra"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from gomilp_amd import lp, synth
from oracle import oracle as O
import test_gpu_golden as T

def show(c0, A0, b0, cons, knobs):
    cc, AA, bb = O.child_standard_form(c0, A0, b0, cons)
    o = O.simplex(cc, AA, bb, 0.0, None, trace=True)
    cx = lp.Context(**knobs)
    root = cx.upload(c0, A0, b0)
    g = (root.child(cons) if cons else root).solve(0.0, trace=True)
    print("cons", cons)
    print(" oracle status", o.status, "z %.17g" % o.z, "p1/p2", o.pivots_phase1, o.pivots_phase2, "bland", o.bland_steps)
    print(" gpu    status", g.status, "z %.17g" % g.z, "p1/p2", g.stats["pivots_phase1"], g.stats["pivots_phase2"], "bland", g.stats["bland_steps"], "exact", g.stats["cond_fallbacks"])
    for t in range(max(len(o.pivots), len(g.pivots))):
        po = tuple(o.pivots[t]) if t < len(o.pivots) else None
        pg = tuple(g.pivots[t]) if t < len(g.pivots) else None
        print("  ", t, "oracle", po, "gpu", pg, "" if po and pg and po[0] == pg[0] and po[2:] == pg[2:] else "  <--")
    if o.x is not None and g.x is not None:
        print(" x equal", np.array_equal(o.x, g.x)); print(" oracle x", o.x); print(" gpu    x", g.x)
    cx.close()

knobs = dict((k, int(v)) for k, v in (a.split("=") for a in sys.argv[3:] if "=" in a))
if sys.argv[1] == "tree":
    c, G, h, integ = T._degenerate_integer_milp(int(sys.argv[2]))
    c0, A0, b0 = O.convert_to_equalities(c, None, None, G, h)
    print("G", G, "h", h, "c", c)
    show(c0, A0, b0, [], knobs)
else:
    rng = np.random.default_rng(31)
    me, ng = 6, 14
    Ae, Ge = rng.integers(-2, 3, (me, ng)).astype(float), rng.integers(0, 3, (me, ng)).astype(float)
    x0 = rng.integers(0, 3, ng).astype(float)
    A0 = np.zeros((2 * me, ng + me)); A0[:me, :ng] = Ae; A0[me:, :ng] = Ge; A0[me:, ng:] = np.eye(me)
    b0 = np.concatenate([Ae @ x0, Ge @ x0 + rng.integers(1, 4, me)])
    c0 = np.concatenate([rng.integers(0, 4, ng).astype(float), np.zeros(me)])
    for cons in [[], [(2, 1, 1.0)]]:
        show(c0, A0, b0, cons, knobs)
