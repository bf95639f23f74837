"""Developer tool: the multi-workgroup block kernel (knob bt_groups) against the live oracle at small sizes — root LPs and
children that need Phase I / the Bland rule — and against the M fixture.  usage: groups_check.py [small|M|all]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
from oracle import oracle as O

def t5(piv):
    a = np.array(piv, dtype=np.int64).reshape(-1, 6)
    return a[:, [0, 2, 3, 4, 5]]

what = sys.argv[1] if len(sys.argv) > 1 else "all"
extra = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[2:] if "=" in a)   # e.g. block_k=32
knobs = [int(v) for v in sys.argv[2:] if "=" not in v]
bad = 0
if what in ("small", "all"):
    for m, seed in ((64, 5), (128, 7), (256, 7), (512, 3)):
        c, A, b = synth.dense_lp_standard_form(m, seed)
        o = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True)
        mask = synth.integrality_mask(m, m)
        for G in (knobs or (2, 4, 8)):
            cx = lp.Context(bt_groups=G)
            root = cx.upload(c, A, b)
            g = root.solve(0.0, trace=True)
            ok = g.status == o.status and np.array_equal(t5(g.pivots), t5(o.pivots)) and np.array_equal(g.x, o.x) and g.z == o.z
            print("m=%d G=%d root: status %d pivots %d %s" % (m, G, g.status, len(g.pivots), "OK" if ok else "MISMATCH"), flush=True)
            bad += 0 if ok else 1
            if m <= 256:
                for ci, ch in enumerate(synth.frontier_children(g.x if g.status == 0 else o.x, mask, 3)):
                    cc, AA, bb = O.child_standard_form(c, A, b, ch)
                    oc = O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True, trace=True)
                    p = root.child(ch); gc = p.solve(0.0, trace=True); p.free()
                    ok = gc.status == oc.status and gc.stats["pivots_phase1"] == oc.pivots_phase1 and gc.stats["pivots_phase2"] == oc.pivots_phase2
                    if ok and oc.status == lp.OK:
                        ok = np.array_equal(t5(gc.pivots), t5(oc.pivots)) and np.array_equal(gc.x, oc.x) and gc.z == oc.z
                    print("   child %d: status %d/%d p1 %d p2 %d bland %d/%d %s" % (ci, gc.status, oc.status, oc.pivots_phase1, oc.pivots_phase2, gc.stats["bland_steps"], oc.bland_steps,
                                                                                  "OK" if ok else "MISMATCH"), flush=True)
                    bad += 0 if ok else 1
            cx.close()
if what in ("M", "all"):
    fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lp_M.npz"))
    m, seed = synth.CONFIGS["M"]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    for G in (knobs or (4, 8, 2)):
        cx = lp.Context(bt_groups=G, sample_events=64, chunk=64, **extra)
        p = cx.upload(c, A, b)
        r = p.solve(0.0, trace=True)
        got, want = t5(r.pivots), fx["trace"][:, [0, 2, 3, 4, 5]].astype(np.int64)
        ok = r.status == 0 and got.shape == want.shape and np.array_equal(got, want) and np.array_equal(r.x, fx["x"]) and r.z == float(fx["z"])
        print("M G=%d: status %d pivots %d %s" % (G, r.status, len(got), "OK" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
        for i in range(3):
            t0 = time.perf_counter(); r = p.solve(0.0); dt = time.perf_counter() - t0
            ks = r.stats["pivot_kernel_seconds"]
            print("   total %.2f ms loop %.2f ms inner %.2f us/launch update %.2f us/launch (%d pivots)" % (1e3 * dt, 1e3 * r.stats["seconds_pivot_loop"],
                  1e6 * ks[0] / max(ks[1], 1), 1e6 * ks[2] / max(ks[1], 1), r.stats["pivots_phase2"]), flush=True)
        cx.close()
if what in ("C4",):
    fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lp_C4_prefix.npz"))
    m, seed = synth.CONFIGS["C4"]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    for G in (knobs or (8, 0)):
        cx = lp.Context(bt_groups=G, sample_events=64, chunk=64, **extra)
        p = cx.upload(c, A, b)
        r = p.solve(0.0, trace=True)
        want = fx["trace"][:, [0, 2, 3, 4, 5]].astype(np.int64)
        got = t5(r.pivots)[: len(want)]
        ok = r.status == 0 and np.array_equal(got, want)
        print("C4 G=%d: status %d pivots %d prefix %s z %.17g" % (G, r.status, len(r.pivots), "OK" if ok else "MISMATCH", r.z), flush=True)
        bad += 0 if ok else 1
        for i in range(2):
            t0 = time.perf_counter(); r = p.solve(0.0); dt = time.perf_counter() - t0
            ks = r.stats["pivot_kernel_seconds"]
            print("   total %.2f ms loop %.2f ms inner %.2f us/launch update %.2f us/launch (%d pivots)" % (1e3 * dt, 1e3 * r.stats["seconds_pivot_loop"],
                  1e6 * ks[0] / max(ks[1], 1), 1e6 * ks[2] / max(ks[1], 1), r.stats["pivots_phase2"]), flush=True)
        cx.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
