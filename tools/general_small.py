"""Developer tool: where the device column search starts to pay — equality forms of 2 mg rows, whole solve with the search on the host against
the device (knob general_min_rows).  usage: general_small.py [mg ...]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp
for mg in ([int(a) for a in sys.argv[1:]] or [16, 32, 48, 64, 90, 112, 150]):
    ng = mg + max(8, mg // 2)
    rng = np.random.default_rng(5)
    x0 = np.abs(rng.standard_normal(ng))
    A0 = np.zeros((2 * mg, ng + mg)); A0[:mg, :ng] = rng.standard_normal((mg, ng)); A0[mg:, :ng] = rng.standard_normal((mg, ng)); A0[mg:, ng:] = np.eye(mg)
    b0 = np.concatenate([A0[:mg, :ng] @ x0, A0[mg:, :ng] @ x0 + np.abs(rng.standard_normal(mg))])
    c0 = np.concatenate([np.abs(rng.standard_normal(ng)), np.zeros(mg)])
    out = []
    ref = None
    for minrows in (100000, 2):
        cx = lp.Context(general_min_rows=minrows)
        p = cx.upload(c0, A0, b0)
        p.solve(0.0)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r = p.solve(0.0); best = min(best, time.perf_counter() - t0)
        cx.close()
        if ref is None: ref = r
        same = r.status == ref.status and r.z == ref.z and (r.x is None) == (ref.x is None) and (r.x is None or np.array_equal(r.x, ref.x))
        out.append("%s %.2f ms%s" % ("host search" if minrows > 2 else "device search", 1e3 * best, "" if same else " DIFFERENT RESULT"))
    print("rows %4d cols %4d status %d pivots %d + %d:" % (2 * mg, ng + mg, ref.status, ref.stats["pivots_phase1"], ref.stats["pivots_phase2"]), " | ".join(out), flush=True)
