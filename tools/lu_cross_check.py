"""Developer tool: the cross-workgroup LU panel (lu_cross.hip, knob lu_cross) against the one-workgroup panel — same bits — and what the final
solve costs with each.  usage: lu_cross_check.py [config ...]   (C3 C2 M, ties150, ties300)"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
from tools.lu_ties import integer_lp
names = sys.argv[1:] or ["ties150", "C3", "ties300", "C2", "M"]
bad = 0
for name in names:
    if name.startswith("ties"):
        c, A, b = integer_lp(int(name[4:]), 0)
    else:
        m, seed = synth.CONFIGS[name]
        c, A, b = synth.dense_lp_standard_form(m, seed)
    res = {}
    for cross in (0, 1):
        cx = lp.Context(lu_cross=cross, lu_blocked=2, max_pivots=20000)
        p = cx.upload(c, A, b)
        best = None
        for i in range(3):
            r = p.solve(0.0)
            if best is None or r.stats["seconds_final_solve"] < best.stats["seconds_final_solve"]: best = r
        cx.close()
        res[cross] = best
        s = best.stats
        print("%-8s lu_cross %d: status %d pivots %d final %.3f ms (device %.3f) dense steps %d rounds %d retries %d" % (
            name, cross, best.status, s["pivots_phase1"] + s["pivots_phase2"], 1e3 * s["seconds_final_solve"], 1e3 * s.get("seconds_final_device", 0), s["lu_dense_steps"], s["lu_rounds"], s.get("device_retries", 0)), flush=True)
    a, bq = res[0], res[1]
    same = a.status == bq.status and np.array_equal(a.basis, bq.basis) and (a.x is None) == (bq.x is None) and (a.x is None or np.array_equal(a.x, bq.x)) and (a.z == bq.z or (a.z != a.z and bq.z != bq.z))
    print("%-8s same bits: %s" % (name, same), flush=True)
    bad += 0 if same else 1
print("TOTAL mismatches", bad)
