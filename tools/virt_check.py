"""Developer tool: wide waves on virtual tableaus (BatchLP::virt: set-up pivot and first block on computed entries, only the survivors' tableaus
written) against the materialised path (pool knob batch_virt = 0): bits of every result, and the wave times.
usage: virt_check.py [branch_vars ...]   (8 = 256 children, 11 = 2048, 13 = 8192)"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
bad = 0
for nv in [int(a) for a in sys.argv[1:]] or [8, 11]:
    children = synth.frontier_children(root.x, mask, nv)
    packed = lp.pack_children(children)
    res = {}
    for virt in (1, 0):
        pool = lp.FrontierPool(workers=4, batch_virt=virt); pool.set_root(c, A, b)
        ts = []
        for r in range(5):
            t0 = time.perf_counter(); out = pool.solve(packed); ts.append(time.perf_counter() - t0)
        res[virt] = out
        print("%d children, batch_virt %d: best %.2f ms median %.2f ms (%.1f k relaxations/s) batch %.2f ms supersteps %d launches %d fallbacks %d feasible %d pivots %d+%d" % (
            len(children), virt, 1e3 * min(ts), 1e3 * float(np.median(ts)), len(children) / float(np.median(ts)) / 1e3, 1e3 * out.stats["seconds_batch"], out.stats["supersteps"],
            out.stats["kernel_launches"], out.stats["host_fallbacks"], int((out.status == 0).sum()), out.stats["pivots_phase1"], out.stats["pivots_phase2"]), flush=True)
        pool.close()
    a, p = res[1], res[0]
    same = np.array_equal(a.status, p.status) and np.array_equal(a.has_x, p.has_x)
    ok = a.status == 0
    same = same and np.array_equal(a.z[ok], p.z[ok]) and np.array_equal(a.x[ok], p.x[ok])
    same = same and (a.stats["pivots_phase1"], a.stats["pivots_phase2"], a.stats["bland_steps"]) == (p.stats["pivots_phase1"], p.stats["pivots_phase2"], p.stats["bland_steps"])
    print("  virtual vs materialised: status / z / x bits / pivot totals %s" % ("identical" if same else "DIFFERENT"), flush=True)
    bad += 0 if same else 1
print("mismatches %d" % bad)
sys.exit(1 if bad else 0)
