"""Developer tool: the C5 wave cold (reference-faithful) vs warm-started (dual simplex from the root's optimal basis)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
nvars = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
pool = lp.FrontierPool(workers=4)
pool.set_root(c, A, b)
root = pool.solve_root(0.0)
print("root", root.status, root.z, root.stats["pivots_phase2"])
children = synth.frontier_children(root.x, mask, nvars)
cold = None
for mode in (0, 1):
    pool.set("warm_start", mode)
    for r in range(5):
        t0 = time.perf_counter(); res = pool.solve(children); dt = time.perf_counter() - t0
    st = res.stats
    print("warm" if mode else "cold", "wave %.2f ms %.0f relax/s" % (1e3 * dt, len(children) / dt), "pivots", st["pivots_phase1"], st["pivots_phase2"], "supersteps", st["supersteps"],
          "fallbacks", st["host_fallbacks"], "ok", int((res.status == 0).sum()), "infeasible", int((res.status == 2).sum()), flush=True)
    if mode == 0:
        cold = res
    else:
        same_status = np.array_equal(cold.status, res.status)
        ok = cold.status == 0
        dz = np.abs(cold.z[ok] - res.z[ok]).max() if ok.any() else 0.0
        dx = np.abs(cold.x[ok] - res.x[ok]).max() if ok.any() else 0.0
        print("same status", same_status, "max |dz| %.3g max |dx| %.3g" % (dz, dx), "bit-identical x:", int(sum(np.array_equal(cold.x[i], res.x[i]) for i in np.nonzero(ok)[0])), "of", int(ok.sum()))
        if not same_status:
            bad = np.nonzero(cold.status != res.status)[0]
            print("status differs at", bad[:10], cold.status[bad[:10]], res.status[bad[:10]])
pool.close()
