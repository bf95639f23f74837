"""Developer tool: time the C5 wave (256 children of the 512x1024 root) — device-batched vs one stream per relaxation.
usage: wave_n.py [workers] [reps] [nvars]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nvars = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
children = synth.frontier_children(root.x, mask, nvars)
ref = None
for batched in (1, 0):
    pool = lp.FrontierPool(workers=workers, batched=batched)
    pool.set_root(c, A, b)
    for r in range(reps):
        t0 = time.perf_counter(); res = pool.solve(children); dt = time.perf_counter() - t0
        st = res.stats
        print("batched", batched, "wave %.2f ms  %.0f relax/s" % (1e3 * dt, len(children) / dt), "batch %.2f ms supersteps %d" % (1e3 * st["seconds_batch"], st["supersteps"]),
              "batched_relax", st["batched_relaxations"], "fallbacks", st["host_fallbacks"], "pivots", st["pivots_phase1"], st["pivots_phase2"], "bland", st["bland_steps"],
              "ok", int((res.status == 0).sum()), "infeasible", int((res.status == 2).sum()), flush=True)
    if ref is None:
        ref = res
    else:
        print("same status", np.array_equal(ref.status, res.status), "same z", np.array_equal(ref.z, res.z, equal_nan=True), "same x", np.array_equal(ref.x, res.x))
    pool.close()
