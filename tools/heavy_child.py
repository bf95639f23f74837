"""Developer tool: the heaviest child of the C5 wave alone through the batched path — where its time goes.
usage: heavy_child.py [reps]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
children = synth.frontier_children(root.x, mask, 8)
for sample in (0, 1):
    pool = lp.FrontierPool(workers=4, batched=1, sample_batch=sample)
    pool.set_root(c, A, b)
    for r in range(reps):
        t0 = time.perf_counter(); res = pool.solve(children[:1]); dt = time.perf_counter() - t0
        st = res.stats
        print("sample", sample, "total %.3f ms batch %.3f ms supersteps %d blocks %d sampled %d inner %.3f ms update %.3f ms" % (
            1e3 * dt, 1e3 * st["seconds_batch"], st["supersteps"], st["blocks"], st["blocks_sampled"], 1e3 * st["seconds_inner_kernels"], 1e3 * st["seconds_update_kernels"]),
            "pivots", st["pivots_phase1"], st["pivots_phase2"], "bland", st["bland_steps"], flush=True)
    pool.close()
