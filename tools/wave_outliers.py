"""Developer tool: distribution of the C5 wave time over many waves (outliers), with and without Python's GC.  usage: wave_outliers.py [waves]"""
import sys, time, os, gc; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
children = synth.frontier_children(root.x, mask, 8)
pool = lp.FrontierPool(workers=4)
pool.set_root(c, A, b)
for r in range(3): pool.solve(children)
for mode in ("gc on", "gc off"):
    if mode == "gc off": gc.disable()
    ts = []
    for r in range(n):
        t0 = time.perf_counter(); res = pool.solve(children); ts.append(1e3 * (time.perf_counter() - t0))
        if ts[-1] > 8:   # what the C side says about a slow wave
            st = res.stats
            print("  slow wave %d: %.2f ms | C side total %.2f batch %.2f busy %.2f supersteps %d blocks %d launches %d fallbacks %d" % (r, ts[-1], 1e3 * st["seconds_total"], 1e3 * st["seconds_batch"],
                  1e3 * st["seconds_busy_sum"], st["supersteps"], st["blocks"], st["kernel_launches"], st["host_fallbacks"]), flush=True)
    ts = np.array(ts)
    print(mode, "median %.2f ms mean %.2f max %.2f  >8ms: %d of %d" % (np.median(ts), ts.mean(), ts.max(), int((ts > 8).sum()), n), " ".join("%.1f" % t for t in ts), flush=True)
pool.close()
