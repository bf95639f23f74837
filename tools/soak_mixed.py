"""Developer tool: loop-kernel slots under mixed shapes — C4 (needs the device alone) next to metric-size and 1100-row solves (one slot
each) from concurrent host threads; every result must equal the solve done alone."""
import sys, os, time, threading; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [("C4", None), ("M", None), ("M", 102), (None, 1100), (None, 1300), ("C2", None)]
cxs, ps, ref = [], [], []
for name, extra in shapes:
    if name: m, seed = synth.CONFIGS[name]; seed = extra or seed
    else: m, seed = extra, 77
    q = synth.dense_lp_standard_form(m, seed)
    cx = lp.Context(); cxs.append(cx); p = cx.upload(*q); ps.append(p); ref.append(p.solve(0.0))
bad = 0
t0 = time.perf_counter()
for rep in range(reps):
    got = [None] * len(ps)
    def run(i): got[i] = ps[i].solve(0.0)
    ths = [threading.Thread(target=run, args=(i,)) for i in range(len(ps))]
    [t.start() for t in ths]; [t.join() for t in ths]
    for i, (g, o) in enumerate(zip(got, ref)):
        wrong = g.status != o.status or not np.array_equal(g.x, o.x)
        if wrong or g.stats["device_retries"]:
            bad += 1
            print("round %d shape %s: %s (status %d vs %d, retries %d, pivots %d)" % (rep, shapes[i], "MISMATCH" if wrong else "retry, same bits", g.status, o.status, g.stats["device_retries"],
                                                                                  g.stats["pivots_phase1"] + g.stats["pivots_phase2"]), flush=True)
print("mixed shapes x %d rounds: %.1f ms per round, mismatches / retries %d" % (reps, 1e3 * (time.perf_counter() - t0) / reps, bad), flush=True)
for cx in cxs: cx.close()
sys.exit(1 if bad else 0)
