"""Developer tool: repeat the metric solve and a concurrent 1100-row solve many times in one process; every result must
be bit-identical to the first (the multi-workgroup block kernel synchronises through memory: a soak for timing-dependent faults)."""
import sys, os, time, threading; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
m, seed = synth.CONFIGS["M"]
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(); p = cx.upload(c, A, b)
first = p.solve(0.0, trace=True)
bad = 0
t0 = time.perf_counter()
for i in range(reps):
    r = p.solve(0.0, trace=True)
    if r.status != first.status or not np.array_equal(r.x, first.x) or r.pivots != first.pivots: bad += 1
print("M x %d: %.1f ms each, mismatches %d" % (reps, 1e3 * (time.perf_counter() - t0) / reps, bad), flush=True)
cx.close()
cxs, ps = [], []
for i in range(6):
    q = synth.dense_lp_standard_form(1100, 60 + i); cxx = lp.Context(); cxs.append(cxx); ps.append(cxx.upload(*q))
ref = [q.solve(0.0) for q in ps]
bad2 = 0
for rep in range(reps // 3):
    got = [None] * 6
    def run(i): got[i] = ps[i].solve(0.0)
    ths = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    [t.start() for t in ths]; [t.join() for t in ths]
    bad2 += sum(1 for g, o in zip(got, ref) if g.status != o.status or not np.array_equal(g.x, o.x))
print("6 concurrent 1100-row solves x %d: mismatches %d" % (reps // 3, bad2), flush=True)
for cxx in cxs: cxx.close()
sys.exit(1 if bad or bad2 else 0)
