"""Developer tool (build container, hours): continue the oracle's C4 pivot sequence behind the 1000-pivot prefix fixture, from the basis
the prefix ends with (lp.Simplex's initialBasic, simplex.go:147-161; the nonbasic list is rebuilt in ascending order there, so only the
(entering, leaving) VARIABLES of a pivot are comparable with the uninterrupted run, not its positions — on this LP no two reduced costs
or ratios tie, so the variables are decided by values alone).  Appends chunk by chunk to tests/golden/lp_C4_cont.npz.
usage: c4_continue.py [total_pivots] [threads]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import synth
from oracle import oracle as O
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
O.set_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fn = os.path.join(root, "tests", "golden", "lp_C4_cont.npz")
m, seed = synth.CONFIGS["C4"]
c, A, b = synth.dense_lp_standard_form(m, seed)
pre = np.load(os.path.join(root, "tests", "golden", "lp_C4_prefix.npz"))
if os.path.exists(fn):
    d = np.load(fn); pairs = [tuple(p) for p in d["pairs"]]; basis = d["basis"]
else:
    pairs, basis = [], pre["basis"].astype(np.int64)
start = int(pre["trace"].shape[0])
chunk = 50
while len(pairs) < total:
    t0 = time.time()
    o = O.simplex(c, A, b, 0.0, basis, trace=True, stop_after_pivots=chunk)
    assert o.status == 0 and o.basis is not None
    pairs += [(int(p[4]), int(p[5])) for p in o.pivots]
    basis = np.asarray(o.basis, dtype=np.int64)
    np.savez_compressed(fn + ".tmp.npz", m=m, seed=seed, start=start, pairs=np.array(pairs, dtype=np.int64), basis=basis)
    os.replace(fn + ".tmp.npz", fn)
    print("pivots %d..%d done (%.0f s for %d)" % (start, start + len(pairs), time.time() - t0, len(o.pivots)), flush=True)
    if len(o.pivots) < chunk: break
