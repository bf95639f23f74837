"""Developer tool: first pivot at which a solve of a golden config leaves the oracle's trace.  usage: first_diff.py M [knob=value ...]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
name = sys.argv[1]
knobs = dict((k, int(v)) for k, v in (a.split("=") for a in sys.argv[2:]))
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", {"M": "lp_M.npz", "C2": "lp_C2.npz", "C4": "lp_C4_prefix.npz"}[name]))
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(**knobs)
p = cx.upload(c, A, b)
for rep in range(3):
    r = p.solve(0.0, trace=True)
    got = np.array(r.pivots, dtype=np.int64).reshape(-1, 6)[:, [0, 2, 3, 4, 5]]
    want = fx["trace"][:, [0, 2, 3, 4, 5]].astype(np.int64)
    n = min(len(got), len(want))
    d = (got[:n] != want[:n]).any(axis=1)
    print(name, knobs, "status", r.status, "pivots", len(got), "oracle", len(want), "first diff", int(np.argmax(d)) if d.any() else -1, flush=True)
cx.close()
