"""Developer tool: badly scaled small LPs (entries 1e-13 .. 1e3) — where the oracle leaves its loop with mat.Condition, what does the GPU engine return?"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp
from oracle import oracle as O

def scaled_lp(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.integers(2, 5)); nv = int(rng.integers(2, 5))
    G = rng.standard_normal((m, nv)) * 10.0 ** rng.integers(-13, 4, (m, nv))
    h = np.abs(rng.standard_normal(m)) * 10.0 ** rng.integers(-3, 3, m)
    c = -np.abs(rng.standard_normal(nv))
    return np.concatenate([c, np.zeros(m)]), np.hstack([G, np.eye(m)]), h

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    tally = {}
    zbad = 0
    for seed in range(N):
        c, A, b = scaled_lp(seed)
        o = O.simplex(c, A, b, 0.0, None)
        g = lp.simplex(c, A, b, 0.0, None)
        key = (o.status, g.status)
        tally[key] = tally.get(key, 0) + 1
        if o.status == g.status and o.x is not None and g.x is not None:
            if not abs(g.z - o.z) <= 1e-6 * max(1.0, abs(o.z)):
                zbad += 1
    print("(oracle status, gpu status) -> count:", dict(sorted(tally.items())), "same status but z off:", zbad)
