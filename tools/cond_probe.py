"""Developer tool: badly scaled small LPs (entries 1e-13 .. 1e3) — where the oracle leaves its loop with mat.Condition, what does the GPU engine return?"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp
from oracle import oracle as O

def scaled_lp(seed, m=None, nv=None):
    """m = None: the 2..4-row family of round 2; m given: the same entry distribution at that size (nv defaults to m // 2)."""
    rng = np.random.default_rng(seed)
    if m is None:
        m = int(rng.integers(2, 5)); nv = int(rng.integers(2, 5))
    elif nv is None:
        nv = m // 2
    G = rng.standard_normal((m, nv)) * 10.0 ** rng.integers(-13, 4, (m, nv))
    h = np.abs(rng.standard_normal(m)) * 10.0 ** rng.integers(-3, 3, m)
    c = -np.abs(rng.standard_normal(nv))
    return np.concatenate([c, np.zeros(m)]), np.hstack([G, np.eye(m)]), h

def scaled_cols_lp(seed, m):
    """A bounded dense LP of m rows whose structural COLUMNS are rescaled by 10^(span * j / nv), span = 19 * u(seed): about a third of
    the family ends on a basis with kappa_1 > 1e16 (a column of norm ~1e17 next to the slack columns), the rest stays below."""
    from gomilp_amd import synth
    c, G, h = synth.dense_lp_inequality_form(m, 1000 + seed)
    nv = G.shape[1]
    span = 19.0 * np.random.default_rng(seed).random()
    sc = 10.0 ** (span * np.arange(nv) / nv)
    return np.concatenate([c * sc, np.zeros(m)]), np.hstack([G * sc, np.eye(m)]), h


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else None
    tally = {}
    zbad = 0
    for seed in range(N):
        c, A, b = scaled_cols_lp(seed, M) if (M and len(sys.argv) > 3) else scaled_lp(seed, M)
        o = O.simplex(c, A, b, 0.0, None)
        if M: print("seed", seed, "...", flush=True)
        g = lp.simplex(c, A, b, 0.0, None)
        key = (o.status, g.status)
        tally[key] = tally.get(key, 0) + 1
        if M and (o.status != g.status or o.status != 0):
            print("seed", seed, "oracle", o.status, "pivots", o.pivots_phase1, o.pivots_phase2, "gpu", g.status, g.stats["pivots_phase1"], g.stats["pivots_phase2"],
                  "kappa1 %.3g" % g.stats["cond1_final"], "fallbacks", g.stats["cond_fallbacks"], "rebuilds", g.stats["refreshes"], flush=True)
        if o.status == g.status and o.x is not None and g.x is not None:
            if not abs(g.z - o.z) <= 1e-6 * max(1.0, abs(o.z)):
                zbad += 1
                print("seed", seed, "z off: oracle %.17g gpu %.17g pivots %d %d" % (o.z, g.z, o.pivots_phase2, g.stats["pivots_phase2"]), flush=True)
    print("(oracle status, gpu status) -> count:", dict(sorted(tally.items())), "same status but z off:", zbad)
