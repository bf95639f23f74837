"""Developer tool: the batched persistent loop kernel (k_b_loop) against the launch-pair schedule and the single-relaxation engine on random
narrow waves: children of 200- / 384- / 512-row roots with 1..6 random branch rows (feasible and Phase-I starts mixed, some infeasible,
some with the zero-level artificial exchange), wave after wave on one pool (launch parity, counters and buffer choice carried along).
Every status / z / x must be bit-identical across the three paths.  usage: soak_loop.py [waves]"""
import sys, os, math, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(11)
bad = tot = loops = 0
for m0, seed in ((200, 21), (384, 22), (512, 3)):
    c, A, b = synth.dense_lp_standard_form(m0, seed)
    cx = lp.Context(); root = cx.upload(c, A, b); r0 = root.solve(0.0)
    frac = [j for j in range(m0) if r0.x[j] != math.floor(r0.x[j])]
    pl = lp.FrontierPool(workers=4); pl.set_root(c, A, b)
    pp = lp.FrontierPool(workers=4, batch_loop=0); pp.set_root(c, A, b)
    for w in range(waves):
        n = int(rng.integers(1, 25))
        kids = []
        for _ in range(n):
            K = int(rng.integers(1, 7))
            ch = []
            for _k in range(K):
                j = int(rng.choice(frac)) if rng.random() < 0.8 else int(rng.integers(0, m0))
                fl = float(math.floor(r0.x[j]))
                ch.append((j, 1, fl) if rng.random() < 0.6 else (j, -1, -(fl + 1.0)))
            kids.append(ch)
        a = pl.solve(kids); p = pp.solve(kids)
        loops += 1
        for i, ch in enumerate(kids):
            tot += 1
            same = a.status[i] == p.status[i] and (a.status[i] != 0 or (a.z[i] == p.z[i] and np.array_equal(a.x[i], p.x[i])))
            if same and (i % 5 == 0):
                g = root.child(ch).solve(0.0)
                same = g.status == a.status[i] and (g.status != 0 or (g.z == a.z[i] and np.array_equal(g.x[: A.shape[1]], a.x[i][: A.shape[1]])))
            if not same:
                bad += 1
                print("MISMATCH m0 %d wave %d child %d %s: loop status %d z %.17g | pairs status %d z %.17g" % (m0, w, i, ch, a.status[i], a.z[i], p.status[i], p.z[i]), flush=True)
    print("m0 %d: %d waves done, fallbacks loop %d pairs %d" % (m0, waves, a.stats["host_fallbacks"], p.stats["host_fallbacks"]), flush=True)
    pl.close(); pp.close(); cx.close()
print("relaxations %d in %d waves, mismatches %d" % (tot, loops, bad))
sys.exit(1 if bad else 0)
