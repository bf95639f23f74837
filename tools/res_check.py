"""Developer tool: the register-resident tableau kernel (res_kernels.hip k_b_res, pool knob batch_res) against the oracle fixture of the C5
wave, against the launch-pair schedule and against the single-relaxation engine; and what it costs.
usage: res_check.py [heavy] [wave] [soak N]"""
import sys, os, math, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth

HERE = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:] or ["heavy", "wave", "soak", "6"]
bad = 0


def c5():
    m, seed = synth.CONFIGS["C5"]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    mask = synth.integrality_mask(m, m)
    cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
    return c, A, b, synth.frontier_children(root.x, mask, 8)


if "heavy" in args or "wave" in args:
    fx = np.load(os.path.join(os.path.dirname(HERE), "tests", "golden", "frontier_C5.npz"))
    c, A, b, children = c5()
    feas = [i for i, ch in enumerate(children) if all(r >= -1e-13 for (_, _, r) in ch)]
    for res_on in (1, 0):
        pool = lp.FrontierPool(workers=8, batch_res=res_on); pool.set_root(c, A, b)
        if "heavy" in args:
            best = 1e9
            for r in range(6):
                t0 = time.perf_counter(); res = pool.solve([children[i] for i in feas]); best = min(best, time.perf_counter() - t0)
            st = res.stats
            ok = all(res.status[k] == fx["status"][i] and (res.status[k] != 0 or (res.z[k] == fx["z"][i] and np.array_equal(res.x[k], fx["x"][i]))) for k, i in enumerate(feas))
            bad += 0 if ok else 1
            print("batch_res %d: feasible-start group %s alone: best %.3f ms (batch %.3f ms) supersteps %d blocks %d launches %d pivots %d+%d bland %d fallbacks %d | fixture bits: %s" % (
                res_on, feas, 1e3 * best, 1e3 * st["seconds_batch"], st["supersteps"], st["blocks"], st["kernel_launches"], st["pivots_phase1"], st["pivots_phase2"], st["bland_steps"],
                st["host_fallbacks"], "identical" if ok else "DIFFERENT"), flush=True)
        if "wave" in args:
            ts = []
            for r in range(12):
                t0 = time.perf_counter(); res = pool.solve(children); ts.append(time.perf_counter() - t0)
            ok = bool(np.array_equal(res.status, fx["status"]))
            okm = fx["status"] == 0
            ok = ok and np.array_equal(res.z[okm], fx["z"][okm]) and np.array_equal(res.x[okm], fx["x"][okm])
            piv = fx["pivots"]
            ok = ok and res.stats["pivots_phase1"] == int(piv[:, 0].sum()) and res.stats["pivots_phase2"] == int(piv[:, 1].sum())
            bad += 0 if ok else 1
            print("batch_res %d: C5 wave of 256: best %.3f ms median %.3f ms (%.1f k relaxations/s at the median) fallbacks %d bland %d | fixture bits + pivot totals: %s" % (
                res_on, 1e3 * min(ts), 1e3 * float(np.median(ts)), 256e-3 / float(np.median(ts)), res.stats["host_fallbacks"], res.stats["bland_steps"], "identical" if ok else "DIFFERENT"), flush=True)
        pool.close()

if "soak" in args:
    waves = int(args[args.index("soak") + 1]) if len(args) > args.index("soak") + 1 else 6
    rng = np.random.default_rng(17)
    tot = nw = 0
    for m0, seed in ((200, 21), (384, 22), (512, 3), (300, 23)):
        c, A, b = synth.dense_lp_standard_form(m0, seed)
        cx = lp.Context(); root = cx.upload(c, A, b); r0 = root.solve(0.0)
        frac = [j for j in range(m0) if r0.x[j] != math.floor(r0.x[j])]
        pr = lp.FrontierPool(workers=4, batch_res=1); pr.set_root(c, A, b)
        pp = lp.FrontierPool(workers=4, batch_res=0, batch_loop=0); pp.set_root(c, A, b)
        fb = [0, 0]
        for w in range(waves):
            n = int(rng.integers(1, 13))
            kids = []
            for _ in range(n):
                K = int(rng.integers(1, 7))
                ch = []
                for _k in range(K):
                    j = int(rng.choice(frac)) if rng.random() < 0.8 else int(rng.integers(0, m0))
                    fl = float(math.floor(r0.x[j]))
                    ch.append((j, 1, fl) if rng.random() < 0.6 else (j, -1, -(fl + 1.0)))
                kids.append(ch)
            a = pr.solve(kids); p = pp.solve(kids)
            nw += 1; fb[0] += a.stats["host_fallbacks"]; fb[1] += p.stats["host_fallbacks"]
            for i, ch in enumerate(kids):
                tot += 1
                same = a.status[i] == p.status[i] and (a.status[i] != 0 or (a.z[i] == p.z[i] and np.array_equal(a.x[i], p.x[i])))
                if same and (i % 4 == 0):
                    g = root.child(ch).solve(0.0)
                    same = g.status == a.status[i] and (g.status != 0 or (g.z == a.z[i] and np.array_equal(g.x[: A.shape[1]], a.x[i][: A.shape[1]])))
                if not same:
                    bad += 1
                    print("MISMATCH m0 %d wave %d child %d %s: resident status %d z %.17g | pairs status %d z %.17g" % (m0, w, i, ch, a.status[i], a.z[i], p.status[i], p.z[i]), flush=True)
        print("m0 %d: %d waves, fallbacks resident %d pairs %d" % (m0, waves, fb[0], fb[1]), flush=True)
        pr.close(); pp.close(); cx.close()
    print("soak: relaxations %d in %d waves" % (tot, nw))
print("mismatches %d" % bad)
sys.exit(1 if bad else 0)
