"""Developer tool: equality-constrained LP (no slack basis): where the time goes (search on the host vs the rest).  general_n.py m n"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp
from oracle import oracle as O
m = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(5)
x0 = np.abs(rng.standard_normal(n))
A = rng.standard_normal((m, n)); b = A @ x0
G = rng.standard_normal((m, n)); h = G @ x0 + np.abs(rng.standard_normal(m))
c = np.abs(rng.standard_normal(n))
c0, A0, b0 = O.convert_to_equalities(c, A, b, G, h)
for dev, blk in (((1, 1),) if os.environ.get("GOMILP_DEBUG_GS") else ((1, 1), (1, 0), (0, 0))):
    cx = lp.Context(general_device=dev, general_block=blk)
    p = cx.upload(c0, A0, b0)
    for i in range(3):
        t0 = time.perf_counter(); g = p.solve(0.0); dt = time.perf_counter() - t0
        s = g.stats
        print("search on", ("device, blocked   " if blk else "device, one by one") if dev else "host              ", "rows", 2 * m, "cols", n + m, "status", g.status, "total %.1f ms" % (1e3 * dt), "loop %.1f final %.1f" % (1e3 * s["seconds_pivot_loop"], 1e3 * s["seconds_final_solve"]),
              "pivots", s["pivots_phase1"], s["pivots_phase2"], flush=True)
    cx.close()
t0 = time.perf_counter(); idx = lp.find_independent(A0); print("search alone %.1f ms" % (1e3 * (time.perf_counter() - t0)), len(idx))
