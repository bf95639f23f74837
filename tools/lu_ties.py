"""Developer tool: the final solve's pivot search on integer data — many rows with the same |a_ik|, so the first one in LAPACK's logical row order
decides (dgetf2.go:38) and the compressed panel has to bring its index maps up to date (lu_compressed.hip).  The compressed schedules
(lu_blocked 3 / 2) against the blocked panels and the one-launch-per-column form (1 / 0: kernels with their own pivot search) — same bits.
(No oracle leg: the oracle's pivot loop does not terminate on integer_lp(40, 1) within a minute — a degenerate vertex where the
reference's three fresh solves per pivot keep trading the same columns; the engine ends after 31 pivots.)  usage: lu_ties.py [m ...]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp


def integer_lp(m, seed):
    r = np.random.default_rng(9000 + seed)
    nv = m
    G = r.integers(0, 4, (m, nv)).astype(float)
    G[r.random((m, nv)) < 0.5] = 0.0
    G[0] = np.maximum(G[0], 1.0)
    h = r.integers(1, 9, m).astype(float) * 4.0
    cc = -r.integers(1, 5, nv).astype(float)
    A = np.hstack([G, np.eye(m)]); c = np.concatenate([cc, np.zeros(m)])
    return c, A, h


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [40, 150, 300, 520, 700, 1100]
    bad = 0
    for m in sizes:
        for seed in range(2):
            c, A, b = integer_lp(m, seed)
            res = {}
            for blocked in (3, 2, 1, 0):
                cx = lp.Context(lu_blocked=blocked)
                p = cx.upload(c, A, b)
                t0 = time.perf_counter(); res[blocked] = p.solve(0.0); dt = time.perf_counter() - t0
                if os.environ.get("LU_TIES_VERBOSE"): print("  m %d seed %d lu_blocked %d: %.1f ms, status %d, pivots %d + %d" % (m, seed, blocked, 1e3 * dt, res[blocked].status, res[blocked].stats["pivots_phase1"], res[blocked].stats["pivots_phase2"]), flush=True)
                cx.close()
            ref = res[0]
            same = all(res[k].status == ref.status and np.array_equal(res[k].basis, ref.basis) and np.array_equal(res[k].x, ref.x) and res[k].z == ref.z for k in (3, 2, 1))
            line = "m %4d seed %d status %d pivots %d + %d dense steps %d rounds %d same bits over the four schedules: %s" % (
                m, seed, ref.status, ref.stats["pivots_phase1"], ref.stats["pivots_phase2"], res[3].stats["lu_dense_steps"], res[3].stats["lu_rounds"], same)
            print(line, flush=True)
            bad += 0 if same else 1
    print("TOTAL mismatches", bad)
