"""Developer tool: the blocked device column search (general_block.hip) against the per-candidate device form and the host form —
the same accepted columns — and what each costs.  usage: general_block_check.py [quick]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp
quick = len(sys.argv) > 1
bad = 0
# the family of test_device_column_search_matches_the_host_search: 2..60 rows, zero / duplicated / scaled columns, integer data
for seed in range(60 if quick else 200):
    r = np.random.default_rng(1000 + seed)
    m = int(r.integers(2, 60)); n = int(r.integers(m, m + 40))
    if seed % 3 == 1:
        A = r.integers(-2, 3, (m, n)).astype(float)
    elif seed % 3 == 2:
        A = r.standard_normal((m, n)); A[:, r.integers(0, n, 3)] = 0; A[:, -1] = A[:, -2] * 2
    else:
        A = r.standard_normal((m, n)); A[r.random((m, n)) < 0.3] = 0
    try:
        blk = lp.find_independent_device(A, general_block=1)
    except RuntimeError as e:
        blk = str(e)
    try:
        one = lp.find_independent_device(A, general_block=0)
    except RuntimeError as e:
        one = str(e)
    host = lp.find_independent(A, True)
    if blk != one or (isinstance(one, list) and one != host):
        bad += 1
        print("MISMATCH seed", seed, m, n, "blocked", blk if isinstance(blk, str) else len(blk), "one", one if isinstance(one, str) else len(one), "host", len(host), flush=True)
print("small family: mismatches", bad, flush=True)
def eq_form(mg, ng, seed=5):
    rng = np.random.default_rng(seed)
    A0 = np.zeros((2 * mg, ng + mg)); A0[:mg, :ng] = rng.standard_normal((mg, ng)); A0[mg:, :ng] = rng.standard_normal((mg, ng)); A0[mg:, ng:] = np.eye(mg)
    return A0
def dep_form(m, n, seed):
    r = np.random.default_rng(seed)
    A = r.standard_normal((m, n)); A[r.random((m, n)) < 0.2] = 0.0
    A[:, n - 3] = 2.0 * A[:, n - 1]; A[:, n - 7] = A[:, n - 2] - A[:, n - 5]; A[:, n - 11] = 0.0
    A[:, n - 40] = A[:, n - 38] + A[:, n - 39]; A[:, n // 2] = 0.5 * A[:, n // 2 + 5]
    return A
cases = [("eq 400x560", eq_form(200, 360)), ("dep 320x400", dep_form(320, 400, 320)), ("eq 1000x1200", eq_form(500, 700)), ("dep 1000x1100", dep_form(1000, 1100, 7))]
if not quick:
    cases += [("dep 1500x1600 (NB 8)", dep_form(1500, 1600, 8)), ("eq 2400x2900 (NB 4)", eq_form(1200, 1700, 9))]
for name, A in cases:
    res = {}
    for gb in (1, 0):
        lp.find_independent_device(A, general_block=gb)
        t0 = time.perf_counter(); res[gb] = lp.find_independent_device(A, general_block=gb); dt = time.perf_counter() - t0
        print(name, "blocked" if gb else "one by one", "%.1f ms (context + upload + search + square step)" % (1e3 * dt), flush=True)
    t0 = time.perf_counter(); host = lp.find_independent(A, True); dth = time.perf_counter() - t0
    ok = res[1] == res[0] == host
    print(name, "host %.1f ms;" % (1e3 * dth), "same columns:", ok, len(host), flush=True)
    if not ok:
        bad += 1
        d = [i for i in range(min(len(res[1]), len(host))) if res[1][i] != host[i]]
        print("  first difference at position", d[:1], "lengths", len(res[1]), len(res[0]), len(host))
print("TOTAL mismatches", bad)
