#!/usr/bin/env python3
"""Generate the oracle fixtures under tests/golden/ for the BASELINE configs (run in the build container, CPU only).

The CPU oracle (oracle/, the reference algorithm: three fresh gonum-order LU factorizations per pivot) is too slow to
run inside the GPU test suite at the sizes the headline numbers are quoted on (2048x4096: ~20 CPU-minutes), so its
outputs are computed ONCE here and committed as data; tests/test_gpu_golden.py compares the HIP path with them.

    python tools/gen_golden.py c2 m c5 c3 c1 c4        # any subset; each target writes its own file

Files (numpy .npz, all arrays plain data — inputs are regenerated from the seeds by gomilp_amd/synth.py):
    lp_C2.npz / lp_M.npz     full pivot trace (phase, bland, min_idx, replace, entering, leaving), final basis, x, z
    lp_C4_prefix.npz         the first N Phase-II pivots of the 4096x8192 LP (a full solve is CPU-days)
    frontier_C5.npz          the 256 children of the 512x1024 root: status, z, has_x, x, pivot counts
    milp_C3.npz              FIFO branch-and-bound over the 512x1024 MILP, node budget 127: per node parent, constraints,
                             status, z, decision, x
    milp_C1.npz              the 10-variable / 5-constraint plumbing case (BASELINE config 1), whole tree
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

from gomilp_amd import synth
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
THREADS = int(os.environ.get("GOLDEN_THREADS", "6"))

DECISIONS = ["", "INITIAL_RX_FEASIBLE_FOR_IP", "SUBPROBLEM_IS_DEGENERATE", "SUBPROBLEM_NOT_FEASIBLE", "WORSE_THAN_INCUMBENT",
             "BETTER_THAN_INCUMBENT_FEASIBLE", "BETTER_THAN_INCUMBENT_BRANCHING"]


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def trace_array(piv):
    return np.array(piv, dtype=np.int32).reshape(-1, 6)


def trace_sha(tr: np.ndarray) -> str:
    """sha256 over (phase, min_idx, replace, entering, leaving) as little-endian int32 — the `bland` flag is not part
    of the reference's state, only of how the pair was found."""
    return hashlib.sha256(np.ascontiguousarray(tr[:, [0, 2, 3, 4, 5]].astype("<i4")).tobytes()).hexdigest()


def gen_lp(name: str, stop: int = -1):
    m, seed = synth.CONFIGS[name]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    O.set_threads(THREADS)
    t0 = time.time()
    cap = 1 << 16
    r = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True, stop_after_pivots=stop, trace_cap=cap)
    dt = time.time() - t0
    tr = trace_array(r.pivots)
    log(name, "m", m, "seed", seed, "status", r.status, "pivots", len(tr), "truncated", r.truncated, "%.0f s" % dt)
    fn = os.path.join(OUT, "lp_%s%s.npz" % (name, "_prefix" if stop >= 0 else ""))
    np.savez_compressed(
        fn, m=m, seed=seed, status=r.status, truncated=int(r.truncated), trace=tr, trace_sha256=trace_sha(tr),
        basis=np.zeros(0, np.int32) if r.basis is None else r.basis.astype(np.int32),
        x=np.zeros(0) if r.x is None else r.x, z=r.z, pivots_phase1=r.pivots_phase1, pivots_phase2=r.pivots_phase2,
        bland_steps=r.bland_steps, oracle_seconds=dt, oracle_threads=THREADS)
    log("wrote", fn)


def gen_c5(nvars: int = 8):
    m, seed = synth.CONFIGS["C5"]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    mask = synth.integrality_mask(m, m)
    O.set_threads(THREADS)
    t0 = time.time()
    root = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True)
    log("C5 root status", root.status, "pivots", len(root.pivots), "%.0f s" % (time.time() - t0))
    children = synth.frontier_children(root.x, mask, nvars)
    n0 = A.shape[1]
    cnt = len(children)
    status = np.zeros(cnt, np.int32)
    z = np.full(cnt, np.nan)
    has_x = np.zeros(cnt, np.int32)
    x = np.zeros((cnt, n0))
    piv = np.zeros((cnt, 3), np.int32)   # phase1, phase2, bland
    shas = []
    O.set_threads(1)
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        cc, AA, bb = O.child_standard_form(c, A, b, children[i])
        return i, O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True, trace=True)

    t0 = time.time()
    res = [None] * cnt
    with ThreadPoolExecutor(max_workers=THREADS) as ex:   # ctypes releases the GIL inside the oracle call
        for i, r in ex.map(one, range(cnt)):
            res[i] = r
            if (i + 1) % 32 == 0:
                log("C5 child", i + 1, "/", cnt, "%.0f s" % (time.time() - t0))
    for i, r in enumerate(res):
        status[i], z[i] = r.status, r.z
        if r.x is not None:
            has_x[i] = 1
            x[i] = r.x[:n0]
        piv[i] = (r.pivots_phase1, r.pivots_phase2, r.bland_steps)
        shas.append(trace_sha(trace_array(r.pivots)))
    cons = np.array([[(v, s, h) for (v, s, h) in ch] for ch in children], dtype=np.float64)   # cnt x K x 3
    fn = os.path.join(OUT, "frontier_C5.npz")
    np.savez_compressed(fn, m=m, seed=seed, nvars=nvars, root_x=root.x, root_z=root.z, root_trace_sha256=trace_sha(trace_array(root.pivots)),
                        constraints=cons, status=status, z=z, has_x=has_x, x=x, pivots=piv, trace_sha256=np.array(shas))
    log("wrote", fn, "feasible", int((status == 0).sum()), "infeasible", int((status == O.ERR_INFEASIBLE).sum()),
        "max pivots", int((piv[:, 0] + piv[:, 1]).max()), "max bland", int(piv[:, 2].max()))


def dump_tree(fn, res, n0, extra):
    nodes = [nd for nd in res.nodes if nd.status != -1]   # solved nodes only (queued children beyond the budget have none)
    K = max((len(nd.constraints) for nd in nodes), default=0)
    cons = np.zeros((len(nodes), max(K, 1), 3))
    ncons = np.zeros(len(nodes), np.int32)
    x = np.zeros((len(nodes), n0))
    has_x = np.zeros(len(nodes), np.int32)
    for i, nd in enumerate(nodes):
        ncons[i] = len(nd.constraints)
        for k, t in enumerate(nd.constraints):
            cons[i, k] = t
        if nd.x is not None:
            has_x[i] = 1
            x[i] = nd.x[:n0]
    np.savez_compressed(
        fn, ids=np.array([nd.id for nd in nodes], np.int32), parent=np.array([nd.parent for nd in nodes], np.int32),
        ncons=ncons, constraints=cons, status=np.array([nd.status for nd in nodes], np.int32),
        z=np.array([nd.z for nd in nodes]), decision=np.array([DECISIONS.index(nd.decision) for nd in nodes], np.int32),
        has_x=has_x, x=x, error="" if res.error is None else res.error, result_z=res.z,
        result_x=np.zeros(0) if res.x is None else res.x, **extra)
    log("wrote", fn, "nodes", len(nodes), "error", res.error)


def gen_c3(budget: int = 127):
    m, seed = synth.CONFIGS["C3"]
    c, G, h = synth.dense_lp_inequality_form(m, seed)
    integ = [j % 4 == 0 for j in range(m)]
    O.set_threads(THREADS)
    done = [0]
    t0 = time.time()

    def sf(cc, AA, bb):
        r = O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True)
        done[0] += 1
        if done[0] % 16 == 0:
            log("C3 node", done[0], "%.0f s" % (time.time() - t0))
        return r

    res = O.solve_milp(c, None, None, G, h, integ, max_nodes=budget, simplex_fn=sf)
    dump_tree(os.path.join(OUT, "milp_C3.npz"), res, 2 * m, dict(m=m, seed=seed, budget=budget))


def c1_problem(single: bool = False):
    """BASELINE config 1: 10 variables / 5 inequality constraints, seed 0, U(0,1) data like the other configs.
    Odd variables integer: the reference always branches on the LAST integer index (branching.go:54-72), so the tree
    does not terminate (like K8) and the node budget ends it; `single`: only x5 integer — the tree terminates."""
    c, G, h = synth.dense_lp_inequality_form(5, 0, nv=10)
    h = 3.0 * h   # room for integer points
    integ = [j == 5 for j in range(10)] if single else [j % 2 == 1 for j in range(10)]
    return c, G, h, integ


def gen_c1():
    for single in (False, True):
        c, G, h, integ = c1_problem(single)
        res = O.solve_milp(c, None, None, G, h, integ, max_nodes=63)
        dump_tree(os.path.join(OUT, "milp_C1%s.npz" % ("_single" if single else "")), res, 15, dict(budget=63))


def eq_problem(seed: int = 77, me: int = 200, nv: int = 320):
    """A MILP with 200 equality and 200 inequality rows (api.go:124 EqualTo; ilp.go:43-57 gives [[A, 0], [G, I]]: no slack basis, the
    column search of simplex.go:611-637 decides the start of every node): Gaussian data, feasible by construction."""
    rng = np.random.default_rng(seed)
    x0 = np.abs(rng.standard_normal(nv))
    A, G = rng.standard_normal((me, nv)), rng.standard_normal((me, nv))
    b, h = A @ x0, G @ x0 + np.abs(rng.standard_normal(me))
    c = np.abs(rng.standard_normal(nv))
    integ = [j % 5 == 0 for j in range(nv)]
    return c, A, b, G, h, integ


def gen_eq(budget: int = 30):
    c, A, b, G, h, integ = eq_problem()
    O.set_threads(THREADS)
    done = [0]
    t0 = time.time()

    def sf(cc, AA, bb):
        r = O.simplex(cc, AA, bb, 0.0, None)      # the reference's own O(m^4) column search per node
        done[0] += 1
        log("EQ node", done[0], "%.0f s" % (time.time() - t0))
        return r

    res = O.solve_milp(c, A, b, G, h, integ, max_nodes=budget, simplex_fn=sf)
    dump_tree(os.path.join(OUT, "milp_EQ.npz"), res, len(c) + G.shape[0], dict(budget=budget))


EQLP_CASES = [(5, 420, 300), (6, 900, 500)]      # (seed, n, m): 600- and 1000-row standard forms without a slack basis


def eqlp_problem(seed: int, n: int, m: int):
    """m equality + m inequality rows over n variables (the data of test_general_initial_basis_beyond_512_rows)."""
    rng = np.random.default_rng(seed)
    x0 = np.abs(rng.standard_normal(n))
    A = rng.standard_normal((m, n)); b = A @ x0
    G = rng.standard_normal((m, n)); h = G @ x0 + np.abs(rng.standard_normal(m))
    c = np.abs(rng.standard_normal(n))
    return O.convert_to_equalities(c, A, b, G, h)


def gen_eqlp():
    """The reference's own O(m^4) column search (simplex.go:611-637) and pivot loop on the two large equality LPs: start basis,
    pivot trace, final basis, x and z bits."""
    O.set_threads(THREADS)
    for seed, n, m in EQLP_CASES:
        c0, A0, b0 = eqlp_problem(seed, n, m)
        t0 = time.time()
        start = O.find_linearly_independent(A0)
        log("EQLP", 2 * m, "rows: column search", "%.0f s" % (time.time() - t0))
        r = O.simplex(c0, A0, b0, 0.0, None, trace=True)
        dt = time.time() - t0
        tr = trace_array(r.pivots)
        fn = os.path.join(OUT, "lp_EQ%d.npz" % (2 * m))
        np.savez_compressed(
            fn, seed=seed, n=n, m=m, status=r.status, start_basis=np.array(start, np.int32), trace=tr, trace_sha256=trace_sha(tr),
            basis=np.zeros(0, np.int32) if r.basis is None else r.basis.astype(np.int32), x=np.zeros(0) if r.x is None else r.x,
            z=r.z, pivots_phase1=r.pivots_phase1, pivots_phase2=r.pivots_phase2, bland_steps=r.bland_steps, oracle_seconds=dt,
            oracle_threads=THREADS)
        log("wrote", fn, "status", r.status, "pivots", len(tr), "%.0f s" % dt)


def main(argv):
    os.makedirs(OUT, exist_ok=True)
    O.build()
    targets = [a.lower() for a in argv] or ["c1", "c2", "c5", "c3", "m", "c4"]
    for t in targets:
        if t == "c2":
            gen_lp("C2")
        elif t == "m":
            gen_lp("M")
        elif t == "c4":
            gen_lp("C4", stop=int(os.environ.get("GOLDEN_C4_PIVOTS", "1000")))
        elif t == "c5":
            gen_c5()
        elif t == "c3":
            gen_c3()
        elif t == "c1":
            gen_c1()
        elif t == "eq":
            gen_eq()
        elif t == "eqlp":
            gen_eqlp()
        else:
            raise SystemExit("unknown target " + t)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
