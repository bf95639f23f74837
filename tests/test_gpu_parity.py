"""GPU parity tests: the HIP path (through the C-ABI of include/gomilp_lp.h) against the CPU oracle
on the same seeded inputs, and against the reference's golden vectors.  Bar: identical status,
identical pivot sequence, identical final basis, x and z BIT-IDENTICAL (fp64)."""
import json
import math
import os

import numpy as np
import pytest

from gomilp_amd import lp, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


@pytest.fixture(scope="module")
def ctx():
    c = lp.Context()
    yield c
    c.close()


def _same_trace(g, o):
    return [(p[0], p[2], p[3], p[4], p[5]) for p in g] == [(p[0], p[2], p[3], p[4], p[5]) for p in o]


def _check_against_oracle(ctx, c, A, b, tol=0.0, expect_bitwise=True):
    o = O.simplex(c, A, b, tol, None, fast_initial_basis=True, trace=True)
    rl = ctx.upload(c, A, b)
    g = rl.solve(tol, trace=True)
    rl.free()
    assert g.status == o.status, (lp.STATUS_NAMES[g.status], O.STATUS_NAMES[o.status], g.stats)
    if o.x is None:
        assert g.x is None
        assert (math.isnan(g.z) and math.isnan(o.z)) or g.z == o.z
        return g, o
    assert _same_trace(g.pivots, o.pivots), "pivot sequence differs"
    assert g.stats["pivots_phase1"] == o.pivots_phase1 and g.stats["pivots_phase2"] == o.pivots_phase2
    assert np.array_equal(g.basis, o.basis)
    if expect_bitwise:
        assert np.array_equal(g.x, o.x), float(np.max(np.abs(g.x - o.x)))
        assert g.z == o.z
    return g, o


def test_loaded_native_library():
    L = lp.lib()
    assert L.gomilp_device_count() >= 1
    assert L.gomilp_compiled_arch() == b"gfx950"


def test_golden_k1_flat_call():
    kat = [k for k in KATS["lp"] if k["id"] == "K1-lp"][0]
    r = lp.simplex(kat["c"], kat["A"], kat["b"], 0.0, None)
    assert r.status == lp.OK
    assert np.array_equal(r.x, np.array(kat["want_x"], dtype=float)) and r.z == kat["want_z"]


def test_golden_k9_singular():
    kat = [k for k in KATS["lp"] if k["id"] == "K9"][0]
    r = lp.simplex(kat["c"], kat["A"], kat["b"], 0.0, None)
    assert r.status == lp.ERR_SINGULAR and r.x is None and math.isnan(r.z)


@pytest.mark.parametrize("m,seed", [(3, 11), (8, 5), (16, 7), (33, 9), (64, 7), (128, 7), (256, 7)])
def test_dense_lp_matches_oracle_bitwise(ctx, m, seed):
    c, A, b = synth.dense_lp_standard_form(m, seed)
    g, o = _check_against_oracle(ctx, c, A, b)
    assert g.stats["pivots_phase2"] > 0


PIPE_KNOBS = {"blocked": dict(tableau=1, blocked=1), "tableau": dict(tableau=1, blocked=0),
              "fused": dict(tableau=0, fused=1), "three-kernel": dict(tableau=0, fused=0)}


@pytest.mark.parametrize("m,seed,pipe", [(128, 9, "fused"), (128, 9, "three-kernel"), (256, 11, "fused"), (512, 3, "fused"),
                                         (512, 3, "three-kernel"), (512, 3, "tableau"), (300, 5, "tableau"),
                                         (512, 3, "blocked"), (300, 5, "blocked"), (64, 7, "blocked"), (300, 5, "blocked16")])
def test_all_pivot_pipelines_match_oracle(m, seed, pipe):
    """Four device formulations of the pivot: blocked tableau with deferred rank-K updates (default when
    n-m < 2m), single-kernel tableau, fused two-kernel revised simplex (ld = 128*NV), three-kernel revised simplex
    (any shape).  Each must reproduce the oracle's pivot sequence and the reference's bits."""
    c, A, b = synth.dense_lp_standard_form(m, seed)
    cx = lp.Context(chunk=16, block_k=16 if pipe == "blocked16" else 0, **PIPE_KNOBS[pipe.replace("16", "")])
    pipe = pipe.replace("16", "")
    try:
        g, o = _check_against_oracle(cx, c, A, b)
        assert g.stats["pipeline"] == pipe
    finally:
        cx.close()


@pytest.mark.parametrize("pipe", ["blocked", "tableau", "three-kernel"])
@pytest.mark.parametrize("signs", [(-1,), (1, -1, 1)])
def test_children_on_each_pipeline(pipe, signs):
    c, A, b = _child(24, 3, signs)
    cx = lp.Context(chunk=8, **PIPE_KNOBS[pipe])
    try:
        g, o = _check_against_oracle(cx, c, A, b)
        assert g.stats["pipeline"] == pipe
    finally:
        cx.close()


def test_rectangular_more_columns(ctx):
    c, A, b = synth.dense_lp_standard_form(40, 21, nv=90)
    _check_against_oracle(ctx, c, A, b)


def _child(m, seed, signs):
    """root LP + bnb rows like subproblem.go:141-159: returns (c, A, b) of a child relaxation"""
    c0, A0, b0 = synth.dense_lp_standard_form(m, seed)
    root = O.simplex(c0, A0, b0, 0.0, None, fast_initial_basis=True)
    frac = [j for j in range(m) if root.x[j] != math.floor(root.x[j])]
    cons = []
    for k, s in enumerate(signs):
        j = frac[-1 - k]
        fl = math.floor(root.x[j])
        cons.append((j, 1, float(fl)) if s > 0 else (j, -1, float(-(fl + 1))))
    return O.child_standard_form(c0, A0, b0, cons)


@pytest.mark.parametrize("signs", [(-1,), (1,), (-1, -1), (1, -1, 1)])
def test_bnb_children_phase1_and_bland(ctx, signs):
    c, A, b = _child(24, 3, signs)
    _check_against_oracle(ctx, c, A, b)


def test_infeasible_child(ctx):
    # x_0 >= 5 against rows that cap every x below ~2: Phase I ends with the artificial positive
    c0, A0, b0 = synth.dense_lp_standard_form(12, 4)
    c, A, b = O.child_standard_form(c0, A0, b0, [(0, -1, -50.0)])
    g, o = _check_against_oracle(ctx, c, A, b)
    assert g.status == lp.ERR_INFEASIBLE


def test_unbounded(ctx):
    # minimise -x0 with x0 - x1 + s = 1: ray along (1,1)
    c = np.array([-1.0, 0.0, 0.0])
    A = np.array([[1.0, -1.0, 1.0]])
    b = np.array([1.0])
    g, o = _check_against_oracle(ctx, c, A, b)
    assert g.status == lp.ERR_UNBOUNDED and g.z == -math.inf


def test_verify_inputs_errors(ctx):
    c = np.array([-1.0, 0.0, 0.0, 0.0])
    A = np.array([[1.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    b = np.array([1.0, 1.0])
    g, o = _check_against_oracle(ctx, c, A, b)  # zero column with c >= 0 -> ErrZeroColumn
    assert g.status == lp.ERR_ZERO_COLUMN
    A2 = np.array([[1.0, 1.0, 1.0, 0.0], [0.0, 0.0, 0.0, 0.0]])
    g, o = _check_against_oracle(ctx, c, A2, np.array([1.0, 1.0]))
    assert g.status == lp.ERR_INFEASIBLE
    g, o = _check_against_oracle(ctx, c, A2, np.array([1.0, 0.0]))
    assert g.status == lp.ERR_ZERO_ROW


def test_exactly_constrained_m_equals_n(ctx):
    rng = np.random.default_rng(5)
    A = rng.uniform(0.1, 1.0, size=(9, 9)) + 4 * np.eye(9)
    x = rng.uniform(0.5, 1.5, size=9)
    b = A @ x
    c = rng.uniform(-1, 1, size=9)
    _check_against_oracle(ctx, c, A, b)


@pytest.mark.parametrize("m,seed", [(200, 2), (448, 3), (512, 3), (513, 3), (700, 3), (1024, 1), (2048, 2)])
def test_blocked_lu_equals_per_column_lu_bitwise(m, seed):
    """The blocked final solve (lu_kernels.hip: register panel + trailing rank-nb update) must give the same bits
    as the one-launch-per-column schedule (simplex_kernels.hip k_lu_step), which the small cases pin to the oracle."""
    c, A, b = synth.dense_lp_standard_form(m, seed)
    res = []
    for blocked in (3, 2, 1, 0):   # compressed rounds with the look-ahead schedule (default), the same with the whole update behind each panel, blocked panels, one launch per column
        cx = lp.Context(lu_blocked=blocked)
        try:
            rl = cx.upload(c, A, b)
            res.append(rl.solve(0.0))
            rl.free()
        finally:
            cx.close()
    for r in res[:3]:
        assert r.status == lp.OK == res[3].status
        assert np.array_equal(r.basis, res[3].basis)
        assert np.array_equal(r.x, res[3].x) and r.z == res[3].z
    assert res[0].stats["lu_rounds"] == res[1].stats["lu_rounds"] > 0 and res[2].stats["lu_rounds"] == 0 == res[3].stats["lu_rounds"]   # (rounds: the compressed schedules only; the look-ahead changes when an update runs, not what a round is)
    assert res[0].stats["lu_dense_steps"] == res[1].stats["lu_dense_steps"] == res[2].stats["lu_dense_steps"]
    # the unit-column fast path of the panel kernel must be deterministic (it once raced: repeat the solve)
    cx = lp.Context()
    try:
        rl = cx.upload(c, A, b)
        for _ in range(3):
            again = rl.solve(0.0)
            assert again.status == lp.OK and np.array_equal(again.x, res[3].x) and again.z == res[3].z
        rl.free()
    finally:
        cx.close()


def _lu_ties():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lu_ties", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lu_ties.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("m", [40, 150, 300])
def test_lu_schedules_agree_where_the_pivot_search_ties(m):
    """Integer data (entries 0..3, half of them zero): most pivot searches of the final solve find several rows with the same
    |a_ik|, and dgetf2.go:38 takes the first in LAPACK's logical row order — the order the row interchanges of every earlier
    step, bookkeeping steps included, have produced.  The compressed schedules (lu_blocked 3 / 2: index maps in LDS) against the
    blocked panels and the one-launch-per-column form (1 / 0: kernels with a pivot search of their own): same basis, x and z bits;
    and the oracle's.  (Seeds on which the reference's rule terminates: see the next test.  The pivot budget only guards the run.)"""
    mod = _lu_ties()
    for seed in (0, 2):
        c, A, b = mod.integer_lp(m, seed)
        want = O.simplex(c, A, b, stop_after_pivots=5000)
        assert not want.truncated
        res = {}
        for blocked in (3, 2, 1, 0):
            cx = lp.Context(lu_blocked=blocked, max_pivots=5000)
            try:
                res[blocked] = cx.upload(c, A, b).solve(0.0)
            finally:
                cx.close()
        ref = res[0]
        for k in (3, 2, 1):
            assert res[k].status == ref.status, (m, seed, k)
            assert np.array_equal(res[k].basis, ref.basis) and np.array_equal(res[k].x, ref.x) and res[k].z == ref.z, (m, seed, k)
        assert ref.status == want.status == lp.OK
        if m <= 256:   # (exact steps on fresh gonum-order solves up to 256 rows: the reference's path through the degenerate vertices, bit for bit)
            assert ref.stats["pivots_phase1"] + ref.stats["pivots_phase2"] == want.pivots_phase1 + want.pivots_phase2
            assert np.array_equal(ref.x, np.asarray(want.x)) and ref.z == want.z, (m, seed)


@pytest.mark.parametrize("name", ["ties150", "ties300", "C3", "C2", "M"])
def test_cross_workgroup_lu_panel_equals_the_one_workgroup_panel_bitwise(name):
    """lu_cross.hip (opt-in, knob lu_cross): a round's rows on 2 / 4 / 8 workgroups of one XCD — replicated index maps and slot tables, one
    exchange per dense step whose record carries the candidate's pivot-row entries, the bookkeeping runs off the chain (a row knows the
    step that retires it; the maps are replayed from the log of pivot rows when a pivot search ties: the integer cases, most of their
    steps) — against the one-workgroup panel: same pivots, basis, x and z bits, same dense steps, no fall-back."""
    if name.startswith("ties"):
        c, A, b = _lu_ties().integer_lp(int(name[4:]), 0)
    else:
        m, seed = synth.CONFIGS[name]
        c, A, b = synth.dense_lp_standard_form(m, seed)
    res = {}
    for cross in (0, 1):
        cx = lp.Context(lu_cross=cross, lu_blocked=2, max_pivots=20000)
        try:
            res[cross] = cx.upload(c, A, b).solve(0.0)
        finally:
            cx.close()
    a, x = res[0], res[1]
    assert a.status == x.status == lp.OK
    assert np.array_equal(a.basis, x.basis) and np.array_equal(a.x, x.x) and a.z == x.z
    assert a.stats["lu_dense_steps"] == x.stats["lu_dense_steps"] and x.stats["lu_rounds"] >= a.stats["lu_rounds"] > 0
    assert x.stats.get("device_retries", 0) == 0


def test_engine_follows_the_reference_into_its_cycle():
    """lp.Simplex has no iteration limit (simplex.go:233), and on the 40-row integer LP of seed 1 its rule does not terminate: from
    pivot 24 on two columns trade places at position 14 for ever — each exchange a NON-degenerate step by a rounding-size amount, so
    the Bland branch (:268-277) never takes over.  The oracle shows it (3000 pivots, truncated); the engine, which takes the same
    decisions, walks the same cycle: with a budget of 400 pivots both traces are equal pivot by pivot.  (A caller that wants an end
    sets `max_pivots`; the reference offers none.)"""
    mod = _lu_ties()
    c, A, b = mod.integer_lp(40, 1)
    want = O.simplex(c, A, b, trace=True, stop_after_pivots=400)
    assert want.truncated and len(want.pivots) == 400
    assert list(want.pivots[398][4:6]) == list(want.pivots[396][4:6]) and list(want.pivots[399][4:6]) == list(want.pivots[397][4:6])   # the cycle of two
    cx = lp.Context(max_pivots=400)
    try:
        got = cx.upload(c, A, b).solve(0.0, trace=True)
    finally:
        cx.close()
    assert got.status == lp.ERR_UNSUPPORTED and len(got.pivots) == 400      # the pivot budget ran out, as asked
    five = lambda tr: [(p[0], p[2], p[3], p[4], p[5]) for p in tr]   # phase, minIdx, replace, entering, leaving (as every trace test: the flag of the step that found them is the engine's own)
    assert five(got.pivots) == five(want.pivots)


def test_full_size_properties_C2(ctx):
    """1024x2048 (BASELINE config C2): size-independent checks — primal/dual feasibility, complementary
    slackness through an independent LAPACK solve, objective against HiGHS."""
    from scipy.optimize import linprog
    m, seed = synth.CONFIGS["C2"]
    cc, G, h = synth.dense_lp_inequality_form(m, seed)
    c, A, b = synth.dense_lp_standard_form(m, seed)
    rl = ctx.upload(c, A, b)
    g = rl.solve(0.0)
    rl.free()
    assert g.status == lp.OK
    x = g.x
    assert np.all(x >= 0)
    assert np.max(np.abs(A @ x - b)) <= 1e-9
    B = A[:, g.basis]
    y = np.linalg.solve(B.T, c[g.basis])
    r = c - A.T @ y
    assert r.min() >= -1e-9            # optimality of the final basis (simplex.go:248 with tol = 0, up to rounding)
    assert abs(c @ x - g.z) <= 1e-12 * max(1, abs(g.z))
    ref = linprog(cc, A_ub=G, b_ub=h, method="highs")
    assert abs(ref.fun - g.z) <= 1e-9 * max(1.0, abs(g.z))
    assert g.stats["drift_xb"] < 1e-8


def _frontier_case(m, seed, nvars):
    c0, A0, b0 = synth.dense_lp_standard_form(m, seed)
    root = O.simplex(c0, A0, b0, 0.0, None, fast_initial_basis=True)
    mask = synth.integrality_mask(m, m)
    children = synth.frontier_children(root.x, mask, nvars)
    return c0, A0, b0, children


def test_device_child_assembly_equals_host_assembly(ctx):
    c0, A0, b0, children = _frontier_case(24, 3, 3)
    root = ctx.upload(c0, A0, b0)
    for cons in children:
        ch = root.child(cons)
        g = ch.solve(0.0, trace=True)
        ch.free()
        cc, AA, bb = O.child_standard_form(c0, A0, b0, cons)
        o = O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True, trace=True)
        assert g.status == o.status
        if o.x is not None:
            assert _same_trace(g.pivots, o.pivots)
            assert np.array_equal(g.x, o.x) and g.z == o.z
    root.free()


@pytest.mark.parametrize("workers", [1, 4])
def test_frontier_pool_matches_oracle(workers):
    c0, A0, b0, children = _frontier_case(128, 3, 4)
    assert len(children) == 16
    pool = lp.FrontierPool(workers=workers)
    try:
        pool.set_root(c0, A0, b0)
        res = pool.solve(children)
    finally:
        pool.close()
    n0 = A0.shape[1]
    statuses = set()
    for i, cons in enumerate(children):
        cc, AA, bb = O.child_standard_form(c0, A0, b0, cons)
        o = O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True)
        statuses.add(o.status)
        assert res.status[i] == o.status, (i, res.status[i], o.status)
        if o.status == O.OK:
            assert res.has_x[i] == 1
            assert np.array_equal(res.x[i], o.x[:n0]) and res.z[i] == o.z
    assert res.stats["relaxations"] == len(children) and res.stats["workers"] == workers
    assert O.OK in statuses


# ---- reference golden vectors through the C-ABI (general initial basis: no slack structure in K2-K8) ----------------

def _gpu_simplex_fn(ctx):
    def fn(c, A, b):
        rl = ctx.upload(c, A, b)
        r = rl.solve(0.0)
        rl.free()
        return O.LPResult(status=r.status, z=r.z, x=r.x, basis=r.basis)
    return fn


@pytest.mark.parametrize("kat", KATS["milp"], ids=[k["id"] for k in KATS["milp"]])
def test_reference_milp_goldens_on_gpu(ctx, kat):
    """K1-K8 of /root/reference/ilp_test.go and api_test.go: every LP relaxation of the branch-and-bound runs on the
    GPU (through the C-ABI); the expected x and z are the literals the reference asserts with reflect.DeepEqual / ==."""
    arr = lambda v: None if v is None else np.array(v, dtype=np.float64)
    res = O.solve_milp(arr(kat["c"]), arr(kat["A"]), arr(kat["b"]), arr(kat["G"]), arr(kat["h"]), kat["integrality"],
                       max_nodes=60, simplex_fn=_gpu_simplex_fn(ctx))
    assert res.error == kat["want_err"]
    if kat["want_err"] is None:
        assert np.array_equal(res.x, arr(kat["want_x"])), (res.x.tolist(), kat["want_x"])
        if kat["want_z"] is not None:
            assert res.z == kat["want_z"]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_general_initial_basis_random_equality_lps(ctx, seed):
    """LPs with equality rows and no slack structure (getRandomMILP-style N(0,1) data, ilp_test.go:370-429):
    findLinearlyIndependent's general case + Phase I from a non-permutation basis."""
    rng = np.random.default_rng(seed)
    m, n = 6 + seed, 15 + 2 * seed
    A = rng.standard_normal((m, n))
    x0 = np.abs(rng.standard_normal(n))
    b = A @ x0                      # feasible by construction
    c = np.abs(rng.standard_normal(n))  # bounded below on x >= 0
    _check_against_oracle_general(ctx, c, A, b)


def _check_against_oracle_general(ctx, c, A, b):
    o = O.simplex(c, A, b, 0.0, None, trace=True)
    rl = ctx.upload(c, A, b)
    g = rl.solve(0.0, trace=True)
    rl.free()
    assert g.status == o.status, (lp.STATUS_NAMES[g.status], O.STATUS_NAMES[o.status])
    if o.x is not None:
        assert _same_trace(g.pivots, o.pivots)
        assert np.array_equal(g.basis, o.basis)
        assert np.array_equal(g.x, o.x) and g.z == o.z


# ---- host B&B driver (gomilp_amd/bnb.py) against the oracle's restatement of tree.go ----------------------------------

@pytest.mark.parametrize("m,seed,budget", [(16, 3, 15), (24, 5, 31)])
def test_bnb_driver_matches_oracle_tree(m, seed, budget):
    from gomilp_amd import bnb
    c, G, h = synth.dense_lp_inequality_form(m, seed)
    integrality = [j % 4 == 0 for j in range(m)]
    want = O.solve_milp(c, None, None, G, h, integrality, max_nodes=budget)
    got = bnb.solve_milp(c, None, None, G, h, integrality, max_nodes=budget, workers=4)
    assert got.error == want.error
    assert len(got.nodes) == len(want.nodes)
    for a, b_ in zip(got.nodes, want.nodes):
        assert (a.id, a.parent, a.constraints) == (b_.id, b_.parent, b_.constraints)
        if b_.status == -1:          # never solved (budget exhausted)
            assert a.status == -1
            continue
        assert a.status == b_.status and a.decision == b_.decision
        if b_.status == O.OK:
            assert a.z == b_.z and np.array_equal(a.x, b_.x[: len(a.x)])
    if want.x is not None:
        assert np.array_equal(got.x, want.x[: len(got.x)]) and got.z == want.z


def test_bnb_driver_reference_goldens():
    from gomilp_amd import bnb
    arr = lambda v: None if v is None else np.array(v, dtype=np.float64)
    for kat in KATS["milp"]:
        res = bnb.solve_milp(arr(kat["c"]), arr(kat["A"]), arr(kat["b"]), arr(kat["G"]), arr(kat["h"]), kat["integrality"],
                             max_nodes=40, workers=2)
        assert res.error == kat["want_err"], kat["id"]
        if kat["want_err"] is None:
            assert np.array_equal(res.x, arr(kat["want_x"])), kat["id"]
            if kat["want_z"] is not None:
                assert res.z == kat["want_z"], kat["id"]


@pytest.mark.parametrize("m,seed,after", [(40, 5, 0), (40, 5, 7), (96, 2, 25), (200, 3, 60)])
def test_supplied_initial_basis(ctx, m, seed, after):
    """initialBasic != nil (simplex.go:147-160): Phase I is skipped, the pivots continue from the supplied vertex.
    The basis after `after` pivots of a plain solve is feasible by construction."""
    c, A, b = synth.dense_lp_standard_form(m, seed)
    n = A.shape[1]
    plain = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, trace=True)
    basic = [n - 1 - pos for pos in range(m)]          # slack basis in findLinearlyIndependent's scan order
    for p in plain.pivots[:after]:
        basic[p[3]] = p[4]                               # (phase, bland, min_idx, replace, entering, leaving)
    o = O.simplex(c, A, b, 0.0, basic, trace=True)
    rl = ctx.upload(c, A, b)
    g = rl.solve(0.0, trace=True, initial_basic=basic)
    assert g.status == o.status == lp.OK
    assert _same_trace(g.pivots, o.pivots)
    assert np.array_equal(g.basis, o.basis) and np.array_equal(g.x, o.x) and g.z == o.z
    # an infeasible vertex and a singular set panic in the reference (initializeFromBasic)
    bad = list(basic)
    bad[0] = bad[1]
    assert rl.solve(0.0, initial_basic=bad).status == lp.ERR_PANIC == O.simplex(c, A, b, 0.0, bad).status
    rl.free()


def test_random_roots_and_children_sweep(ctx):
    """A small slice of tools/parity_sweep.py (300 random cases, 0 mismatches when this test was written): random
    sizes, every second case a child with 1-5 branching rows (Phase I, Bland steps, the artificial exchange)."""
    rng = np.random.default_rng(2024)
    for k in range(14):
        m = int(rng.integers(8, 160)); seed = int(rng.integers(1, 10**6))
        c, A, b = synth.dense_lp_standard_form(m, seed)
        if k % 2 == 1:
            r0 = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True)
            frac = [j for j in range(m) if r0.x[j] != np.floor(r0.x[j])]
            K = int(rng.integers(1, min(5, len(frac)) + 1))
            cons = []
            for j in rng.choice(frac, size=K, replace=False):
                fl = float(np.floor(r0.x[j]))
                cons.append((int(j), 1, fl) if rng.random() < 0.5 else (int(j), -1, -(fl + 1)))
            c, A, b = O.child_standard_form(c, A, b, cons)
        _check_against_oracle(ctx, c, A, b)


@pytest.mark.parametrize("m,seed", [(1100, 14), (1300, 15), (2048, 2)])
def test_1024_thread_block_kernel_vs_single_kernel_pipeline(m, seed):
    """Sizes beyond 1024 rows run the 1024-thread register-resident kernel (LDS ring for one column slot's terms).  The
    CPU oracle needs minutes there (tools/parity_big.py: identical), so the suite checks the blocked pipeline against the
    single-kernel tableau pipeline — a different formulation validated against the oracle at the small sizes above:
    same pivot sequence, same basis, bit-identical x."""
    c, A, b = synth.dense_lp_standard_form(m, seed)
    res = []
    for blocked in (1, 0):
        cx = lp.Context(blocked=blocked)
        try:
            rl = cx.upload(c, A, b)
            res.append(rl.solve(0.0, trace=True))
            rl.free()
        finally:
            cx.close()
    g, t = res
    assert g.stats["pipeline"] == "blocked" and t.stats["pipeline"] == "tableau"
    assert g.status == lp.OK == t.status
    assert _same_trace(g.pivots, t.pivots)
    assert np.array_equal(g.basis, t.basis) and np.array_equal(g.x, t.x) and g.z == t.z
