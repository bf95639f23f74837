"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

ctypes front-end for the C restatement of gonum's ``lp.Simplex`` (oracle/gonum_lp.c) plus a
small Python restatement of GoMILP's caller semantics:

* ``convert_to_equalities``  — /root/reference/subproblem.go:81-139
* ``solve_milp``             — /root/reference/ilp.go:43-116, tree.go:66-263 (1 worker, FIFO),
                               branching.go:54-72, subproblem.go:141-259

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (gomilp_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from collections import deque
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgomilp_oracle.so")

# status codes (oracle/gonum_lp.h == include/gomilp_lp.h)
OK, ERR_BLAND, ERR_INFEASIBLE, ERR_LINSOLVE, ERR_UNBOUNDED, ERR_SINGULAR = 0, 1, 2, 3, 4, 5
ERR_ZERO_COLUMN, ERR_ZERO_ROW, ERR_CONDITION, ERR_PHASE1_WRAPPED, ERR_BAD_SHAPE, ERR_PANIC = 6, 7, 8, 9, 10, 11

STATUS_NAMES = {
    OK: "ok", ERR_BLAND: "ErrBland", ERR_INFEASIBLE: "ErrInfeasible", ERR_LINSOLVE: "ErrLinSolve",
    ERR_UNBOUNDED: "ErrUnbounded", ERR_SINGULAR: "ErrSingular", ERR_ZERO_COLUMN: "ErrZeroColumn",
    ERR_ZERO_ROW: "ErrZeroRow", ERR_CONDITION: "mat.Condition", ERR_PHASE1_WRAPPED: "phase1-wrapped",
    ERR_BAD_SHAPE: "panic:badShape", ERR_PANIC: "panic",
}


class _Pivot(C.Structure):
    _fields_ = [("phase", C.c_int32), ("bland", C.c_int32), ("min_idx", C.c_int64), ("replace", C.c_int64),
                ("entering", C.c_int64), ("leaving", C.c_int64)]


class _Ctx(C.Structure):
    _fields_ = [
        ("fast_initial_basis", C.c_int32),
        ("stop_after_pivots", C.c_int64),
        ("trace", C.POINTER(_Pivot)),
        ("trace_cap", C.c_int64),
        ("trace_len", C.c_int64),
        ("pivots_phase1", C.c_int64), ("pivots_phase2", C.c_int64), ("bland_steps", C.c_int64),
        ("lu_factorizations", C.c_int64), ("cond_evaluations", C.c_int64),
        ("phase1_used", C.c_int32), ("truncated", C.c_int32), ("wrapped_code", C.c_int32),
        ("seconds_loop", C.c_double),
        ("art_exchanges", C.c_int64),
    ]


_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/libgomilp_oracle.so with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("gonum_lp.c", "gonum_linalg.c", "gonum_blas.h", "gonum_linalg.h", "gonum_lp.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
        _lib.oracle_lp_simplex.restype = C.c_int
        _lib.oracle_lp_simplex.argtypes = [dp, dp, C.c_int64, dp, C.c_int64, C.c_int64, C.c_double, ip, dp, dp,
                                           C.POINTER(C.c_int32), ip, C.POINTER(_Ctx)]
        _lib.oracle_ctx_init.argtypes = [C.POINTER(_Ctx)]
        _lib.oracle_find_linearly_independent.restype = C.c_int64
        _lib.oracle_find_linearly_independent.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, ip, C.POINTER(_Ctx)]
        _lib.g_set_threads.argtypes = [C.c_int]
        _lib.g_cond1.restype = C.c_double
        _lib.g_cond1.argtypes = [C.c_int64, C.c_int64, dp, C.c_int64]
        _lib.g_lu_cond_rowsum.restype = C.c_double
        _lib.g_lu_cond_rowsum.argtypes = [C.c_int64, dp, C.c_int64, C.c_int, C.POINTER(C.c_int)]
        _lib.g_dgetrf.restype = C.c_int
        _lib.g_dgetrf.argtypes = [C.c_int64, C.c_int64, dp, C.c_int64, ip]
        _lib.g_dgetf2.restype = C.c_int
        _lib.g_dgetf2.argtypes = [C.c_int64, C.c_int64, dp, C.c_int64, ip]
        _lib.g_solve_vec.restype = C.c_int
        _lib.g_solve_vec.argtypes = [C.c_int64, dp, C.c_int64, dp]
        _lib.g_solve_vec_trans.restype = C.c_int
        _lib.g_solve_vec_trans.argtypes = [C.c_int64, dp, C.c_int64, dp]
        _lib.g_dgeqrf.argtypes = [C.c_int64, C.c_int64, dp, C.c_int64, dp]
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def set_threads(n: int) -> None:
    lib().g_set_threads(int(n))


@dataclass
class LPResult:
    status: int
    z: float
    x: Optional[np.ndarray]           # None where the reference returns nil
    basis: Optional[np.ndarray]       # final basicIdxs, positional order
    pivots: List[tuple] = field(default_factory=list)   # (phase, bland, min_idx, replace, entering, leaving)
    pivots_phase1: int = 0
    pivots_phase2: int = 0
    bland_steps: int = 0
    lu_factorizations: int = 0
    cond_evaluations: int = 0
    phase1_used: bool = False
    truncated: bool = False
    wrapped_code: int = 0
    seconds_loop: float = 0.0
    art_exchanges: int = 0

    @property
    def ok(self) -> bool:
        return self.status == OK


def simplex(c, A, b, tol: float = 0.0, initial_basic=None, *, fast_initial_basis: bool = False,
            trace: bool = False, stop_after_pivots: int = -1, trace_cap: int = 1 << 16) -> LPResult:
    """lp.Simplex(c, A, b, tol, initialBasic) — simplex.go:88."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("A must be 2-D")
    m, n = A.shape
    if c.shape != (n,) or b.shape != (m,):
        return LPResult(ERR_BAD_SHAPE, math.nan, None, None)
    L = lib()
    ctx = _Ctx()
    L.oracle_ctx_init(C.byref(ctx))
    ctx.fast_initial_basis = 1 if fast_initial_basis else 0
    ctx.stop_after_pivots = int(stop_after_pivots)
    tbuf = None
    if trace:
        tbuf = (_Pivot * trace_cap)()
        ctx.trace = C.cast(tbuf, C.POINTER(_Pivot))
        ctx.trace_cap = trace_cap
    x = np.zeros(n, dtype=np.float64)
    basis = np.zeros(m, dtype=np.int64)
    z = C.c_double(math.nan)
    has_x = C.c_int32(0)
    ib = None
    if initial_basic is not None:
        ib = np.ascontiguousarray(initial_basic, dtype=np.int64)
    st = L.oracle_lp_simplex(_dp(c), _dp(A), n, _dp(b), m, n, float(tol), _ip(ib) if ib is not None else None,
                             C.byref(z), _dp(x), C.byref(has_x), _ip(basis), C.byref(ctx))
    piv = []
    if trace:
        for i in range(min(ctx.trace_len, trace_cap)):
            p = tbuf[i]
            piv.append((p.phase, p.bland, p.min_idx, p.replace, p.entering, p.leaving))
    return LPResult(
        status=st, z=z.value, x=x if has_x.value else None, basis=basis if has_x.value and m != n else None,
        pivots=piv, pivots_phase1=ctx.pivots_phase1, pivots_phase2=ctx.pivots_phase2, bland_steps=ctx.bland_steps,
        lu_factorizations=ctx.lu_factorizations, cond_evaluations=ctx.cond_evaluations,
        phase1_used=bool(ctx.phase1_used), truncated=bool(ctx.truncated), wrapped_code=ctx.wrapped_code,
        seconds_loop=ctx.seconds_loop, art_exchanges=ctx.art_exchanges)


def find_linearly_independent(A, fast: bool = False) -> List[int]:
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    ctx = _Ctx()
    lib().oracle_ctx_init(C.byref(ctx))
    ctx.fast_initial_basis = 1 if fast else 0
    idx = np.zeros(m, dtype=np.int64)
    cnt = lib().oracle_find_linearly_independent(_dp(A), n, m, n, _ip(idx), C.byref(ctx))
    return [int(v) for v in idx[:cnt]]


def cond1(A) -> float:
    A = np.ascontiguousarray(A, dtype=np.float64)
    r, c = A.shape
    return lib().g_cond1(r, c, _dp(A), c)


def lu_cond(A, trans: bool = False):
    """(cond, det_is_zero) of mat.LU after Factorize(A) (trans: Factorize(A.T())): what LU.Solve's two guards look at (mat/lu.go:301,321)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    assert A.shape == (n, n)
    dz = C.c_int(0)
    v = lib().g_lu_cond_rowsum(n, _dp(A), n, 1 if trans else 0, C.byref(dz))
    return float(v), bool(dz.value)


# --------------------------------------------------------------------------------------
# GoMILP caller semantics (host side; Python because it is control flow, not arithmetic)
# --------------------------------------------------------------------------------------

def convert_to_equalities(c, A, b, G, h):
    """subproblem.go:81-139: [[A, 0], [G, I]], c' = [c, 0], b' = [b; h].  A may be None."""
    c = np.asarray(c, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    nvar, nineq = c.shape[0], h.shape[0]
    ncons = 0 if A is None else np.asarray(A).shape[0]
    c_new = np.concatenate([c, np.zeros(nineq)])
    b_new = np.concatenate([np.zeros(0) if A is None else np.asarray(b, dtype=np.float64), h])
    a_new = np.zeros((ncons + nineq, nvar + nineq))
    if A is not None:
        a_new[:ncons, :nvar] = A
    a_new[ncons:, :nvar] = G
    a_new[ncons:, nvar:] = np.eye(nineq)
    return c_new, a_new, b_new


def max_fun_branch_point(c, integrality) -> int:
    """branching.go:54-72 — `candidateValue` is never updated, so this is the LAST integer index (or 0)."""
    cur = 0
    for i, v in enumerate(c):
        if integrality[i] and abs(v) >= 0.0:
            cur = i
    return cur


def _is_all_integer(k: float) -> bool:
    """tree.go:290-297: k == math.Trunc(k) (true for ±Inf, false for NaN)."""
    if math.isnan(k):
        return False
    if math.isinf(k):
        return True
    return k == math.trunc(k)


def feasible_for_ip(integrality, x) -> bool:
    """tree.go:276-288 — exact-equality integrality test."""
    assert len(integrality) == len(x)
    return all(_is_all_integer(float(xi)) for i, xi in enumerate(x) if integrality[i])


@dataclass
class BnbNode:
    id: int
    parent: int
    constraints: List[tuple]      # (branched_variable, sign(+1/-1), hsharp) — subproblem.go:36-44, gsharp = sign*e_var
    status: int = -1
    z: float = math.nan
    x: Optional[np.ndarray] = None
    decision: str = ""


@dataclass
class MilpResult:
    error: Optional[str]          # None | "DeadlineExceeded" | "NO_INTEGER_FEASIBLE_SOLUTION" | "panic:<...>"
    x: Optional[np.ndarray]
    z: float
    nodes: List[BnbNode]


SimplexFn = Callable[[np.ndarray, np.ndarray, np.ndarray], LPResult]


def child_standard_form(c0, A0, b0, constraints):
    """subproblem.go:55-78 + :152 — G♯ rows are ±e_j; returns (c, A, b) of the child relaxation."""
    K = len(constraints)
    G = np.zeros((K, len(c0)))
    h = np.zeros(K)
    for k, (var, sign, rhs) in enumerate(constraints):
        G[k, var] = float(sign)
        h[k] = rhs
    return convert_to_equalities(c0, A0, b0, G, h)


def solve_milp(c, A, b, G, h, integrality, *, max_nodes: int = 10_000, simplex_fn: Optional[SimplexFn] = None) -> MilpResult:
    """milpProblem.solve with ONE worker (deterministic FIFO order): ilp.go:75-116, tree.go:66-263.

    `max_nodes` stands in for the context deadline: when the budget of solved nodes is spent the
    result is ("DeadlineExceeded", incumbent-or-zero-solution) like ilp.go:93-100.
    """
    if simplex_fn is None:
        simplex_fn = lambda cc, AA, bb: simplex(cc, AA, bb, 0.0, None)
    c = np.asarray(c, dtype=np.float64)
    integrality = list(integrality)
    assert len(integrality) == len(c)
    # toInitialSubproblem, ilp.go:43-72
    if G is not None:
        c0, A0, b0 = convert_to_equalities(c, A, b, G, h)
        int0 = integrality + [False] * (len(c0) - len(c))
    else:
        c0, A0, b0 = c, np.asarray(A, dtype=np.float64), np.asarray(b, dtype=np.float64)
        int0 = integrality
    nodes: List[BnbNode] = []
    root = BnbNode(0, 0, [])
    nodes.append(root)
    res = simplex_fn(c0, A0, b0)  # subproblem.go:172
    root.status, root.z, root.x = res.status, res.z, res.x
    if res.status != OK:
        return MilpResult("panic:" + STATUS_NAMES[res.status], None, math.nan, nodes)  # subproblem.go:173-176
    if feasible_for_ip(int0, res.x):
        root.decision = "INITIAL_RX_FEASIBLE_FOR_IP"
        return MilpResult(None, res.x[: len(c)].copy(), res.z, nodes)
    incumbent: Optional[BnbNode] = None
    queue = deque()
    next_id = [0]

    def check(node: BnbNode) -> Optional[str]:
        nonlocal incumbent
        inc_z = math.inf if incumbent is None else incumbent.z
        if node.status != OK:
            if node.status == ERR_INFEASIBLE:
                node.decision = "SUBPROBLEM_IS_DEGENERATE"      # ilp.go:37-40 (labels are swapped there)
            elif node.status == ERR_SINGULAR:
                node.decision = "SUBPROBLEM_NOT_FEASIBLE"
            else:
                return "panic:" + STATUS_NAMES[node.status]     # tree.go:266-273
        elif inc_z <= node.z:
            node.decision = "WORSE_THAN_INCUMBENT"
        elif inc_z > node.z:
            if feasible_for_ip(int0, node.x):
                incumbent = node
                node.decision = "BETTER_THAN_INCUMBENT_FEASIBLE"
            else:
                node.decision = "BETTER_THAN_INCUMBENT_BRANCHING"
                j = max_fun_branch_point(c0, int0)
                fl = math.floor(node.x[j])
                for sign, rhs in ((1, fl), (-1, -(fl + 1))):     # subproblem.go:215-218
                    next_id[0] += 1
                    ch = BnbNode(next_id[0], node.id, node.constraints + [(j, sign, float(rhs))])
                    nodes.append(ch)
                    queue.append(ch)
        else:
            return "panic:unexpected case"
        return None

    err = check(root)
    if err:
        return MilpResult(err, None, math.nan, nodes)
    solved = 0
    while queue:
        if solved >= max_nodes:
            if incumbent is not None:
                return MilpResult("DeadlineExceeded", incumbent.x.copy(), incumbent.z, nodes)
            return MilpResult("DeadlineExceeded", None, 0.0, nodes)
        node = queue.popleft()
        cc, AA, bb = child_standard_form(c0, A0, b0, node.constraints)
        r = simplex_fn(cc, AA, bb)
        solved += 1
        node.status, node.z = r.status, r.z
        node.x = None if r.x is None else (r.x[: len(c0)].copy() if r.status == OK else r.x)
        err = check(node)
        if err:
            return MilpResult(err, None, math.nan, nodes)
    if incumbent is None:
        return MilpResult("NO_INTEGER_FEASIBLE_SOLUTION", None, 0.0, nodes)
    return MilpResult(None, incumbent.x[: len(c)].copy(), incumbent.z, nodes)
