"""Host-side mirror of the reference interface for the hot path, over the C-ABI.

``simplex(c, A, b, tol, initial_basic)`` has the argument meaning and error behaviour of gonum's
``lp.Simplex`` (/root/reference/vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go:88) as GoMILP calls
it (/root/reference/subproblem.go:154,172).  Everything runs in libgomilp_hip.so on the MI355X; there is no
CPU fallback: a missing library or device raises / returns ERR_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import build as _build

# status codes — include/gomilp_lp.h
OK, ERR_BLAND, ERR_INFEASIBLE, ERR_LINSOLVE, ERR_UNBOUNDED, ERR_SINGULAR = 0, 1, 2, 3, 4, 5
ERR_ZERO_COLUMN, ERR_ZERO_ROW, ERR_CONDITION, ERR_PHASE1_WRAPPED, ERR_BAD_SHAPE, ERR_PANIC = 6, 7, 8, 9, 10, 11
ERR_DEVICE, ERR_UNSUPPORTED = 12, 13

STATUS_NAMES = {
    OK: "ok", ERR_BLAND: "ErrBland", ERR_INFEASIBLE: "ErrInfeasible", ERR_LINSOLVE: "ErrLinSolve",
    ERR_UNBOUNDED: "ErrUnbounded", ERR_SINGULAR: "ErrSingular", ERR_ZERO_COLUMN: "ErrZeroColumn",
    ERR_ZERO_ROW: "ErrZeroRow", ERR_CONDITION: "mat.Condition", ERR_PHASE1_WRAPPED: "phase1-wrapped",
    ERR_BAD_SHAPE: "panic:badShape", ERR_PANIC: "panic", ERR_DEVICE: "device", ERR_UNSUPPORTED: "unsupported",
}


class Stats(C.Structure):
    _fields_ = [
        ("pivots_phase1", C.c_int64), ("pivots_phase2", C.c_int64), ("bland_steps", C.c_int64),
        ("refreshes", C.c_int64), ("kernel_launches", C.c_int64),
        ("phase1_used", C.c_int32), ("device_id", C.c_int32), ("wrapped_status", C.c_int32), ("pipeline", C.c_int32),
        ("seconds_total", C.c_double), ("seconds_upload", C.c_double), ("seconds_pivot_loop", C.c_double),
        ("seconds_final_solve", C.c_double), ("drift_xb", C.c_double), ("pivot_kernel_seconds", C.c_double * 4),
        ("seconds_final_device", C.c_double), ("seconds_final_host", C.c_double), ("lu_dense_steps", C.c_int64),
        ("lu_rounds", C.c_int64), ("art_exchanges", C.c_int64), ("cond_fallbacks", C.c_int64), ("device_retries", C.c_int64),
        ("cond1_final", C.c_double), ("condinf_final", C.c_double),
    ]


class Pivot(C.Structure):
    _fields_ = [("phase", C.c_int32), ("bland", C.c_int32), ("min_idx", C.c_int64), ("replace", C.c_int64),
                ("entering", C.c_int64), ("leaving", C.c_int64)]


class FrontierStats(C.Structure):
    _fields_ = [("relaxations", C.c_int64), ("pivots_phase1", C.c_int64), ("pivots_phase2", C.c_int64),
                ("bland_steps", C.c_int64), ("phase1_runs", C.c_int64), ("kernel_launches", C.c_int64),
                ("workers", C.c_int32), ("device_id", C.c_int32), ("seconds_total", C.c_double),
                ("seconds_busy_sum", C.c_double), ("batched_relaxations", C.c_int64), ("host_fallbacks", C.c_int64),
                ("supersteps", C.c_int64), ("seconds_batch", C.c_double), ("blocks", C.c_int64), ("blocks_sampled", C.c_int64),
                ("seconds_inner_kernels", C.c_double), ("seconds_update_kernels", C.c_double),
                ("warm_started", C.c_int64), ("warm_fallbacks", C.c_int64), ("warm_kept", C.c_int64), ("pivots_dual", C.c_int64)]


EXPORTS = [
    "gomilp_frontier_solve_warm", "gomilp_pool_release_warm",
    "gomilp_lp_upload_child", "gomilp_pool_create", "gomilp_pool_destroy", "gomilp_pool_set", "gomilp_pool_set_root", "gomilp_frontier_solve", "gomilp_pool_add_root", "gomilp_frontier_solve_roots", "gomilp_pool_solve_root", "gomilp_debug_find_independent", "gomilp_debug_find_independent_device", "gomilp_debug_cond_estimate", "gomilp_debug_gonum_lu_cond",
    "gomilp_lp_simplex", "gomilp_ctx_create", "gomilp_ctx_destroy", "gomilp_ctx_device", "gomilp_ctx_set",
    "gomilp_lp_upload", "gomilp_lp_free", "gomilp_lp_solve_resident", "gomilp_lp_last_trace", "gomilp_version",
    "gomilp_device_count", "gomilp_compiled_arch", "gomilp_comm_unique_id", "gomilp_comm_create", "gomilp_comm_destroy",
    "gomilp_comm_rank", "gomilp_comm_world", "gomilp_incumbent_allreduce", "gomilp_incumbent_pick",
]

_lib = None


def lib():
    """Load libgomilp_hip.so (building it in-tree if hipcc is present and it is stale).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build()
    L = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.gomilp_version.restype = C.c_char_p
    L.gomilp_compiled_arch.restype = C.c_char_p
    L.gomilp_device_count.restype = C.c_int
    L.gomilp_ctx_create.restype = C.c_void_p
    L.gomilp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.gomilp_ctx_destroy.argtypes = [C.c_void_p]
    L.gomilp_ctx_device.argtypes = [C.c_void_p]
    L.gomilp_ctx_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.gomilp_lp_upload.restype = C.c_int64
    L.gomilp_lp_upload.argtypes = [C.c_void_p, dp, dp, C.c_int64, dp, C.c_int64, C.c_int64]
    L.gomilp_lp_free.argtypes = [C.c_void_p, C.c_int64]
    L.gomilp_lp_solve_resident.argtypes = [C.c_void_p, C.c_int64, C.c_double, ip, dp, dp, C.POINTER(C.c_int32), ip,
                                           C.POINTER(Stats)]
    L.gomilp_lp_last_trace.restype = C.c_int64
    L.gomilp_lp_last_trace.argtypes = [C.c_void_p, C.POINTER(Pivot), C.c_int64]
    L.gomilp_lp_simplex.argtypes = [dp, dp, C.c_int64, dp, C.c_int64, C.c_int64, C.c_double, ip, dp, dp,
                                    C.POINTER(C.c_int32), ip, C.POINTER(Stats)]
    i32p = C.POINTER(C.c_int32)
    L.gomilp_lp_upload_child.restype = C.c_int64
    L.gomilp_lp_upload_child.argtypes = [C.c_void_p, C.c_int64, C.c_int32, i32p, dp, dp]
    L.gomilp_pool_create.restype = C.c_void_p
    L.gomilp_pool_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.gomilp_pool_destroy.argtypes = [C.c_void_p]
    L.gomilp_pool_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.gomilp_pool_set_root.argtypes = [C.c_void_p, dp, dp, C.c_int64, dp, C.c_int64, C.c_int64]
    L.gomilp_frontier_solve.argtypes = [C.c_void_p, C.c_int64, ip, i32p, dp, dp, C.c_double, dp, dp, i32p, i32p,
                                        C.POINTER(FrontierStats)]
    L.gomilp_frontier_solve_warm.argtypes = [C.c_void_p, C.c_int64, ip, i32p, dp, dp, ip, ip, i32p, C.c_int32, C.c_double, dp, dp, i32p, i32p,
                                             C.POINTER(FrontierStats)]
    L.gomilp_pool_release_warm.argtypes = [C.c_void_p, C.c_int64]
    L.gomilp_pool_solve_root.argtypes = [C.c_void_p, C.c_double, dp, dp, i32p, C.POINTER(Stats)]
    L.gomilp_pool_add_root.argtypes = [C.c_void_p, dp, dp, C.c_int64, dp, C.c_int64, C.c_int64]
    L.gomilp_frontier_solve_roots.argtypes = [C.c_void_p, C.c_int64, i32p, ip, i32p, dp, dp, C.c_double, dp, dp, C.c_int64, i32p, i32p,
                                              C.POINTER(FrontierStats)]
    L.gomilp_debug_cond_estimate.restype = C.c_double
    L.gomilp_debug_cond_estimate.argtypes = [dp, C.c_int64, C.c_int]
    L.gomilp_debug_gonum_lu_cond.restype = C.c_int
    L.gomilp_debug_gonum_lu_cond.argtypes = [dp, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.gomilp_debug_find_independent.restype = C.c_int64
    L.gomilp_debug_find_independent_device.restype = C.c_int64
    L.gomilp_debug_find_independent_device.argtypes = [C.c_void_p, C.c_int64, ip, C.c_int64]
    L.gomilp_debug_find_independent.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, ip, C.c_int]
    L.gomilp_comm_unique_id.argtypes = [C.c_char_p]
    L.gomilp_comm_create.restype = C.c_void_p
    L.gomilp_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.gomilp_comm_destroy.argtypes = [C.c_void_p]
    L.gomilp_comm_rank.argtypes = [C.c_void_p]
    L.gomilp_comm_world.argtypes = [C.c_void_p]
    L.gomilp_incumbent_allreduce.argtypes = [C.c_void_p, C.c_double, C.c_int64, dp, ip]
    L.gomilp_incumbent_pick.restype = None
    L.gomilp_incumbent_pick.argtypes = [dp, C.c_int, dp, ip]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


@dataclass
class LPResult:
    status: int
    z: float
    x: Optional[np.ndarray]
    basis: Optional[np.ndarray]
    stats: dict = field(default_factory=dict)
    pivots: List[tuple] = field(default_factory=list)

    @property
    def ok(self) -> bool:
        return self.status == OK


def _stats_dict(s: Stats) -> dict:
    d = {k: getattr(s, k) for k, _ in Stats._fields_ if k not in ("pivot_kernel_seconds", "pipeline")}
    d["pipeline"] = {0: "three-kernel", 1: "fused", 2: "tableau", 3: "blocked"}.get(s.pipeline, str(s.pipeline))
    d["pivot_kernel_seconds"] = list(s.pivot_kernel_seconds)
    return d


def simplex(c, A, b, tol: float = 0.0, initial_basic=None) -> LPResult:
    """lp.Simplex drop-in through the flat C-ABI call (host buffers in, host buffers out)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    m, n = A.shape
    if c.shape != (n,) or b.shape != (m,):
        return LPResult(ERR_BAD_SHAPE, math.nan, None, None)  # the reference panics (simplex.go:387-398)
    x = np.zeros(n)
    basis = np.zeros(m, dtype=np.int64)
    z = C.c_double(math.nan)
    has_x = C.c_int32(0)
    st = Stats()
    ib = None if initial_basic is None else np.ascontiguousarray(initial_basic, dtype=np.int64)
    if ib is not None and ib.shape != (m,):
        return LPResult(ERR_PANIC, math.nan, None, None)  # "lp: incorrect number of initial vectors" (simplex.go:149-151)
    rc = lib().gomilp_lp_simplex(_dp(c), _dp(A), n, _dp(b), m, n, float(tol), None if ib is None else _ip(ib),
                                 C.byref(z), _dp(x), C.byref(has_x), _ip(basis), C.byref(st))
    return LPResult(rc, z.value, x if has_x.value else None, basis if has_x.value and m != n else None, _stats_dict(st))


class Context:
    """One engine context (HIP stream + work buffers) on one GPU; problems stay resident in HBM."""

    def __init__(self, device: int = -1, **knobs):
        st = C.c_int(0)
        self._h = lib().gomilp_ctx_create(int(device), C.byref(st))
        if not self._h:
            raise RuntimeError("gomilp_ctx_create failed: %s (no HIP device or libgomilp_hip.so not usable)"
                               % STATUS_NAMES.get(st.value, st.value))
        for k, v in knobs.items():
            self.set(k, v)

    def set(self, key: str, value: int) -> None:
        rc = lib().gomilp_ctx_set(self._h, key.encode(), int(value))
        if rc != OK:
            raise ValueError("bad knob %s=%s" % (key, value))

    @property
    def device(self) -> int:
        return lib().gomilp_ctx_device(self._h)

    def upload(self, c, A, b) -> "ResidentLP":
        A = np.ascontiguousarray(A, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        if c.shape != (n,) or b.shape != (m,):
            raise ValueError("lp: size mismatch")
        pid = lib().gomilp_lp_upload(self._h, _dp(c), _dp(A), n, _dp(b), m, n)
        if pid < 0:
            raise RuntimeError("gomilp_lp_upload failed: %s" % STATUS_NAMES.get(-pid, -pid))
        return ResidentLP(self, pid, m, n)

    def close(self) -> None:
        if self._h:
            lib().gomilp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentLP:
    def __init__(self, ctx: Context, pid: int, m: int, n: int):
        self.ctx, self.pid, self.m, self.n = ctx, pid, m, n

    def child(self, constraints) -> "ResidentLP":
        """B&B child assembled on the device: constraints = [(var, sign, rhs), ...] (subproblem.go:36-44)."""
        K = len(constraints)
        var = np.array([c[0] for c in constraints], dtype=np.int32)
        sign = np.array([c[1] for c in constraints], dtype=np.float64)
        rhs = np.array([c[2] for c in constraints], dtype=np.float64)
        pid = lib().gomilp_lp_upload_child(self.ctx._h, self.pid, K, var.ctypes.data_as(C.POINTER(C.c_int32)), _dp(sign),
                                           _dp(rhs))
        if pid < 0:
            raise RuntimeError("gomilp_lp_upload_child failed: %s" % STATUS_NAMES.get(-pid, -pid))
        return ResidentLP(self.ctx, pid, self.m + K, self.n + K)

    def solve(self, tol: float = 0.0, trace: bool = False, initial_basic=None) -> LPResult:
        L = lib()
        ib = None if initial_basic is None else np.ascontiguousarray(initial_basic, dtype=np.int64)
        if ib is not None and ib.shape != (self.m,):
            raise ValueError("lp: incorrect number of initial vectors")   # simplex.go:149-151 panics
        x = np.zeros(self.n)
        basis = np.zeros(self.m, dtype=np.int64)
        z = C.c_double(math.nan)
        has_x = C.c_int32(0)
        st = Stats()
        self.ctx.set("trace", 1 if trace else 0)
        rc = L.gomilp_lp_solve_resident(self.ctx._h, self.pid, float(tol), _ip(ib) if ib is not None else None, C.byref(z), _dp(x), C.byref(has_x),
                                        _ip(basis), C.byref(st))
        piv = []
        if trace:
            total = L.gomilp_lp_last_trace(self.ctx._h, None, 0)
            if total > 0:
                buf = (Pivot * total)()
                L.gomilp_lp_last_trace(self.ctx._h, buf, total)
                piv = [(p.phase, p.bland, p.min_idx, p.replace, p.entering, p.leaving) for p in buf]
        return LPResult(rc, z.value, x if has_x.value else None,
                        basis if has_x.value and self.m != self.n else None, _stats_dict(st), piv)

    def free(self) -> None:
        if self.pid >= 0:
            lib().gomilp_lp_free(self.ctx._h, self.pid)
            self.pid = -1


@dataclass
class FrontierResult:
    status: np.ndarray   # int32[count]
    z: np.ndarray        # float64[count]
    x: np.ndarray        # float64[count, n0] (rows valid where has_x)
    has_x: np.ndarray    # int32[count]
    stats: dict


class PackedChildren:
    """Children of a wave in the flat form the C-ABI takes (koff / var / sign / rhs): built once per child list — a host that keeps its
    frontier in this form (a Go host would) pays nothing per wave."""

    def __init__(self, children):
        count = len(children)
        self.count = count
        self.koff = np.zeros(count + 1, dtype=np.int64)
        for i, ch in enumerate(children):
            self.koff[i + 1] = self.koff[i] + len(ch)
        tot = int(self.koff[-1])
        self.var = np.zeros(max(tot, 1), dtype=np.int32)
        self.sign = np.zeros(max(tot, 1), dtype=np.float64)
        self.rhs = np.zeros(max(tot, 1), dtype=np.float64)
        k = 0
        for ch in children:
            for (v, s, r) in ch:
                self.var[k], self.sign[k], self.rhs[k] = v, s, r
                k += 1


def pack_children(children) -> PackedChildren:
    return children if isinstance(children, PackedChildren) else PackedChildren(children)


class FrontierPool:
    """`workers` engine contexts on one GPU solving independent child relaxations of one root concurrently
    (the solveWorker pool of /root/reference/tree.go:98-100,196-205 for one FIFO level)."""

    def __init__(self, device: int = -1, workers: int = 4, **knobs):
        st = C.c_int(0)
        self._h = lib().gomilp_pool_create(int(device), int(workers), C.byref(st))
        if not self._h:
            raise RuntimeError("gomilp_pool_create failed: %s" % STATUS_NAMES.get(st.value, st.value))
        self.n0 = self.m0 = 0
        for k, v in knobs.items():
            self.set(k, v)

    def set(self, key: str, value: int) -> None:
        """"batched" (1: device-batched pivot loops, 0: one worker thread + stream per relaxation) or any Context knob."""
        rc = lib().gomilp_pool_set(self._h, key.encode(), int(value))
        if rc != OK:
            raise ValueError("bad knob %s=%s" % (key, value))

    def set_root(self, c0, A0, b0) -> None:
        A0 = np.ascontiguousarray(A0, dtype=np.float64)
        c0 = np.ascontiguousarray(c0, dtype=np.float64)
        b0 = np.ascontiguousarray(b0, dtype=np.float64)
        m0, n0 = A0.shape
        rc = lib().gomilp_pool_set_root(self._h, _dp(c0), _dp(A0), n0, _dp(b0), m0, n0)
        if rc != OK:
            raise RuntimeError("gomilp_pool_set_root failed: %s" % STATUS_NAMES.get(rc, rc))
        self.m0, self.n0 = m0, n0
        self._widths = [n0]

    def solve_root(self, tol: float = 0.0) -> LPResult:
        """The root relaxation (subproblem.go:172) on the pool's first worker."""
        x = np.zeros(self.n0)
        z = C.c_double(math.nan)
        has_x = C.c_int32(0)
        st = Stats()
        rc = lib().gomilp_pool_solve_root(self._h, float(tol), C.byref(z), _dp(x), C.byref(has_x), C.byref(st))
        return LPResult(rc, z.value, x if has_x.value else None, None, _stats_dict(st))

    def add_root(self, c, A, b) -> int:
        """Another root in the same pool (resident once, on the pool's first worker): returns its index (>= 1)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        r = lib().gomilp_pool_add_root(self._h, _dp(c), _dp(A), n, _dp(b), m, n)
        if r < 0:
            raise RuntimeError("gomilp_pool_add_root failed: %s" % STATUS_NAMES.get(-r, -r))
        self._widths = getattr(self, "_widths", [self.n0]) + [n]
        return r

    def solve(self, children, tol: float = 0.0, roots=None) -> FrontierResult:
        """children: list of constraint lists [(var, sign, rhs), ...]; roots: per child the index of its root
        (default: all children of the set_root problem).  Independent LPs: children [] of different roots."""
        pk = pack_children(children)   # (a PackedChildren is taken as is)
        count, koff, var, sign, rhs = pk.count, pk.koff, pk.var, pk.sign, pk.rhs
        z = np.full(count, math.nan)
        ldx = max(getattr(self, "_widths", [self.n0]))
        x = np.zeros((count, ldx))
        status = np.zeros(count, dtype=np.int32)
        has_x = np.zeros(count, dtype=np.int32)
        st = FrontierStats()
        i32p = C.POINTER(C.c_int32)
        rof = None if roots is None else np.ascontiguousarray(roots, dtype=np.int32)
        rc = lib().gomilp_frontier_solve_roots(self._h, count, None if rof is None else rof.ctypes.data_as(i32p), _ip(koff),
                                               var.ctypes.data_as(i32p), _dp(sign), _dp(rhs), float(tol), _dp(z), _dp(x), ldx,
                                               status.ctypes.data_as(i32p), has_x.ctypes.data_as(i32p), C.byref(st))
        if rc != OK:
            raise RuntimeError("gomilp_frontier_solve failed: %s" % STATUS_NAMES.get(rc, rc))
        return FrontierResult(status, z, x, has_x, {k: getattr(st, k) for k, _ in FrontierStats._fields_})

    def solve_warm(self, children, parents=None, tags=None, keep=None, dual_budget: int = 0, tol: float = 0.0) -> FrontierResult:
        """gomilp_frontier_solve_warm (opt-in warm start, include/gomilp_lp.h): children of the set_root problem; parents[i] = tag of a kept
        relaxation that is child i minus its last constraint (or -1), tags[i] = the id child i is kept under when keep[i]."""
        count = len(children)
        koff = np.zeros(count + 1, dtype=np.int64)
        for i, ch in enumerate(children):
            koff[i + 1] = koff[i] + len(ch)
        tot = int(koff[-1])
        var = np.zeros(max(tot, 1), dtype=np.int32)
        sign = np.zeros(max(tot, 1), dtype=np.float64)
        rhs = np.zeros(max(tot, 1), dtype=np.float64)
        k = 0
        for ch in children:
            for (v, s, r) in ch:
                var[k], sign[k], rhs[k] = v, s, r
                k += 1
        par = np.full(count, -1, dtype=np.int64) if parents is None else np.ascontiguousarray(parents, dtype=np.int64)
        tg = np.full(count, -1, dtype=np.int64) if tags is None else np.ascontiguousarray(tags, dtype=np.int64)
        kp = np.zeros(count, dtype=np.int32) if keep is None else np.ascontiguousarray(keep, dtype=np.int32)
        z = np.full(count, math.nan)
        x = np.zeros((count, self.n0))
        status = np.zeros(count, dtype=np.int32)
        has_x = np.zeros(count, dtype=np.int32)
        st = FrontierStats()
        i32p = C.POINTER(C.c_int32)
        rc = lib().gomilp_frontier_solve_warm(self._h, count, _ip(koff), var.ctypes.data_as(i32p), _dp(sign), _dp(rhs), _ip(par), _ip(tg),
                                              kp.ctypes.data_as(i32p), int(dual_budget), float(tol), _dp(z), _dp(x),
                                              status.ctypes.data_as(i32p), has_x.ctypes.data_as(i32p), C.byref(st))
        if rc != OK:
            raise RuntimeError("gomilp_frontier_solve_warm failed: %s" % STATUS_NAMES.get(rc, rc))
        return FrontierResult(status, z, x, has_x, {k: getattr(st, k) for k, _ in FrontierStats._fields_})

    def release_warm(self, tag: int = -1) -> None:
        lib().gomilp_pool_release_warm(self._h, int(tag))

    def close(self) -> None:
        if self._h:
            lib().gomilp_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


NO_INCUMBENT = 1 << 52   # include/gomilp_lp.h: GOMILP_NO_INCUMBENT
COMM_ID_BYTES = 128


def find_independent(A, fast: bool = True):
    """Host-side column search of findLinearlyIndependent as the engine performs it (diagnostic; no GPU needed)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    idx = np.zeros(m, dtype=np.int64)
    cnt = lib().gomilp_debug_find_independent(_dp(A), n, m, n, _ip(idx), 1 if fast else 0)
    if cnt < 0:
        raise ValueError("bad shape")
    return [int(v) for v in idx[:cnt]]


def find_independent_device(A, device: int = -1, **knobs):
    """The same search with the column scan on the GPU (general_block.hip: blocks of candidates; knob general_block=0:
    general_kernels.hip, one candidate at a time; the last, square step on the host)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    cx = Context(device, **knobs)
    try:
        p = cx.upload(np.zeros(n), A, np.zeros(m))
        idx = np.zeros(m, dtype=np.int64)
        cnt = lib().gomilp_debug_find_independent_device(cx._h, p.pid, _ip(idx), m)
        if cnt < 0:
            raise RuntimeError("device search failed: %s" % STATUS_NAMES.get(-cnt, -cnt))
        return [int(v) for v in idx[:cnt]]
    finally:
        cx.close()


def comm_unique_id() -> bytes:
    """Rank 0: the 128-byte RCCL id every rank needs for Comm()."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().gomilp_comm_unique_id(buf)
    if rc != OK:
        raise RuntimeError("gomilp_comm_unique_id failed: %s" % STATUS_NAMES.get(rc, rc))
    return buf.raw


def incumbent_pick(table) -> tuple:
    """Host logic of the exchange: lexicographic minimum of a (world, 2) table of (z, index), +inf = no candidate."""
    t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1, 2)
    z = C.c_double(math.inf)
    i = C.c_int64(NO_INCUMBENT)
    lib().gomilp_incumbent_pick(_dp(t), t.shape[0], C.byref(z), C.byref(i))
    return z.value, i.value


class Comm:
    """RCCL communicator of the sharded frontier (one process per GPU): gomilp_incumbent_allreduce through the C-ABI."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int = -1):
        st = C.c_int(0)
        self._h = lib().gomilp_comm_create(int(rank), int(world), unique_id, int(device), C.byref(st))
        if not self._h:
            raise RuntimeError("gomilp_comm_create failed: %s" % STATUS_NAMES.get(st.value, st.value))
        self.rank, self.world = rank, world

    def incumbent_allreduce(self, z_local: float, index_local: int) -> tuple:
        z = C.c_double(math.inf)
        i = C.c_int64(NO_INCUMBENT)
        rc = lib().gomilp_incumbent_allreduce(self._h, float(z_local), int(index_local), C.byref(z), C.byref(i))
        if rc != OK:
            raise RuntimeError("gomilp_incumbent_allreduce failed: %s" % STATUS_NAMES.get(rc, rc))
        return z.value, i.value

    def close(self) -> None:
        if self._h:
            lib().gomilp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def debug_gonum_lu_cond(M, transposed: bool = False):
    """(cond, det_is_zero) as gonum's mat.LU holds them after Factorize(M) (transposed: of M.T) — host only, n <= 64 (gonum_cond.cpp)."""
    M = np.ascontiguousarray(M, dtype=np.float64)
    n = M.shape[0]
    assert M.shape == (n, n)
    c = C.c_double(0.0)
    rc = lib().gomilp_debug_gonum_lu_cond(_dp(M), n, 1 if transposed else 0, C.byref(c))
    if rc < 0:
        raise ValueError("n out of range")
    return float(c.value), bool(rc)


def debug_cond_estimate(B, inf: bool = False) -> float:
    """|B| * (Hager / Higham estimate of |B^-1|) in the 1-norm (inf = False) or the infinity norm — host only (no device)."""
    B = np.ascontiguousarray(B, dtype=np.float64)
    n = B.shape[0]
    assert B.shape == (n, n)
    return float(lib().gomilp_debug_cond_estimate(_dp(B), n, 1 if inf else 0))
