"""Host-side branch-and-bound driver over the GPU relaxation engine (BASELINE config 3).

GoMILP's own control plane (`api.go`, `ilp.go`, `tree.go`, `branching.go`) stays Go in a real deployment and only its
`lp.Simplex` calls move to the GPU.  Go is not available to this pipeline, so the benches and
parity tests drive the engine with this mirror of the caller semantics:

* `toInitialSubproblem`            /root/reference/ilp.go:43-72      (convertToEqualities, subproblem.go:81-139)
* FIFO frontier, serial check      /root/reference/tree.go:66-263    (checkSolution :207-263)
* branching variable               /root/reference/branching.go:54-72 — `maxFunBranchPoint` never updates its candidate
  value, so it is the LAST integer-constrained index; the heuristic field is never copied (ilp.go:59-70), so this is
  the only heuristic that ever runs
* children                         /root/reference/subproblem.go:193-259: x_j <= floor(x_j*), -x_j <= -(floor(x_j*)+1)
* integrality                      /root/reference/tree.go:276-297 (exact equality with Trunc)
* error map                        /root/reference/ilp.go:37-40, tree.go:266-273 (ErrInfeasible / ErrSingular prune; else panic)

The reference solves nodes on goroutine workers and checks them serially; decisions only depend on the order of
the checks.  Here a whole FIFO level ("wave") is solved at once by `FrontierPool` (tree.go:98-100 with as many
workers as the GPU has streams) and then checked in node-id order — the order the reference produces with one
worker.  The incumbent never enters a solve (tree.go:228-230 prunes after the fact), so the results are identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import lp
from .frontier import feasible_for_ip


@dataclass
class Node:
    id: int
    parent: int
    constraints: list            # [(var, sign, rhs), ...]  == bnbConstraint list (subproblem.go:36-44)
    status: int = -1
    z: float = math.nan
    x: Optional[np.ndarray] = None
    decision: str = ""


@dataclass
class Result:
    error: Optional[str]         # None | "DeadlineExceeded" | "NO_INTEGER_FEASIBLE_SOLUTION" | "panic:<status>"
    x: Optional[np.ndarray]
    z: float
    nodes: List[Node] = field(default_factory=list)
    waves: int = 0
    relaxations: int = 0
    pivots: int = 0
    warm_started: int = 0        # warm mode: relaxations that started from their parent's basis, of which handed back to the cold path
    warm_fallbacks: int = 0
    pivots_dual: int = 0


def convert_to_equalities(c, A, b, G, h):
    """subproblem.go:81-139: [[A, 0], [G, I]], c' = [c, 0], b' = [b; h] (A may be None)."""
    c = np.asarray(c, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    nvar, nineq = c.shape[0], h.shape[0]
    ncons = 0 if A is None else np.asarray(A).shape[0]
    a_new = np.zeros((ncons + nineq, nvar + nineq))
    if A is not None:
        a_new[:ncons, :nvar] = A
    a_new[ncons:, :nvar] = G
    a_new[ncons:, nvar:] = np.eye(nineq)
    b_new = np.concatenate([np.zeros(0) if A is None else np.asarray(b, dtype=np.float64), h])
    return np.concatenate([c, np.zeros(nineq)]), a_new, b_new


def max_fun_branch_point(c, integrality) -> int:
    cur = 0
    for i in range(len(c)):
        if integrality[i]:
            cur = i          # `math.Abs(v) >= candidateValue` with candidateValue stuck at 0 (branching.go:59-67)
    return cur


def solve_milp(c, A, b, G, h, integrality, *, max_nodes: int = 255, workers: int = 8, device: int = -1,
               pool=None, warm: bool = False, dual_budget: int = 0) -> Result:
    """milpProblem.solve (ilp.go:75-116) with every relaxation on the GPU.  `max_nodes` stands in for the context
    deadline of the reference (its tree does not terminate on many inputs: SURVEY.md §3.4).

    warm (opt-in; /root/reference/README.md TODO "initiate the simplex at solution of parent"): every node that branches keeps its final
    basis resident (gomilp_frontier_solve_warm), its two children start from it with the dual simplex; a parent is released once both
    children are solved.  Statuses, decisions and z agree with the cold run to 1e-9 (tests), the pivot paths do not."""
    c = np.asarray(c, dtype=np.float64)
    integrality = list(integrality)
    if G is not None:
        c0, A0, b0 = convert_to_equalities(c, A, b, G, h)
        int0 = integrality + [False] * (len(c0) - len(c))
    else:
        c0, A0, b0 = c, np.asarray(A, dtype=np.float64), np.asarray(b, dtype=np.float64)
        int0 = integrality
    n0 = len(c0)
    out = Result(None, None, math.nan)
    root = Node(0, 0, [])
    out.nodes.append(root)
    own_pool = pool is None
    if own_pool:
        pool = lp.FrontierPool(device=device, workers=workers)   # (a caller that solves several MILPs keeps one pool)
    pool.set_root(c0, A0, b0)
    if warm:   # the root through the batched schedule, so that its final state can stay resident for its children
        rr = pool.solve_warm([[]], tags=[0], keep=[1])
        r = lp.LPResult(int(rr.status[0]), float(rr.z[0]), rr.x[0].copy() if rr.has_x[0] else None, None, rr.stats)
    else:
        r = pool.solve_root(0.0)                         # subproblem.go:172
    root.status, root.z, root.x = r.status, r.z, r.x
    out.relaxations, out.pivots = 1, r.stats["pivots_phase1"] + r.stats["pivots_phase2"]
    if r.status != lp.OK:
        out.error = "panic:" + lp.STATUS_NAMES.get(r.status, str(r.status))   # subproblem.go:173-176
        if own_pool:
            pool.close()
        return out
    if feasible_for_ip(int0, r.x):
        root.decision = "INITIAL_RX_FEASIBLE_FOR_IP"
        out.x, out.z = r.x[: len(c)].copy(), r.z
        if own_pool:
            pool.close()
        return out
    incumbent: Optional[Node] = None
    queue: List[Node] = []
    next_id = 0
    kids_left: dict = {}

    def check(node: Node) -> Optional[str]:
        nonlocal incumbent, next_id
        inc_z = math.inf if incumbent is None else incumbent.z
        if node.status != lp.OK:
            if node.status == lp.ERR_INFEASIBLE:
                node.decision = "SUBPROBLEM_IS_DEGENERATE"      # labels swapped in the reference (ilp.go:37-40)
            elif node.status == lp.ERR_SINGULAR:
                node.decision = "SUBPROBLEM_NOT_FEASIBLE"
            else:
                return "panic:" + lp.STATUS_NAMES.get(node.status, str(node.status))
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
                for sign, rhs in ((1, fl), (-1, -(fl + 1))):
                    next_id += 1
                    ch = Node(next_id, node.id, node.constraints + [(j, sign, float(rhs))])
                    out.nodes.append(ch)
                    queue.append(ch)
        else:
            return "panic:unexpected case"
        return None

    err = check(root)
    if err:
        out.error = err
        if own_pool:
            pool.close()
        return out
    try:
        solved = 0
        while queue:
            budget = max_nodes - solved
            if budget <= 0:
                out.error = "DeadlineExceeded"
                break
            wave, queue = queue[:budget], queue[budget:]
            if warm:
                res = pool.solve_warm([nd.constraints for nd in wave], parents=[nd.parent for nd in wave], tags=[nd.id for nd in wave],
                                      keep=[1] * len(wave), dual_budget=dual_budget)
                out.pivots += res.stats["pivots_dual"]
                out.warm_started += res.stats["warm_started"]; out.warm_fallbacks += res.stats["warm_fallbacks"]; out.pivots_dual += res.stats["pivots_dual"]
                for nd in wave:   # a parent whose two children are solved is not needed any more
                    kids_left[nd.parent] = kids_left.get(nd.parent, 2) - 1
                    if kids_left[nd.parent] == 0:
                        pool.release_warm(nd.parent)
            else:
                res = pool.solve([nd.constraints for nd in wave])
            out.waves += 1
            out.relaxations += len(wave)
            out.pivots += res.stats["pivots_phase1"] + res.stats["pivots_phase2"]
            solved += len(wave)
            pending = queue          # nodes beyond the budget stay queued behind this wave's children (FIFO)
            queue = []
            for i, nd in enumerate(wave):
                nd.status, nd.z = int(res.status[i]), float(res.z[i])
                nd.x = res.x[i].copy() if res.has_x[i] else None
                err = check(nd)
                if err:
                    out.error = err
                    return out
                if warm and nd.decision != "BETTER_THAN_INCUMBENT_BRANCHING":
                    pool.release_warm(nd.id)   # a leaf (pruned, integer feasible, infeasible): nobody will start from its basis
            queue = pending + queue
    finally:
        if warm:
            pool.release_warm(-1)   # whatever is still kept (the deadline, an early return): 2-3 MB of HBM per 520-row node
        if own_pool:
            pool.close()
    if out.error == "DeadlineExceeded":
        if incumbent is not None:
            out.x, out.z = incumbent.x[: len(c)].copy(), incumbent.z
        else:
            out.z = 0.0
        return out
    if incumbent is None:
        out.error, out.z = "NO_INTEGER_FEASIBLE_SOLUTION", 0.0
        return out
    out.x, out.z = incumbent.x[: len(c)].copy(), incumbent.z
    return out
