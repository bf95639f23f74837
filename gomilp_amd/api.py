"""Host-side mirror of GoMILP's fluent problem builder (/root/reference/api.go) — SURVEY.md §8(f) ranks 3 / 4.

The caller's side of the boundary: in a deployment `api.go` stays Go and only `subproblem.go`'s call into lp.Simplex is redirected
to the C-ABI; this module gives the benches and tests the same entry point — `Problem` / `Variable` /
`Constraint` with the reference's names, `to_solveable` (api.go:191-289: dense c, A, b, G, h with the bound rows appended the way
the reference appends them), `solve` = presolve -> toSolveable -> tree search over GPU relaxations -> postsolve
(api.go:293-316).  A context deadline becomes a node budget (`max_nodes`), as in gomilp_amd.bnb.

Known quirks of the reference that are kept: `BranchingHeuristic` is never copied into the tree (ilp.go:59-70: always
maxFun, which always returns the last integer index), `Solution.objective` is the sum of the variable values
(presolve.go:92-95), lower bounds <= 0 produce no row (api.go:261).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

BRANCH_MAXFUN, BRANCH_MOST_INFEASIBLE, BRANCH_NAIVE = 0, 1, 2     # branching.go:8-12 (ineffective: see the module docstring)


class Variable:
    """api.go:29-43, :93-113."""

    def __init__(self, name: str) -> None:
        self.name = name
        self.coefficient = 0.0
        self.integer = False
        self.upper = math.inf
        self.lower = 0.0

    def set_coeff(self, coef: float) -> "Variable":
        self.coefficient = float(coef)
        return self

    def is_integer(self) -> "Variable":
        self.integer = True
        return self

    def upper_bound(self, bound: float) -> "Variable":
        self.upper = float(bound)
        return self

    def lower_bound(self, bound: float) -> "Variable":
        self.lower = float(bound)
        return self


@dataclass
class Expression:                    # api.go:47-50
    coef: float
    variable: Variable


class Constraint:
    """api.go:52-64, :124-144.  An equality by default."""

    def __init__(self, problem: "Problem") -> None:
        self.expressions: List[Expression] = []
        self.rhs = 0.0
        self.inequality = False
        self.problem = problem

    def equal_to(self, val: float) -> "Constraint":
        self.inequality = False
        self.rhs = float(val)
        return self

    def smaller_than_or_equal_to(self, val: float) -> "Constraint":
        self.inequality = True
        self.rhs = float(val)
        return self

    def add_expression(self, coef: float, v: Variable) -> "Constraint":
        self.problem.get_variable_index(v)      # panics when the variable was not declared in this problem (api.go:135)
        self.expressions.append(Expression(float(coef), v))
        return self


@dataclass
class MilpProblem:                   # ilp.go:11-27
    c: np.ndarray
    A: Optional[np.ndarray]
    b: Optional[np.ndarray]
    G: Optional[np.ndarray]
    h: Optional[np.ndarray]
    integrality: List[bool]
    branching_heuristic: int = BRANCH_MAXFUN


@dataclass
class Solution:                      # presolve.go:23-38
    objective: float
    by_name: Dict[str, float] = field(default_factory=dict)

    def get_value_for(self, name: str) -> float:
        if name not in self.by_name:
            raise KeyError("Variable name %s not found in Solution" % name)
        return self.by_name[name]


class Problem:
    """api.go:11-27, :67-164."""

    def __init__(self) -> None:
        self.maximize_flag = False
        self.variables: List[Variable] = []
        self.constraints: List[Constraint] = []
        self.branching_heuristic = BRANCH_MAXFUN
        self.workers = 1
        self.instrumentation = None

    def shallow_copy(self) -> "Problem":
        """A Go `Problem` passed by value: new slice headers, the same *Variable / *Constraint objects."""
        q = Problem()
        q.maximize_flag, q.branching_heuristic, q.workers, q.instrumentation = self.maximize_flag, self.branching_heuristic, self.workers, self.instrumentation
        q.variables, q.constraints = list(self.variables), list(self.constraints)
        return q

    def add_variable(self, name: str) -> Variable:
        v = Variable(name)
        self.variables.append(v)
        return v

    def add_constraint(self) -> Constraint:
        c = Constraint(self)
        self.constraints.append(c)
        return c

    def maximize(self) -> None:
        self.maximize_flag = True

    def minimize(self) -> None:
        self.maximize_flag = False

    def set_branching_heuristic(self, choice: int) -> None:
        self.branching_heuristic = choice

    def set_workers(self, n: int) -> None:
        self.workers = n

    def set_instrumentation(self, b) -> None:
        self.instrumentation = b

    def check_expression(self, e: Expression) -> bool:      # api.go:167-178
        return any(v is e.variable for v in self.variables)

    def get_variable_index(self, v: Variable) -> int:       # api.go:181-188
        for i, va in enumerate(self.variables):
            if va is v:
                return i
        raise RuntimeError("variable pointer not found in Problem struct")

    # api.go:191-289
    def to_solveable(self) -> MilpProblem:
        nv = len(self.variables)
        c = np.array([(-v.coefficient if self.maximize_flag else v.coefficient) for v in self.variables], dtype=np.float64)
        integrality = [v.integer for v in self.variables]
        a_rows, b, g_rows, h = [], [], [], []
        for con in self.constraints:
            row = np.zeros(nv)
            for e in con.expressions:
                row[self.get_variable_index(e.variable)] = e.coef
            if con.inequality:
                g_rows.append(row); h.append(con.rhs)
            else:
                a_rows.append(row); b.append(con.rhs)
        for v in self.variables:                              # bounds as inequality rows, after the constraints (api.go:245-272)
            if not (v.upper == math.inf):                         # !math.IsInf(v.upper, 1)
                row = np.zeros(nv); row[self.get_variable_index(v)] = 1.0
                g_rows.append(row); h.append(v.upper)
            if not (v.lower <= 0):
                row = np.zeros(nv); row[self.get_variable_index(v)] = -1.0
                g_rows.append(row); h.append(-v.lower)
        A = np.array(a_rows, dtype=np.float64).reshape(len(b), nv) if b else None
        G = np.array(g_rows, dtype=np.float64).reshape(len(h), nv) if h else None
        return MilpProblem(c, A, np.array(b, dtype=np.float64) if b else None, G, np.array(h, dtype=np.float64) if h else None,
                           integrality, self.branching_heuristic)

    # api.go:293-322
    def solve(self, *, max_nodes: int = 255, milp_solver: Optional[Callable] = None, **solver_args) -> Solution:
        """SolveWithCtx: presolve, toSolveable, milpProblem.solve, postsolve.  `milp_solver(c, A, b, G, h, integrality,
        max_nodes=...)` defaults to gomilp_amd.bnb.solve_milp (every relaxation on the GPU); the result needs `.error`, `.x`.
        Errors of the tree search are raised like the reference returns them (nil solution + error)."""
        from .presolve import PreProcessor
        pre = PreProcessor()
        prepped = pre.pre_solve(self)
        milp = prepped.to_solveable()
        if milp_solver is None:
            from . import bnb
            milp_solver = bnb.solve_milp
        res = milp_solver(milp.c, milp.A, milp.b, milp.G, milp.h, milp.integrality, max_nodes=max_nodes, **solver_args)
        if res.error is not None:
            raise MilpError(res.error)
        raw = {v.name: float(res.x[i]) for i, v in enumerate(prepped.variables)}
        return pre.post_solve(raw)


class MilpError(RuntimeError):
    """milpProblem.solve returned an error (ilp.go:93-108): "DeadlineExceeded", "NO_INTEGER_FEASIBLE_SOLUTION", "panic:..."."""
