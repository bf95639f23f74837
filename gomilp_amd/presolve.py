"""Host-side mirror of GoMILP's presolve / postsolve (/root/reference/presolve.go) — SURVEY.md §8(f) rank 4.

Not on the LP hot path: this is the caller's side of the boundary, kept so that a `Problem` built through `gomilp_amd.api` goes
through exactly the steps `Problem.SolveWithCtx` takes (api.go:293-316) before and after the tree search whose relaxations
run on the GPU.  The reference's quirks are reproduced, not repaired, because they decide what a caller sees:

* a fixed variable comes back from postsolve as `coefficient * lower` — its objective contribution, not its value —
  and the right-hand sides are reduced by the same product instead of `a_ij * lower` (presolve.go:141,152);
* `Solution.Objective` is the SUM OF THE VARIABLE VALUES, not c^T x (presolve.go:92-95);
* `findImplicitlyFixedVars` rewrites the bounds of the caller's own variables (presolve.go:226-229);
* of a pair of duplicated constraints the one with the larger right-hand side is dropped — and with EQUAL right-hand sides
  both are (presolve.go:283-287); three copies of a row leave several references to the same constraint.

The reference's unconditional debug prints (presolve.go:51-73,145,220,238,288) are not part of any result and are dropped.
"""
from __future__ import annotations

from typing import Callable, Dict, List


class PreProcessor:
    """preProcessor (presolve.go:14-47): the undo stack of the reductions that were applied."""

    def __init__(self) -> None:
        self.undoers: List[Callable[[Dict[str, float]], Dict[str, float]]] = []

    # presolve.go:49-76
    def pre_solve(self, p):
        prepped = sanitize_problem(p)
        previous = 0
        while True:
            prepped = self.filter_fixed_vars(prepped)
            prepped = self.find_implicitly_fixed_vars(prepped)
            prepped = remove_empty_constraints(prepped)
            prepped = remove_duplicate_constraints(prepped)
            if len(self.undoers) == previous:
                break
            previous = len(self.undoers)
        return prepped

    # presolve.go:78-99
    def post_solve(self, raw: Dict[str, float]):
        from .api import Solution
        post = raw
        for undo in reversed(self.undoers):
            post = undo(post)
        sol = Solution(0.0, {})
        for name, value in post.items():
            sol.by_name[name] = value
            sol.objective = sol.objective + value      # (sic) presolve.go:94
        return sol

    # presolve.go:128-182
    def filter_fixed_vars(self, p):
        filtered = p.shallow_copy()
        new_vars, fixed = [], {}
        for v in filtered.variables:
            if not is_fixed(v):
                new_vars.append(v)
            else:
                fixed[v.name] = v.coefficient * v.lower          # (sic) :141
        filtered.variables = new_vars
        for c in filtered.constraints:                            # the Constraint objects are shared with the caller's Problem
            keep = []
            for e in c.expressions:
                if is_fixed(e.variable):
                    c.rhs = c.rhs - (e.variable.coefficient * e.variable.lower)   # (sic) :152
                else:
                    keep.append(e)
            c.expressions = keep
        if fixed:
            def undo(s: Dict[str, float]) -> Dict[str, float]:
                for name, value in fixed.items():
                    if name in s:
                        raise RuntimeError("variable %s already in raw solution" % name)   # panic, :168
                    s[name] = value
                return s
            self.undoers.append(undo)
        return filtered

    # presolve.go:188-232
    def find_implicitly_fixed_vars(self, p):
        implicit_zero = []
        for c in p.constraints:
            if c.rhs == 0 and all(not (e.coef < 0) for e in c.expressions):
                for e in c.expressions:
                    if e.coef > 0 and not any(e.variable is v for v in implicit_zero):
                        implicit_zero.append(e.variable)
        for v in implicit_zero:
            v.lower_bound(0).upper_bound(0)          # modifies the caller's variables (:226-229)
        return p


def is_fixed(v) -> bool:                               # presolve.go:118-123
    return v.lower == v.upper


def sanitize_problem(p):                               # presolve.go:104-110
    for c in p.constraints:
        c.expressions = [e for e in c.expressions if e.coef != 0]
    return p


def remove_empty_constraints(p):                       # presolve.go:235-246
    p = p.shallow_copy()
    p.constraints = [c for c in p.constraints if len(c.expressions) > 0]
    return p


def remove_duplicate_constraints(p):                   # presolve.go:249-295
    sets = [frozenset("%s-%r" % (e.variable.name, e.coef) for e in c.expressions) for c in p.constraints]
    equal_pairs, retained = [], []
    for i, s in enumerate(sets):
        unique = True
        for j in range(len(sets)):
            if i == j:
                continue
            if sets[j] == s:
                equal_pairs.append((p.constraints[i], p.constraints[j]))
                unique = False
        if unique:
            retained.append(p.constraints[i])
    for a, b in equal_pairs:
        if a.rhs > b.rhs:                              # equal right-hand sides: neither is kept (:283-287)
            retained.append(b)
    p = p.shallow_copy()
    p.constraints = retained
    return p
